// Micro-benchmark (dev tool): what a kernel's cold start costs inside a graph, with its arguments read by scalar loads
// (default) or preloaded into user SGPRs by the dispatcher (-mllvm -amdgpu-kernarg-preload-count=14).
// A graph of NP (flush, tiny) pairs against a graph of NP flushes; the difference / NP = cost of one tiny launch whose
// arguments, code and data are cold.  Build it twice:
//   hipcc --offload-arch=gfx950 -O3 -w -o /tmp/kargs0 scripts/ubench/kargs.hip
//   hipcc --offload-arch=gfx950 -O3 -w -mllvm -amdgpu-kernarg-preload-count=14 -o /tmp/kargs1 scripts/ubench/kargs.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void flush(f32x4* p, size_t n, float v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (f32x4){v, v, v, v};
}
// uses every argument right away (an address from each pointer)
__global__ __launch_bounds__(256) void tiny(const float* a, const float* b, const float* c, float* y, int n, int lda, float s, float t) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = a[i] * s + b[(size_t)i * lda % n] * t + c[i];
}

static float run_graph(hipGraphExec_t g, hipStream_t st, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipGraphLaunch(g, st); hipStreamSynchronize(st);
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(e0, st); hipGraphLaunch(g, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  return best;
}

int main() {
  const int NP = 100, N = 65536;
  const size_t FL = (size_t)96 << 20;  // bytes flushed between launches (> 8 x 4 MB of L2)
  char* fl; float *a, *b, *c, *y;
  hipMalloc(&fl, FL); hipMalloc(&a, N * 4 * NP); hipMalloc(&b, N * 4 * NP); hipMalloc(&c, N * 4 * NP); hipMalloc(&y, N * 4 * NP);
  hipMemset(a, 0, N * 4 * NP); hipMemset(b, 0, N * 4 * NP); hipMemset(c, 0, N * 4 * NP);
  hipStream_t st; hipStreamCreate(&st);
  hipGraph_t g0, g1; hipGraphExec_t x0, x1;
  for (int with = 0; with < 2; ++with) {
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < NP; ++i) {
      hipLaunchKernelGGL(flush, dim3(1024), dim3(256), 0, st, (f32x4*)fl, FL / 16, (float)i);
      if (with) hipLaunchKernelGGL(tiny, dim3(N / 256), dim3(256), 0, st, a + (size_t)i * N, b + (size_t)i * N, c + (size_t)i * N, y + (size_t)i * N, N, 7, 1.5f, 0.5f);
    }
    hipStreamEndCapture(st, with ? &g1 : &g0);
    hipGraphInstantiate(with ? &x1 : &x0, with ? g1 : g0, nullptr, nullptr, 0);
  }
  const float t0 = run_graph(x0, st, 5), t1 = run_graph(x1, st, 5);
  printf("flush only %.3f ms, flush + tiny %.3f ms -> %.2f us per cold tiny launch (%d pairs)\n", t0, t1, (t1 - t0) * 1e3 / NP, NP);
  // warm: tiny launches back to back
  hipGraph_t g2; hipGraphExec_t x2;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < NP; ++i)
    hipLaunchKernelGGL(tiny, dim3(N / 256), dim3(256), 0, st, a, b, c, y, N, 7, 1.5f, 0.5f);
  hipStreamEndCapture(st, &g2);
  hipGraphInstantiate(&x2, g2, nullptr, nullptr, 0);
  printf("warm back-to-back: %.2f us per tiny launch\n", run_graph(x2, st, 5) * 1e3 / NP);
  return 0;
}
