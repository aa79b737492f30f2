// Micro-benchmark (dev tool): cost of executing straight-line code for the first time in a launch
// (cold instruction cache) vs the second time, on MI355X.  One wave per workgroup executes a block
// of N independent-ish VALU instructions twice; s_memtime stamps around each pass.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

#define BODY(N) asm volatile(".rept " #N "\n v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %1\n .endr" : "+v"(x), "+v"(z) : "v"(y), "v"(z));

template <int KB>
__global__ void k(float* out, unsigned long long* st, int iters) {
  float x = threadIdx.x, y = 1.0f, z = 2.0f;
  unsigned long long t[4];
  for (int it = 0; it < iters && it < 3; ++it) {
    t[it] = __builtin_readcyclecounter();
    if (KB == 1) { BODY(128) }       // 256 instrs x 4 B = 1 KB
    else if (KB == 4) { BODY(512) }
    else if (KB == 16) { BODY(2048) }
    asm volatile("s_nop 0" ::: "memory");
  }
  t[3] = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x + z;
  if (threadIdx.x == 0) {
    st[blockIdx.x * 4 + 0] = t[1] - t[0];
    st[blockIdx.x * 4 + 1] = t[2] - t[1];
    st[blockIdx.x * 4 + 2] = t[3] - t[2];
  }
}

template <int KB>
void run(int blocks, int threads) {
  float* out;
  unsigned long long* st;
  hipMalloc(&out, blocks * threads * 4);
  hipMalloc(&st, blocks * 32);
  std::vector<unsigned long long> h(blocks * 4);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k<KB>, dim3(blocks), dim3(threads), 0, 0, out, st, 3);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), st, blocks * 32, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> a, b, c;
    for (int i = 0; i < blocks; ++i) { a.push_back(h[4 * i]); b.push_back(h[4 * i + 1]); c.push_back(h[4 * i + 2]); }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end()); std::sort(c.begin(), c.end());
    printf("%2d KB code, %4d blocks x %3d thr, launch %d: pass1 med %6llu max %6llu | pass2 med %6llu | pass3 med %6llu cycles\n", KB, blocks,
           threads, rep, a[blocks / 2], a.back(), b[blocks / 2], c[blocks / 2]);
  }
  hipFree(out); hipFree(st);
}

int main() {
  run<1>(256, 64); run<4>(256, 64); run<16>(256, 64);
  run<4>(256, 512); run<16>(256, 512);
  run<16>(1, 64);
  return 0;
}
