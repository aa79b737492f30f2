// Micro-benchmark (dev tool): how fast ONE workgroup (8 waves) can pull a region it touches ONCE (the weight stream of
// the fused row-chain kernels) into VGPRs, in shader clocks measured inside the kernel, by ring depth, region size,
// workgroups on the chip, and whether the region was just read (L2 warm) or not.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/burst scripts/ubench/burst.hip && /tmp/burst
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int D>
__global__ __launch_bounds__(512) void burst(const char* src, int nfr_wave, int own, size_t region, unsigned long long* out, float* sink) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* base = src + (own ? (size_t)blockIdx.x * region : 0) + lane * 16;
  const unsigned long long t0 = __builtin_readcyclecounter();
  f32x4 ring[D];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  unsigned f = wave;
#pragma unroll
  for (int d = 0; d < D; ++d) { ring[d] = *(const f32x4*)(base + (size_t)f * 1024); f += 8; }
  const unsigned long long t1 = __builtin_readcyclecounter();
  for (int it = D; it < nfr_wave; it += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      acc += ring[d];
      ring[d] = *(const f32x4*)(base + (size_t)f * 1024);
      f += 8;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int d = 0; d < D; ++d) acc += ring[d];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.f) sink[0] = acc[0];
  const unsigned long long t2 = __builtin_readcyclecounter();
  if (lane == 0 && wave == 7) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = t2 - t0; }
}

// every workgroup reads ONE slice of the region (1 / gridDim of it): what a previous kernel could do for the next one's weights
__global__ __launch_bounds__(64) void touch(const char* src, size_t region, float* sink) {
  const size_t per = region / gridDim.x;
  const char* base = src + (size_t)blockIdx.x * per + threadIdx.x * 16;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t o = 0; o < per; o += 1024) acc += *(const f32x4*)(base + o);
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.f) sink[0] = acc[0];
}

// cold region, touched once by `tgrid` workgroups of a previous kernel (each line by ONE workgroup, i.e. one XCD), then
// `mid_mb` MB of other traffic, then the shared sweep by 256 workgroups
template <int D>
void run_touched(char* src, char* flush, int kb, int tgrid, int mid_mb, unsigned long long* out) {
  const size_t region = (size_t)kb << 10;
  std::vector<unsigned long long> h(512);
  double best = 1e18;
  for (int rep = 0; rep < 4; ++rep) {
    hipMemset(flush, rep, (size_t)1 << 30);
    hipDeviceSynchronize();
    if (tgrid) hipLaunchKernelGGL(touch, dim3(tgrid), dim3(64), 0, 0, src, region, (float*)nullptr);
    if (mid_mb) hipMemsetAsync(flush, rep + 1, (size_t)mid_mb << 20, 0);
    hipLaunchKernelGGL(burst<D>, dim3(256), dim3(512), 0, 0, src, kb / 8, 0, region, out, (float*)nullptr);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), out, 256 * 16, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 256; ++i) s += (double)h[2 * i + 1];
    best = std::min(best, s / 256);
  }
  printf("D=%2d %5d KB shared, cold, touched by %3d WGs of the previous kernel, %3d MB in between: %8.0f clk -> %5.1f B/clk per CU\n", D, kb,
         tgrid, mid_mb, best, region / best);
  fflush(stdout);
}

template <int D>
void run(char* src, char* flush, int kb, int grid, int own, int cold, unsigned long long* out) {
  const size_t region = (size_t)kb << 10;
  const int nfr_wave = kb / 8;  // 1 KiB fragments per wave
  std::vector<unsigned long long> h(grid * 2);
  double best = 1e18, avg = 0;
  for (int rep = 0; rep < 5; ++rep) {
    if (cold) hipMemset(flush, rep, (size_t)1 << 30);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(burst<D>, dim3(grid), dim3(512), 0, 0, src, nfr_wave, own, region, out, (float*)nullptr);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), out, grid * 16, hipMemcpyDeviceToHost);
    if (rep == 0) continue;
    double s = 0;
    for (int i = 0; i < grid; ++i) s += (double)h[2 * i + 1];
    s /= grid;
    avg += s / 4;
    best = std::min(best, s);
  }
  printf("D=%2d %5d KB/WG grid %3d %-6s %-5s: %8.0f clk  -> %5.1f B/clk per CU  (first ring issued after %llu clk)\n", D, kb, grid,
         own ? "own" : "shared", cold ? "cold" : "warm", best, region / best, h[0]);
  fflush(stdout);
}

int main() {
  char *src, *flush;
  unsigned long long* out;
  hipMalloc(&src, (size_t)1 << 30);
  hipMalloc(&flush, (size_t)1 << 30);
  hipMalloc(&out, 4096 * 16);
  hipMemset(src, 1, (size_t)1 << 30);
  for (int kb : {448, 1408, 3584})
    for (int tgrid : {0, 8, 256})
      for (int mid : {0, 16, 64}) run_touched<8>(src, flush, kb, tgrid, mid, out);
  for (int kb : {448})
    for (int grid : {1, 32, 256})
      for (int own : {0, 1})
        for (int cold : {0, 1}) {
          if (own && (size_t)kb * grid > (1 << 20)) continue;
          run<8>(src, flush, kb, grid, own, cold, out);
          run<16>(src, flush, kb, grid, own, cold, out);
          run<32>(src, flush, kb, grid, own, cold, out);
        }
  return 0;
}
