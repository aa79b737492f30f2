// Micro-benchmark (dev tool): per-CU rate of streaming packed WEIGHT fragments (1 KiB contiguous per wave instruction)
// out of the L2 on MI355X, by destination (VGPRs vs LDS-DMA) and by sharing: every workgroup its own region, or ALL
// workgroups sweeping the same region at the same time (what the workgroups of one N tile of a GEMM do).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/wstream scripts/ubench/wstream.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: global_load_dwordx4 -> VGPR, ring of D loads per wave;  MODE 1: global_load_lds_dwordx4, D in flight per wave
template <int MODE, int D, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void stream(const char* src, size_t region, int shared, int rot, int iters, float* sink) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* base = src + (shared ? 0 : (size_t)blockIdx.x * region);
  const unsigned nfr = (unsigned)(region / 1024);  // 1 KiB fragments in the region
  // wave w takes fragments w, w + WAVES, ...; optionally every workgroup starts at a different fragment
  unsigned f = (unsigned)wave + (rot ? (blockIdx.x * 37u) % nfr : 0u);
  f32x4 ring[D];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const unsigned loff = lane * 16;
  if (MODE == 0) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      ring[d] = *(const f32x4*)(base + (size_t)(f % nfr) * 1024 + loff);
      f += WAVES;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        acc += ring[d];
        ring[d] = *(const f32x4*)(base + (size_t)(f % nfr) * 1024 + loff);
        f += WAVES;
      }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) acc += ring[d];
  } else {
    char* dst = lds + wave * (D * 1024);
    for (int it = 0; it < iters + 1; ++it) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        __builtin_amdgcn_global_load_lds((glb_ptr)(base + (size_t)(f % nfr) * 1024 + loff), (lds_ptr)(dst + d * 1024), 16, 0, 0);
        f += WAVES;
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D / 2) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc[0] = ((float*)lds)[lane];
  }
  if (sink && acc[0] + acc[1] + acc[2] + acc[3] == 123.f) sink[0] = acc[0];
}

template <int MODE, int D, int WAVES>
void run(const char* src, size_t region, int shared, int rot, double clk) {
  const int iters = 1024 / D * 4;
  hipFuncSetAttribute((const void*)stream<MODE, D, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream<MODE, D, WAVES>), dim3(256), dim3(WAVES * 64), 128 * 1024, 0, src, region, shared, rot, iters, nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double bytes_cu = (double)(iters + 1) * D * WAVES * 1024;
  const double gbs = bytes_cu / (ms * 1e-3) / 1e9;
  printf("%-4s %d waves x %2d in flight (%3d KB/CU) region %4zu KB %-8s%s: %6.1f GB/s per CU = %5.1f B/clk (chip %5.2f TB/s)\n",
         MODE ? "LDS" : "VGPR", WAVES, D, D * WAVES, region >> 10, shared ? "SHARED" : "own", rot ? "+rot" : "    ", gbs, gbs / clk,
         gbs * 256 / 1e3);
  fflush(stdout);
}

int main() {
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  const double clk = pr.clockRate / 1e6;
  printf("%s: %d CUs, %.2f GHz\n", pr.name, pr.multiProcessorCount, clk);
  char* src;
  if (hipMalloc(&src, (size_t)1 << 30) != hipSuccess) return 1;
  hipMemset(src, 1, (size_t)1 << 30);
  for (size_t region : {(size_t)112 << 10, (size_t)784 << 10}) {
    for (int shared : {0, 1})
      for (int rot : {0, 1}) {
        if (!shared && rot) continue;
        if (!shared && region * 256 > ((size_t)1 << 30)) continue;
        run<0, 7, 8>(src, region, shared, rot, clk);
        run<0, 14, 8>(src, region, shared, rot, clk);
        run<0, 28, 8>(src, region, shared, rot, clk);
        run<0, 14, 4>(src, region, shared, rot, clk);
        run<1, 8, 8>(src, region, shared, rot, clk);
        run<1, 16, 8>(src, region, shared, rot, clk);
        run<1, 16, 4>(src, region, shared, rot, clk);
      }
  }
  return 0;
}
