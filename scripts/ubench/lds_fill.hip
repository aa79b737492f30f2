// Micro-benchmark (dev tool): per-CU global -> LDS fill rate of the LDS-DMA path (global_load_lds_dwordx4) on MI355X,
// by access pattern, bytes in flight and where the data lives (L2 / Infinity Cache / HBM).  One workgroup per CU
// (128 KB of LDS), 4 loader waves like igemm_ws_kernel / pconv_kernel.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/lds_fill scripts/ubench/lds_fill.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// The 4 waves sweep a [rows x row_bytes] matrix the way the igemm loader does: a wave-instruction covers RPI rows x SEG
// bytes (RPI * SEG = 1 KB), successive instructions advance along the row (next K chunk), then to the next row block.
// PAT 0: SEG = 1 KB contiguous (packed weights), 1: 16 rows x 64 B (32-channel chunk), 2: 8 rows x 128 B (64-channel
// chunk), 3: 4 rows x 256 B.  U instructions per group, D groups in flight per wave.
template <int PAT, int U, int D>
__global__ __launch_bounds__(256) void fill(const char* src, size_t region, int row_bytes, int groups, float* sink) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* base = src + (size_t)blockIdx.x * region;
  constexpr int SEG = PAT == 0 ? 1024 : (PAT == 1 ? 64 : (PAT == 2 ? 128 : 256));
  constexpr int RPI = 1024 / SEG;
  constexpr int LPR = SEG / 16;  // lanes per row
  if (PAT == 0) row_bytes = 1024;
  const int nrows = (int)(region / row_bytes);
  const int segs_per_row = row_bytes / SEG;
  const int lane_off = (lane / LPR) * row_bytes + (lane % LPR) * 16;
  // wave w owns row blocks w, w+4, ... of RPI rows; the cursor is advanced with adds only (the loop must stay far
  // from the wave's instruction-issue limit: ~1 instruction per 4-5 cycles)
  const int nrb_w = __builtin_amdgcn_readfirstlane((nrows / RPI - wave + 3) / 4);
  const char* const p0 = base + (size_t)wave * RPI * row_bytes + lane_off;
  const long jump = (long)4 * RPI * row_bytes - row_bytes;
  const char* p = p0;
  int seg = 0, rbc = 0;
  float acc = 0.f;
  char* dst0 = lds + wave * (32 * 1024);
  int slot = 0;
  for (int g = 0; g < groups; ++g) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      __builtin_amdgcn_global_load_lds((glb_ptr)p, (lds_ptr)(dst0 + slot), 16, 0, 0);
      slot = (slot + 1024) & (32 * 1024 - 1);
      p += SEG;
      if (++seg == segs_per_row) {
        seg = 0;
        p += jump;
        if (++rbc == nrb_w) {
          rbc = 0;
          p = p0;
        }
      }
    }
    wait_vm<(D - 1) * U>();
  }
  wait_vm<0>();
  __syncthreads();
  if (sink && threadIdx.x == 0) acc += ((float*)lds)[1];
  if (sink && acc == 123.f) sink[0] = acc;
}

static int g_wgs = 256;
template <int PAT, int U, int D>
void run(const char* name, const char* src, size_t region, int row_stride, double clk_ghz, const char* where) {
  const int wgs = g_wgs;
  const size_t per_wave_instr = 8192;  // 8 MB per wave, 32 MB per CU
  const int groups = (int)(per_wave_instr / U);
  hipFuncSetAttribute((const void*)fill<PAT, U, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((fill<PAT, U, D>), dim3(wgs), dim3(256), 128 * 1024, 0, src, region, row_stride, groups, nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes_cu = (double)groups * U * 4 * 1024;
  const double gbs = bytes_cu / (ms * 1e-3) / 1e9;
  printf("%3d WGs %-28s %-4s in flight %3d KB/CU: %7.1f GB/s per CU = %5.1f B/clk  (chip %5.2f TB/s)\n", wgs, name, where, D * U * 4,
         gbs, gbs / clk_ghz, gbs * wgs / 1e3);
  fflush(stdout);
}

template <int PAT>
void sweep(const char* name, const char* src, size_t region, int rs, double clk, const char* where) {
  run<PAT, 4, 1>(name, src, region, rs, clk, where);
  run<PAT, 4, 2>(name, src, region, rs, clk, where);
  run<PAT, 4, 4>(name, src, region, rs, clk, where);
  run<PAT, 8, 4>(name, src, region, rs, clk, where);
  run<PAT, 15, 4>(name, src, region, rs, clk, where);
}

int main() {
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  const double clk = pr.clockRate / 1e6;
  printf("%s: %d CUs, %.2f GHz\n", pr.name, pr.multiProcessorCount, clk);
  const size_t total = (size_t)2 << 30;
  char* src;
  if (hipMalloc(&src, total) != hipSuccess) return 1;
  hipMemset(src, 1, total);
  struct {
    const char* where;
    size_t region;
  } lv[] = {{"L2", 28 << 10}, {"MALL", 448 << 10}, {"HBM", 7 << 20}};
  for (int wgs : {256, 32})
  for (auto& l : lv) {
    g_wgs = wgs;
    sweep<0>("1 KB contiguous", src, l.region, 0, clk, l.where);
    sweep<1>("16 rows x 64 B, rows 1792 B", src, l.region, 1792, clk, l.where);
    sweep<2>("8 rows x 128 B, rows 1792 B", src, l.region, 1792, clk, l.where);
    sweep<3>("4 rows x 256 B, rows 1792 B", src, l.region, 1792, clk, l.where);
    sweep<1>("16 rows x 64 B, rows 448 B", src, l.region, 448, clk, l.where);
    sweep<1>("16 rows x 64 B, rows 64 B", src, l.region, 64, clk, l.where);
  }
  return 0;
}
