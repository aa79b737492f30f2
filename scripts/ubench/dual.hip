// Micro-benchmark (dev tool): does a CU get MORE bytes per clock when it pulls its two GEMM operands over two different
// paths at once — weights global -> VGPR (register ring, `global_load_dwordx4`), activations global -> LDS (LDS-DMA,
// `global_load_lds_dwordx4`) — than over either path alone?  The implicit-GEMM K loops of the UNet's 3x3 convs run at
// the LDS-DMA rate (both operands through the ring: 29 B/clk/CU); a kernel that keeps the input patch in
// LDS and streams the weights into registers only pays if the two paths add up.
//   256 workgroups of 8 waves, one per CU, shader clocks read inside the kernel, everything cold (1 GB memset between
//   runs).  NB waves stream a weight region shared by half of the workgroups (451 KB: 112 columns x 2016 K), NA waves
//   DMA a per-workgroup activation region in 16 x 64-byte row pieces (the im2col pattern) into a 32 KB LDS ring.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dual scripts/ubench/dual.hip && /tmp/dual
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// waves [0, NB): register stream of `bkb` KB (shared region, half of the grid each); waves [NB, NB + NA): LDS-DMA of
// `akb` KB (own region), D instructions in flight per wave.  NT: nontemporal hint on the register stream.
// FRAG: lane -> byte mapping of a 1 KiB weight fragment.  false: lane * 16 (contiguous); true: (lane & 15) * 64 +
// (lane >> 4) * 16 — the MFMA operand order (row = lane & 15, k group = lane >> 4) read out of a [16 rows][64 B] run,
// i.e. what astat / mlp / xblock / halo do with the [K/32][n][32] packing: every quarter wave touches 16 different rows
template <int NB, int NA, int D, bool NT, bool FRAG = false>
__global__ __launch_bounds__(512) void dual(const char* wsrc, const char* asrc, int bkb, int akb, int row_bytes,
                                            unsigned long long* out, float* sink) {
  __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned long long t0 = __builtin_readcyclecounter();
  unsigned long long tb = 0, ta = 0;
  if (wave < NB) {
    const char* base = wsrc + (size_t)(blockIdx.x & 1) * ((size_t)bkb << 10) + (FRAG ? (lane & 15) * 64 + (lane >> 4) * 16 : lane * 16);
    const int nfr = bkb / NB;  // 1 KiB fragments of this wave
    f32x4 ring[D];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    unsigned f = wave;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      ring[d] = NT ? __builtin_nontemporal_load((const f32x4*)(base + (size_t)f * 1024)) : *(const f32x4*)(base + (size_t)f * 1024);
      f += NB;
    }
    for (int it = D; it < nfr; it += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        acc += ring[d];
        ring[d] = NT ? __builtin_nontemporal_load((const f32x4*)(base + (size_t)f * 1024)) : *(const f32x4*)(base + (size_t)f * 1024);
        f += NB;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) acc += ring[d];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.f) sink[0] = acc[0];
    tb = __builtin_readcyclecounter() - t0;
  } else if (wave < NB + NA) {
    const int aw = wave - NB;
    // a wave instruction = 16 rows x 64 B of a [rows][row_bytes] matrix; successive instructions walk along the row,
    // then to the next block of 16 rows (igemm_ws loader pattern)
    const char* base = asrc + (size_t)blockIdx.x * ((size_t)akb << 10);
    const int nrows = (akb << 10) / row_bytes;
    const int segs = row_bytes / 64;
    const int lane_off = (lane >> 2) * row_bytes + (lane & 3) * 16;
    const int ninst = akb / NA;
    int rb = aw, seg = 0;
    char* dst0 = lds + aw * (64 * 1024 / NA);
    int slot = 0;
    for (int i = 0; i < ninst; ++i) {
      const char* p = base + (size_t)(rb * 16) * row_bytes + seg * 64 + lane_off;
      __builtin_amdgcn_global_load_lds((glb_ptr)p, (lds_ptr)(dst0 + slot), 16, 0, 0);
      slot = (slot + 1024) & (64 * 1024 / NA - 1);
      if (++seg == segs) {
        seg = 0;
        rb += NA;
        if (rb * 16 >= nrows) rb = aw;
      }
      wait_vm<D - 1>();
    }
    wait_vm<0>();
    ta = __builtin_readcyclecounter() - t0;
  }
  if (lane == 0) {
    if (wave == 0 && NB) out[blockIdx.x * 2] = tb;
    if (wave == NB && NA) out[blockIdx.x * 2 + 1] = ta;
  }
}

template <int NB, int NA, int D, bool NT, bool FRAG = false>
void run(const char* tag, char* w, char* a, char* flush, int bkb, int akb, int row_bytes, unsigned long long* out) {
  std::vector<unsigned long long> h(512);
  double bb = 1e18, ba = 1e18;
  for (int rep = 0; rep < 4; ++rep) {
    hipMemset(flush, rep, (size_t)1 << 30);
    hipMemset(out, 0, 512 * 8);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((dual<NB, NA, D, NT, FRAG>), dim3(256), dim3(512), 0, 0, w, a, bkb, akb, row_bytes, out, (float*)nullptr);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), out, 512 * 8, hipMemcpyDeviceToHost);
    double sb = 0, sa = 0;
    for (int i = 0; i < 256; ++i) sb += (double)h[2 * i], sa += (double)h[2 * i + 1];
    bb = std::min(bb, sb / 256);
    ba = std::min(ba, sa / 256);
  }
  const double tot = std::max(bb, ba);
  printf("%-34s NB=%d NA=%d D=%2d nt=%d  W %4d KB in %7.0f clk (%5.1f B/clk)   A %4d KB in %7.0f clk (%5.1f B/clk)   both %5.1f B/clk/CU\n", tag,
         NB, NA, D, (int)NT, NB ? bkb : 0, NB ? bb : 0.0, NB ? bkb * 1024.0 / bb : 0.0, NA ? akb : 0, NA ? ba : 0.0,
         NA ? akb * 1024.0 / ba : 0.0, ((NB ? bkb : 0) + (NA ? akb : 0)) * 1024.0 / tot);
  fflush(stdout);
}

int main() {
  char *w, *a, *flush;
  unsigned long long* out;
  hipMalloc(&w, (size_t)64 << 20);
  hipMalloc(&a, (size_t)256 << 20);
  hipMalloc(&flush, (size_t)1 << 30);
  hipMalloc(&out, 512 * 8);
  hipMemset(w, 1, (size_t)64 << 20);
  hipMemset(a, 1, (size_t)256 << 20);
  // the 3x3 224 -> 224 conv at 32x32, 64 x 112 tile: 451 KB of weights, 258 KB of im2col rows / 61 KB of patch rows
  const int RB = 448;  // 224 channels fp16
  run<8, 0, 8, false, false>("weights, 8 waves, lane-contiguous", w, a, flush, 448, 0, RB, out);
  run<8, 0, 8, false, true>("weights, 8 waves, MFMA-order gather", w, a, flush, 448, 0, RB, out);
  run<8, 0, 16, false, false>("weights, 8 waves, lane-contiguous", w, a, flush, 448, 0, RB, out);
  run<8, 0, 16, false, true>("weights, 8 waves, MFMA-order gather", w, a, flush, 448, 0, RB, out);
  run<4, 0, 16, false, false>("weights, 4 waves, lane-contiguous", w, a, flush, 448, 0, RB, out);
  run<4, 0, 16, false, true>("weights, 4 waves, MFMA-order gather", w, a, flush, 448, 0, RB, out);
  if (getenv("DUAL_FRAG_ONLY")) return 0;
  run<4, 0, 8, false>("weights only, 4 waves", w, a, flush, 448, 0, RB, out);
  run<4, 0, 16, false>("weights only, 4 waves", w, a, flush, 448, 0, RB, out);
  run<8, 0, 8, false>("weights only, 8 waves", w, a, flush, 448, 0, RB, out);
  run<8, 0, 16, false>("weights only, 8 waves", w, a, flush, 448, 0, RB, out);
  run<8, 0, 8, true>("weights only, 8 waves", w, a, flush, 448, 0, RB, out);
  run<0, 4, 8, false>("im2col rows only (LDS-DMA)", w, a, flush, 0, 256, RB, out);
  run<0, 4, 16, false>("im2col rows only (LDS-DMA)", w, a, flush, 0, 256, RB, out);
  run<0, 8, 8, false>("im2col rows only (LDS-DMA) 8w", w, a, flush, 0, 256, RB, out);
  run<0, 4, 8, false>("weights AND rows by LDS-DMA (709 KB)", w, a, flush, 0, 704, 1024, out);
  run<4, 4, 8, false>("weights -> VGPR + rows -> LDS", w, a, flush, 448, 256, RB, out);
  run<4, 4, 16, false>("weights -> VGPR + rows -> LDS", w, a, flush, 448, 256, RB, out);
  run<4, 4, 8, true>("weights -> VGPR + rows -> LDS", w, a, flush, 448, 256, RB, out);
  run<4, 4, 8, false>("weights -> VGPR + patch -> LDS", w, a, flush, 448, 64, RB, out);
  run<4, 4, 16, false>("weights -> VGPR + patch -> LDS", w, a, flush, 448, 64, RB, out);
  run<6, 2, 8, false>("weights -> VGPR + patch -> LDS", w, a, flush, 448 - 448 % 6, 64, RB, out);
  run<6, 2, 16, false>("weights -> VGPR + patch -> LDS", w, a, flush, 448 - 448 % 6, 64, RB, out);
  // deep level: 14.4 MB of weights over 256 workgroups, each slice read by ONE workgroup (M = 128: one M tile)
  return 0;
}
