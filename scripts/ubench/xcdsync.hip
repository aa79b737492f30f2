// Micro-benchmark (dev tool) behind DESIGN.md 12a: the two prices that decide whether the deep UNet levels can run as a
// per-XCD persistent engine (sample b on XCD b, the 32 CUs of an XCD split a layer's output channels, every XCD streams
// every weight) instead of chip-wide launches:
//   A  an XCD-LOCAL barrier + hand-off: 32 workgroups that share one L2 (found by HW_REG_XCC_ID, not by blockIdx) —
//      plain stores, s_waitcnt vmcnt(0), a workgroup-scope atomic in THAT L2, sc1 (L1-bypassing) polls and payload
//      loads; every word of every hand-off checked for staleness; compared with the same episode on agent-scope
//      atomics + release/acquire fences (what a chip-wide barrier pays);
//   B  a 14.45 MB weight (one 3x3 896->896 conv) streamed global -> VGPR by all 256 CUs, (i) partitioned: CU j reads
//      slice j of 256 (what today's split-K launches fetch), (ii) replicated per XCD: CU r of every XCD reads slice r
//      of 32 (eight XCDs pull the same bytes through the fabric / MALL), over a sequence of different weights larger
//      than the 256 MB Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/xcdsync scripts/ubench/xcdsync.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u32;
typedef __attribute__((address_space(1))) u32 gu32;

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e = (x);                                                               \
    if (e != hipSuccess) {                                                            \
      printf("%s failed: %s\n", #x, hipGetErrorString(e));                            \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

__device__ __forceinline__ int xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15;
}
__device__ __forceinline__ u32 load_sc1(const u32* p) {
  u32 v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ f4 load4_sc1(const f4* p) {
  f4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}

struct SyncArgs {
  u32* arrive;   // [8] device-scope arrival counters (rank within the XCD)
  u32* cnt;      // [8 * 32] one barrier counter per XCD (own 128-B line)
  f4* payload;   // [8][32][PV] f4: slot of (xcd, rank)
  u32* stale;    // [1] mismatching words seen
  u32* tmo;      // [1] timeouts
  long long* cyc;  // [256] cycles per workgroup for the timed rounds
  u32* census;   // [8] workgroups seen per XCD, [8..15] blockIdx % 8 != xcc count
  int rounds, pv, mode;  // mode 0: L2-local atomics + sc1 loads; 1: agent atomics + release / acquire fences + plain loads
  int do_payload;
};

// one XCD-local barrier episode; returns false on timeout
__device__ __forceinline__ bool xcd_barrier(u32* cnt, u32 target, int mode, u32* tmo) {
  if (mode == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its stores have reached the L2
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // performed in this XCD's L2
      u32 spins = 0;
      while (load_sc1(cnt) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 17)) {
          atomicAdd(tmo, 1u);
          break;
        }
      }
    }
    __syncthreads();
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      u32 spins = 0;
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 17)) {
          atomicAdd(tmo, 1u);
          break;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  return true;
}

__global__ __launch_bounds__(512) void sync_kernel(SyncArgs a) {
  extern __shared__ char lds[];  // (100 KB requested: one workgroup per CU)
  __shared__ int s_rank;
  const int xcc = xcc_id();
  if (threadIdx.x == 0) {
    s_rank = (int)atomicAdd(&a.arrive[xcc], 1u);
    atomicAdd(&a.census[xcc], 1u);
    if ((int)(blockIdx.x & 7) != xcc) atomicAdd(&a.census[8 + xcc], 1u);
  }
  __syncthreads();
  const int rank = s_rank;
  if (rank >= 32) return;  // (a placement this protocol does not cover: reported by the census)
  u32* cnt = a.cnt + xcc * 32;
  f4* mine = a.payload + ((size_t)xcc * 32 + rank) * a.pv;
  const f4* xcd0 = a.payload + (size_t)xcc * 32 * a.pv;
  u32 bad = 0;
  // all 32 workgroups of the XCD present before the clock starts
  xcd_barrier(cnt, 32u, a.mode, a.tmo);
  const long long t0 = __builtin_readcyclecounter();
  for (int r = 1; r <= a.rounds; ++r) {
    if (a.do_payload) {
      const float v = (float)(r * 64 + rank);
      for (int i = threadIdx.x; i < a.pv; i += 512) mine[i] = (f4){v, v, v, v};
    }
    xcd_barrier(cnt, 32u * (u32)(r + 1), a.mode, a.tmo);
    if (__hip_atomic_load(a.tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;  // (a stuck protocol ends the run)
    if (a.do_payload) {
      // every workgroup reads every slot of its XCD (32 * pv * 16 B) and checks every word
      const int total = 32 * a.pv;
      for (int i = threadIdx.x; i < total; i += 512) {
        f4 v;
        if (a.mode == 0) {
          v = load4_sc1(xcd0 + i);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
          v = xcd0[i];
        }
        const float want = (float)(r * 64 + i / a.pv);
        bad += (v[0] != want) + (v[1] != want) + (v[2] != want) + (v[3] != want);
      }
      // (a second barrier so that nobody overwrites a slot still being read: part of a real phase pair too)
      xcd_barrier(cnt + 16, 32u * (u32)r, a.mode, a.tmo);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) a.cyc[xcc * 32 + rank] = t1 - t0;
  if (bad) atomicAdd(a.stale, bad);
  if (lds[threadIdx.x] == 77 && a.rounds < 0) a.stale[0] = 1;  // (keeps the LDS request alive)
}

// ---------------------------------------------------------------- B: weight stream, partitioned vs replicated per XCD
struct StreamArgs {
  const char* w;       // nbuf weights of wbytes each, back to back
  size_t wbytes;       // bytes of one weight (multiple of 256 KiB)
  int nbuf, replicated;
  u32* arrive;
  long long* cyc;
  float* sink;
};

template <int D>
__global__ __launch_bounds__(512) void stream_kernel(StreamArgs a) {
  extern __shared__ char lds[];
  __shared__ int s_rank;
  const int xcc = xcc_id();
  if (threadIdx.x == 0) s_rank = (int)atomicAdd(&a.arrive[xcc], 1u);
  __syncthreads();
  const int rank = s_rank & 31;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t slice = a.replicated ? a.wbytes / 32 : a.wbytes / 256;
  const size_t off0 = a.replicated ? (size_t)rank * slice : (size_t)(xcc * 32 + rank) * slice;
  const int nfr = (int)(slice / 1024);  // 1 KiB per wave instruction; wave w takes fragments w, w + 8, ...
  f4 acc = {0, 0, 0, 0};
  const long long t0 = __builtin_readcyclecounter();
  for (int b = 0; b < a.nbuf; ++b) {
    const char* base = a.w + (size_t)b * a.wbytes + off0 + lane * 16;
    f4 ring[D];
    int f = wave;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      ring[d] = *(const f4*)(base + (size_t)(f < nfr ? f : wave) * 1024);
      f += 8;
    }
    for (; f - 8 * D < nfr; ) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        acc += ring[d];
        ring[d] = *(const f4*)(base + (size_t)(f < nfr ? f : wave) * 1024);
        f += 8;
      }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) acc += ring[d];
  }
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) a.cyc[xcc * 32 + rank] = t1 - t0;
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.f) a.sink[0] = acc[0];
  if (lds[threadIdx.x] == 77 && a.nbuf < 0) a.sink[1] = 1;
}

int main() {
  hipDeviceProp_t pr;
  CK(hipGetDeviceProperties(&pr, 0));
  const double ghz = pr.clockRate / 1e6;
  printf("%s: %d CUs, %.2f GHz (cycle counter: s_memtime, 100 MHz constant clock if the numbers say so)\n", pr.name, pr.multiProcessorCount, ghz);
  const int LDS = 100 * 1024;
  CK(hipFuncSetAttribute((const void*)sync_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  u32 *arrive, *cnt, *stale, *tmo, *census;
  long long* cyc;
  f4* payload;
  const int PVMAX = 4096;  // 64 KB per slot
  CK(hipMalloc(&arrive, 64));
  CK(hipMalloc(&cnt, 8 * 32 * 4));
  CK(hipMalloc(&stale, 4));
  CK(hipMalloc(&tmo, 4));
  CK(hipMalloc(&census, 64));
  CK(hipMalloc(&cyc, 256 * 8));
  CK(hipMalloc(&payload, (size_t)256 * PVMAX * 16));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<long long> hc(256);
  for (int mode : {0, 1})
    for (int pv : {0, 16, 64, 256, 1024}) {  // 0 = barrier only; else slot bytes = pv * 16 (256 B .. 16 KB; x32 read per WG)
      const int rounds = 200;
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(arrive, 0, 64));
        CK(hipMemset(cnt, 0, 8 * 32 * 4));
        CK(hipMemset(stale, 0, 4));
        CK(hipMemset(tmo, 0, 4));
        CK(hipMemset(census, 0, 64));
        CK(hipMemset(payload, 0xff, (size_t)256 * PVMAX * 16));
        SyncArgs a{arrive, cnt, payload, stale, tmo, cyc, census, rounds, pv ? pv : 1, mode, pv > 0};
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(sync_kernel, dim3(256), dim3(512), LDS, 0, a);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 0) continue;
        u32 hs, ht, hcen[16];
        CK(hipMemcpy(&hs, stale, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&ht, tmo, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hcen, census, 64, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hc.data(), cyc, 256 * 8, hipMemcpyDeviceToHost));
        long long mx = 0;
        for (auto c : hc) mx = c > mx ? c : mx;
        printf("A mode %d (%s) slot %5d B (each WG reads %4d KB per round): %7.3f us per round by events, %8.1f counter ticks per round; stale words %u, timeouts %u; census",
               mode, mode ? "agent atomics + fences, plain loads" : "L2-local atomics, sc1 loads", pv * 16, pv * 16 * 32 / 1024,
               ms * 1e3 / rounds, (double)mx / rounds, hs, ht);
        for (int i = 0; i < 8; ++i) printf(" %u", hcen[i]);
        u32 off = 0;
        for (int i = 8; i < 16; ++i) off += hcen[i];
        printf(" (blockIdx%%8 != xcc: %u)\n", off);
        fflush(stdout);
      }
    }

  // ---- B
  const size_t wbytes = (size_t)14680064;  // 14 MiB ~ 9 * 896 * 896 * 2 B (14.45 MB), a multiple of 256 KiB
  const int nbuf = 24;                      // 336 MiB > Infinity Cache
  char* w;
  float* sink;
  CK(hipMalloc(&w, wbytes * nbuf));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(w, 1, wbytes * nbuf));
  CK(hipFuncSetAttribute((const void*)stream_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  CK(hipFuncSetAttribute((const void*)stream_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  for (int D : {8, 16})
    for (int repl : {0, 1, 0, 1}) {
      CK(hipMemset(arrive, 0, 64));
      StreamArgs a{w, wbytes, nbuf, repl, arrive, cyc, sink};
      CK(hipEventRecord(e0));
      if (D == 8)
        hipLaunchKernelGGL(stream_kernel<8>, dim3(256), dim3(512), LDS, 0, a);
      else
        hipLaunchKernelGGL(stream_kernel<16>, dim3(256), dim3(512), LDS, 0, a);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double us_buf = ms * 1e3 / nbuf;
      const double per_cu = (repl ? wbytes / 32.0 : wbytes / 256.0);
      printf("B ring %2d x 8 waves, %s: %7.2f us per 14 MiB weight (%d weights, %.0f MiB); per CU %6.1f KB per weight = %5.1f GB/s = %4.1f B/clk; bytes into CUs %6.2f TB/s, distinct bytes %5.2f TB/s\n",
             D, repl ? "REPLICATED per XCD (CU r of each XCD reads slice r of 32)" : "partitioned (CU j reads slice j of 256)      ",
             us_buf, nbuf, wbytes * nbuf / 1048576.0, per_cu / 1024, per_cu / us_buf / 1e3, per_cu / us_buf / 1e3 / ghz,
             per_cu * 256 / us_buf / 1e6, wbytes / us_buf / 1e6);
      fflush(stdout);
    }
  return 0;
}
