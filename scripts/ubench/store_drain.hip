// Micro-benchmark (dev tool): how long do a workgroup's result stores take to be acknowledged
// (s_waitcnt vmcnt(0)) on MI355X, by store pattern and cache policy?  Mirrors the igemm epilogue:
// 256 workgroups x 4 waves, each wave issues NST stores then waits.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef _Float16 f16;
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));

// mode 0: epilogue pattern (lane (lc,lg): row lc, 4 halfs at col lg*4 + 16*j), 8 B per lane, rows ld halfs apart
// mode 1: coalesced rows: lane writes 16 B, 14 lanes per 224-B row segment
template <int MODE, int POLICY>
__global__ __launch_bounds__(256) void k(f16* y, int ld, int nst, unsigned long long* out, int delay) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lc = lane & 15, lg = lane >> 4;
  // busy wait so that every workgroup is resident and stores start together
  unsigned long long t = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t < (unsigned long long)delay) {}
  const unsigned long long t0 = __builtin_readcyclecounter();
  const long row0 = (long)blockIdx.x * 64 + wave * 16;
  if (MODE == 0) {
    f16x4 v = {(f16)1.f, (f16)2.f, (f16)3.f, (f16)4.f};
    for (int j = 0; j < nst; ++j) {
      f16* p = y + (row0 + lc) * ld + j * 16 + lg * 4;
      if (POLICY == 0) *(f16x4*)p = v;
      else if (POLICY == 1) __builtin_nontemporal_store(v, (f16x4*)p);
    }
  } else {
    f16x8 v = {(f16)1.f, (f16)2.f, (f16)3.f, (f16)4.f, (f16)1.f, (f16)2.f, (f16)3.f, (f16)4.f};
    // nst*16 cols per row = nst*32 B; 16 rows per wave -> nst*512 B per wave = nst*32 lanes x 16 B
    const int vec_per_row = nst * 2;  // 16-B vectors per row
    for (int q = lane; q < 16 * vec_per_row; q += 64) {
      const int r = q / vec_per_row, c = q - r * vec_per_row;
      f16* p = y + (row0 + r) * ld + c * 8;
      if (POLICY == 0) *(f16x8*)p = v;
      else if (POLICY == 1) __builtin_nontemporal_store(v, (f16x8*)p);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t2 = __builtin_readcyclecounter();
  if (lane == 0) {
    out[(blockIdx.x * 4 + wave) * 2 + 0] = t1 - t0;
    out[(blockIdx.x * 4 + wave) * 2 + 1] = t2 - t0;
  }
}

template <int MODE, int POLICY>
void run(const char* name, int blocks, int nst, int ld) {
  f16* y;
  unsigned long long* out;
  const size_t ybytes = (size_t)blocks * 64 * ld * 2 + 4096;
  hipMalloc(&y, ybytes);
  hipMalloc(&out, blocks * 8 * sizeof(unsigned long long));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  std::vector<unsigned long long> h(blocks * 8);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, POLICY>), dim3(blocks), dim3(256), 0, 0, y, ld, nst, out, 20000);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<unsigned long long> issue, done;
  for (int i = 0; i < blocks * 4; ++i) {
    issue.push_back(h[2 * i]);
    done.push_back(h[2 * i + 1]);
  }
  std::sort(issue.begin(), issue.end());
  std::sort(done.begin(), done.end());
  printf("%-34s blocks %4d nst %2d: issue med %6llu  acked med %6llu  p90 %6llu  max %6llu cycles   (kernel %.1f us)\n", name,
         blocks, nst, issue[issue.size() / 2], done[done.size() / 2], done[done.size() * 9 / 10], done.back(), ms * 1e3);
  hipFree(y);
  hipFree(out);
}

int main() {
  for (int blocks : {1, 32, 256, 1024}) {
    for (int nst : {1, 7, 28}) {
      run<0, 0>("epilogue 8B/lane plain", blocks, nst, 448);
      run<0, 1>("epilogue 8B/lane nt", blocks, nst, 448);
      run<1, 0>("coalesced 16B/lane plain", blocks, nst, 448);
      run<1, 1>("coalesced 16B/lane nt", blocks, nst, 448);
    }
  }
  return 0;
}
