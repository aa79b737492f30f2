// Micro-benchmark (dev tool): does data written by one kernel stay in the writing XCD's L2 for the next kernel?
// Kernel W: workgroup w writes region w.  Kernel R: workgroup w reads region (w + shift) % nwg.  Workgroups go to XCDs
// round-robin (w % 8), so shift 0 reads what the same XCD wrote, shift 1 what the neighbour XCD wrote, shift 8 the
// same XCD but another CU.  Regions together fit in the L2s (8 x 4 MB).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/xcd_affinity scripts/ubench/xcd_affinity.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void wr(f4* buf, int region_vec, float v) {
  f4* p = buf + (size_t)blockIdx.x * region_vec;
  for (int i = threadIdx.x; i < region_vec; i += 256) p[i] = (f4){v, v + 1, v + 2, v + 3};
}
__global__ __launch_bounds__(256) void rd(const f4* buf, int region_vec, int shift, float* out) {
  const int src = (blockIdx.x + shift) % gridDim.x;
  const f4* p = buf + (size_t)src * region_vec;
  f4 acc = {0, 0, 0, 0};
  for (int i = threadIdx.x; i < region_vec; i += 256) acc += p[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[0] = acc[0];
}

int main(int argc, char** argv) {
  const int only = argc > 1 ? atoi(argv[1]) : -1;
  const int nwg = 1024;
  for (int region_kb : {8, 16, 64}) {
    if (only >= 0 && region_kb != 16) continue;
    const int region_vec = region_kb * 1024 / 16;
    f4* buf;
    float* out;
    hipMalloc(&buf, (size_t)nwg * region_kb * 1024);
    hipMalloc(&out, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int shift : {0, 1, 8, 3, 0, 1}) {
      if (only >= 0 && shift != only) continue;
      float tot = 0.f;
      const int reps = 50;
      for (int r = 0; r < reps + 5; ++r) {
        hipLaunchKernelGGL(wr, dim3(nwg), dim3(256), 0, 0, buf, region_vec, (float)r);
        hipEventRecord(e0);
        hipLaunchKernelGGL(rd, dim3(nwg), dim3(256), 0, 0, buf, region_vec, shift, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (r >= 5) tot += ms;
      }
      printf("region %3d KB x %d WGs (%5.1f MB): read with shift %d: %6.2f us\n", region_kb, nwg, nwg * region_kb / 1024.0,
             shift, tot / reps * 1e3);
    }
    hipFree(buf);
    hipFree(out);
  }
  return 0;
}
