// Probe: do two 256-thread workgroups with 80 KB of dynamic LDS each share one CU on gfx950 (160 KB LDS)?
// Each workgroup spins ~2 ms; a grid of 2 x CUs takes ~2 ms if both fit on a CU, ~4 ms otherwise.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256, 2) spin(double* out, long ticks) {
  extern __shared__ double sm[];
  sm[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  double a = sm[(threadIdx.x * 7) & 255];
  while ((long)(__builtin_amdgcn_s_memtime() - t0) < ticks) a = a * 1.0000001 + 1e-9;
  if (a == 123.456) out[0] = a;
}
int main() {
  double* d;
  hipMalloc(reinterpret_cast<void**>(&d), 8);
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  for (int kb : {56, 64, 72, 78, 79, 80, 81}) {
    const size_t lds = (size_t)kb * 1024;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      std::printf("%d KB: hipFuncSetAttribute failed\n", kb);
      continue;
    }
    int occ = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin, 256, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    spin<<<2 * p.multiProcessorCount, 256, lds>>>(d, 100000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    spin<<<2 * p.multiProcessorCount, 256, lds>>>(d, 4000000);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::printf("%d KB/workgroup: occupancy API %d blocks/CU, grid of 2 x %d CUs took %.3f ms (%s)\n", kb, occ,
                p.multiProcessorCount, ms, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
