// micro-benchmark (MI355X): cost of broadcasting a double from a lane through v_readlane_b32 x2 + v_fma_f64 with
// an SGPR operand, against a pure v_fma_f64 loop.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_readlane.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ double rl(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

template <int MODE>
__global__ void __launch_bounds__(256, 2) k(double* out, int iters) {
  double a[16], u[8], o[16];
  for (int r = 0; r < 16; ++r) { a[r] = threadIdx.x * 1e-3 + r; o[r] = 0.0; }
  for (int i = 0; i < 8; ++i) u[i] = 1.0 + 1e-9 * (threadIdx.x + i);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (MODE == 0) o[r] = fma(rl(a[r], i), u[i], o[r]);
        else o[r] = fma(a[(r + i) & 15], u[i], o[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] += 1e-12 * o[r];
  }
  double s = 0;
  for (int r = 0; r < 16; ++r) s += o[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  double* d;
  hipMalloc(&d, 2048 * 256 * 8);
  const int iters = 2000;
  for (int mode = 0; mode < 2; ++mode) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(2048), dim3(256), 0, 0, d, iters);
      else hipLaunchKernelGGL(k<1>, dim3(2048), dim3(256), 0, 0, d, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 2048 wg * 4 waves / 1024 SIMDs = 8 waves per SIMD in sequence-ish (2 resident)
    double fmas_per_wave = (double)iters * 128;
    double ns_per_fma_slot = ms * 1e6 / (8.0 * fmas_per_wave);
    printf("mode %d: %.3f ms, %.3f ns per (fma%s) per SIMD\n", mode, ms, ns_per_fma_slot, mode == 0 ? " + 2 readlane" : "");
  }
  return 0;
}
