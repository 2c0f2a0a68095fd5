// ubench_fp64_overlap.hip -- do the FP64 matrix pipe and the FP64 vector ALU of one SIMD run side by side?
// (round 4, go / no-go for moving Jacobi cycles of the C2 kernel onto the matrix cores: DESIGN 4.1 (v))
// One workgroup per CU.  Modes:
//   0  4 waves (one per SIMD), all v_fma_f64            1  4 waves, all v_mfma_f64_16x16x4
//   2  8 waves (two per SIMD), all v_fma_f64            3  8 waves, all v_mfma_f64_16x16x4
//   4  8 waves: waves 0-3 v_fma_f64, waves 4-7 v_mfma   (a wave and its SIMD partner, w and w + 4, run different pipes)
//   5  as 4 with the vector waves running the Jacobi step pair's mix (150 FMA : 114 32-bit DPP moves : 95 other FP64 per 359)
//   6  8 waves all running that mix                      7  4 waves running that mix
// Prints per mode: ms, vector Tflop/s, matrix Tflop/s (chip-wide), and the per-role wave cycles (s_memtime, median over CUs).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_fp64_overlap.hip -o gpurun_out/ubench_fp64_overlap
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

template <int MIX>
__device__ __forceinline__ void valu_loop(int iters, double* out) {
  double x[16];
  const double a = 1.0000001, b = 1e-9;
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = 1.0 + threadIdx.x * 1e-6 + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {            // 10 x 16 = 160 FMAs per iteration (a step pair has 150)
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = __builtin_fma(x[i], a, b);
    }
    if (MIX) {
      // 114 32-bit DPP moves (row shifts) + 95 FP64 multiplies / adds: the rest of a Jacobi step pair
#pragma unroll
      for (int r = 0; r < 57; ++r) {
        int lo = __double2loint(x[r & 15]), hi = __double2hiint(x[r & 15]);
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);   // wave_shl:1
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
        x[r & 15] = __hiloint2double(hi, lo);
      }
#pragma unroll
      for (int r = 0; r < 95; ++r) x[r & 15] = x[r & 15] * a + ((r & 1) ? 0.0 : b) * 0.0 + 0.0 * x[(r + 1) & 15];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(x[i]));
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
  if (s == 12345.678) out[threadIdx.x] = s;
}

__device__ __forceinline__ void mfma_loop(int iters, double* out) {
  v4d acc[4];
  for (int t = 0; t < 4; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + threadIdx.x * 1e-6, b = 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    asm volatile("" : "+v"(a), "+v"(b));
  }
  double s = 0.0;
  for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  if (s == 12345.678) out[threadIdx.x] = s;
}

__global__ void __launch_bounds__(512) k(int mode, int it_v, int it_m, double* out, unsigned long long* cyc) {
  const int w = threadIdx.x >> 6;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  bool vec = true;
  if (mode == 1 || mode == 3) vec = false;
  if (mode == 4 || mode == 5) vec = w < 4;
  if (vec) {
    if (mode >= 5) valu_loop<1>(it_v, out);
    else valu_loop<0>(it_v, out);
  } else {
    mfma_loop(it_m, out);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}

int main() {
  int dev = 0;
  hipSetDevice(dev);
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, dev);
  const int ncu = pr.multiProcessorCount;
  double* out;
  unsigned long long* cyc;
  hipMalloc(&out, 512 * sizeof(double));
  hipMalloc(&cyc, ncu * 8 * sizeof(unsigned long long));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int it_v = 4000, it_m = 2500;   // per wave: 4000 x 160 FMAs (x 128 flop) / 2500 x 16 MFMAs (x 2048 flop): ~82 Mflop each
  printf("CUs %d, clock %d MHz\n", ncu, pr.clockRate / 1000);
  for (int mode = 0; mode < 8; ++mode) {
    const int threads = (mode == 0 || mode == 1 || mode == 7) ? 256 : 512;
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(cyc, 0, ncu * 8 * sizeof(unsigned long long));
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(ncu), dim3(threads), 0, 0, mode, it_v, it_m, out, cyc);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep < 2) continue;
      std::vector<unsigned long long> h(ncu * 8);
      hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
      std::vector<unsigned long long> cv, cm;
      const int nw = threads / 64;
      for (int b = 0; b < ncu; ++b)
        for (int w = 0; w < nw; ++w) {
          bool vec = !(mode == 1 || mode == 3);
          if (mode == 4 || mode == 5) vec = w < 4;
          (vec ? cv : cm).push_back(h[b * 8 + w]);
        }
      auto med = [](std::vector<unsigned long long>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return (double)v[v.size() / 2]; };
      const double nwv = (double)cv.size(), nwm = (double)cm.size();
      const double fl_v = nwv * it_v * 160.0 * 128.0, fl_m = nwm * it_m * 16.0 * 2048.0;
      const double ins_v = (mode >= 5) ? it_v * (160.0 + 114.0 + 95.0 * 2) : it_v * 160.0;
      printf("mode %d  %8.3f ms  vector %6.2f Tflop/s (FMA only)  matrix %6.2f Tflop/s  | wave cycles: vector %.0f (%.2f cyc/instr)  matrix %.0f (%.1f cyc/MFMA)\n",
             mode, ms, fl_v / ms / 1e9, fl_m / ms / 1e9, med(cv), cv.empty() ? 0.0 : med(cv) / ins_v, med(cm), cm.empty() ? 0.0 : med(cm) / (it_m * 16.0));
    }
  }
  return 0;
}
