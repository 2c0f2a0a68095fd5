// ubench_newbcast.hip -- v_fmac_f64_dpp with row_newbcast (gfx950: the one DPP control FP64 instructions take): correctness of the
// broadcast inside every row of 16 lanes, and its issue cost beside v_mfma_f64_16x16x4 -- the building block of the wave kernel's
// "narrow last block on the vector ALU" Gram (round 4): 4 x 4 broadcast-FMAs per 4-observation step in place of the 4 matrix
// instructions whose tiles are 3/4 padding at k = 50 (columns 48 .. 51 of 64).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_newbcast.hip -o gpurun_out/ubench_newbcast
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int A>
__device__ __forceinline__ void fmac_bcast(double& acc, const double ys, const double y) {
  if constexpr (A == 0) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(ys), "v"(y));
  if constexpr (A == 1) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(ys), "v"(y));
  if constexpr (A == 2) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(ys), "v"(y));
  if constexpr (A == 3) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(ys), "v"(y));
}
// mode 0: per iteration 10 MFMAs (the Gram step as it is); 1: 6 MFMAs + 16 broadcast-FMAs (the narrow block on the vector ALU);
// 2: 16 broadcast-FMAs alone; 3: 6 MFMAs alone
__global__ void __launch_bounds__(256) k(const double* in, double* out, int iters, int mode, unsigned long long* cyc) {
  const int l = threadIdx.x & 63;
  double y[4] = {in[l], in[64 + l], in[128 + l], in[l] * 0.5};
  double acc[16];
  v4d t[10];
  for (int i = 0; i < 16; ++i) acc[i] = 0.0;
  for (int i = 0; i < 10; ++i) t[i] = v4d{0.0, 0.0, 0.0, 0.0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const int nm = (mode == 0) ? 10 : (mode == 2 ? 0 : 6);
#pragma unroll
    for (int i = 0; i < 10; ++i)
      if (i < nm) t[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(y[i & 3], y[(i + 1) & 3], t[i], 0, 0, 0);
    if (mode == 1 || mode == 2) {
#pragma unroll
      for (int I = 0; I < 4; ++I) {
        fmac_bcast<0>(acc[4 * I + 0], y[3], y[I]);
        fmac_bcast<1>(acc[4 * I + 1], y[3], y[I]);
        fmac_bcast<2>(acc[4 * I + 2], y[3], y[I]);
        fmac_bcast<3>(acc[4 * I + 3], y[3], y[I]);
      }
    }
    asm volatile("" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i] * (i + 1);
  for (int i = 0; i < 10; ++i) s += t[i][0] + t[i][1] + t[i][2] + t[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[mode] = t1 - t0;
}
// issue order of the Gram step: (a) 10 tiles, one instruction each, per 4-observation step (the kernel's order: every instruction
// reads and writes a DIFFERENT accumulator quad) against (b) the same 40 instructions of four steps tile by tile (four in a row on
// ONE accumulator) and (c) two in a row
template <int ORDER>
__global__ void __launch_bounds__(256) korder(const double* in, double* out, int iters, unsigned long long* cyc) {
  const int l = threadIdx.x & 63;
  double y[4][4];
  for (int s = 0; s < 4; ++s)
    for (int i = 0; i < 4; ++i) y[s][i] = in[(16 * s + 5 * i + l) % 192];
  v4d t[10];
  for (int i = 0; i < 10; ++i) t[i] = v4d{0.0, 0.0, 0.0, 0.0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (ORDER == 0) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 10; ++i) t[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(y[s][i & 3], y[s][(i + 1) & 3], t[i], 0, 0, 0);
    } else if constexpr (ORDER == 1) {
#pragma unroll
      for (int i = 0; i < 10; ++i)
#pragma unroll
        for (int s = 0; s < 4; ++s) t[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(y[s][i & 3], y[s][(i + 1) & 3], t[i], 0, 0, 0);
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 10; ++i)
#pragma unroll
          for (int s = 2 * h; s < 2 * h + 2; ++s) t[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(y[s][i & 3], y[s][(i + 1) & 3], t[i], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(y[s][0]), "+v"(y[s][1]), "+v"(y[s][2]), "+v"(y[s][3]));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s_ = 0;
  for (int i = 0; i < 10; ++i) s_ += t[i][0] + t[i][1] + t[i][2] + t[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s_;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void check(const double* in, double* out) {
  const int l = threadIdx.x;
  double ys = in[l], y0 = in[64 + l], acc[4] = {0, 0, 0, 0};
  fmac_bcast<0>(acc[0], ys, y0);
  fmac_bcast<1>(acc[1], ys, y0);
  fmac_bcast<2>(acc[2], ys, y0);
  fmac_bcast<3>(acc[3], ys, y0);
  for (int a = 0; a < 4; ++a) out[a * 64 + l] = acc[a];
}
int main() {
  double *in, *out;
  unsigned long long* cyc;
  (void)hipMalloc(&in, 192 * 8);
  (void)hipMalloc(&out, 512 * 256 * 8);
  (void)hipMalloc(&cyc, 4 * 8);
  double h[192];
  for (int i = 0; i < 192; ++i) h[i] = 1.0 + 0.001 * i;
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, in, out);
  double o[256];
  (void)hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int a = 0; a < 4; ++a)
    for (int l = 0; l < 64; ++l)
      if (o[a * 64 + l] != h[16 * (l / 16) + a] * h[64 + l]) ++bad;
  printf("v_fmac_f64_dpp row_newbcast: %d of 256 values wrong\n", bad);
  const int iters = 20000;
  for (int mode = 0; mode < 4; ++mode) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, in, out, iters, mode, cyc);   // 2 workgroups per CU: 2 waves per SIMD
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
    }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[4];
    (void)hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    printf("mode %d: %.3f ms, %.1f wave cycles per iteration (two waves per SIMD)\n", mode, ms, (double)c[mode] / iters);
  }
  for (int order = 0; order < 3; ++order) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      if (order == 0) hipLaunchKernelGGL(korder<0>, dim3(512), dim3(256), 0, 0, in, out, iters / 4, cyc);
      if (order == 1) hipLaunchKernelGGL(korder<1>, dim3(512), dim3(256), 0, 0, in, out, iters / 4, cyc);
      if (order == 2) hipLaunchKernelGGL(korder<2>, dim3(512), dim3(256), 0, 0, in, out, iters / 4, cyc);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
    }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[4];
    (void)hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    printf("40 matrix instructions per iteration, order %d (0: step-major, 1: four in a row per tile, 2: two in a row): %.3f ms, %.1f wave cycles per instruction\n",
           order, ms, (double)c[0] / (iters / 4) / 40.0);
  }
  return bad != 0;
}
