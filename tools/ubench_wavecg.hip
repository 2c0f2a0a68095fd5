// Micro-benchmark (MI355X): the inner loop of a WAVE-level Lanczos / CG on a 64 x 64 matrix held as FP64 matrix-core
// A operands in registers (13 k-steps x 4 row blocks), 16 right-hand sides, B operand through a wave-private LDS block,
// two reductions per iteration inside the wave -- what an eigen-free variant of the k = 50 one-wave kernel would run.
// Question: microseconds per iteration and wave at 8 waves per CU (2 workgroups of 256 threads).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_wavecg.hip -o tools/ubench_wavecg
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int KS = 13;   // k-steps of 4 columns (k = 50 -> 52 columns)
__device__ __forceinline__ int rpos(int row) { return row; }
__global__ void __launch_bounds__(256, 2) wavecg(const double* __restrict__ Ain, double* out, int iters, int npoints) {
  extern __shared__ double smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, col = lane & 15, rq = lane >> 4;
  double* vb = smem + wv * 2048;   // [64 rows][16], two blocks would be 16 KB per wave
  double a[4][KS];
  double keep = 0.0;
  for (int p = 0; p < npoints; ++p) {
    // load A operands (stand-in for the Gram's accumulators)
#pragma unroll
    for (int R = 0; R < 4; ++R)
#pragma unroll
      for (int s = 0; s < KS; ++s) a[R][s] = Ain[((size_t)(blockIdx.x * 4 + wv) * 4 + R) * KS * 64 + s * 64 + lane];
    double v[4][4], vp[4][4];
#pragma unroll
    for (int R = 0; R < 4; ++R)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[R][r] = 1.0 / (1 + R * 16 + rq + 4 * r + col);
        vp[R][r] = 0.0;
      }
    double beta = 0.0, asum = 0.0;
    for (int j = 0; j < iters; ++j) {
      // v -> LDS (B operand layout [row][16])
#pragma unroll
      for (int R = 0; R < 4; ++R)
#pragma unroll
        for (int r = 0; r < 4; ++r) vb[(R * 16 + rq + 4 * r) * 16 + col] = v[R][r];
      d4 acc[4];
#pragma unroll
      for (int R = 0; R < 4; ++R) acc[R] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const double b = vb[(4 * s) * 16 + lane];
#pragma unroll
        for (int R = 0; R < 4; ++R) acc[R] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[R][s], b, acc[R], 0, 0, 0);
      }
      // w = A v - beta vp; alpha = v.w; w -= alpha v; beta' = |w|; v' = w / beta'
      double pa = 0.0;
#pragma unroll
      for (int R = 0; R < 4; ++R)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc[R][r] = fma(-beta, vp[R][r], acc[R][r]);
          pa = fma(v[R][r], acc[R][r], pa);
        }
      pa += __shfl_xor(pa, 16, 64);
      pa += __shfl_xor(pa, 32, 64);
      double pb = 0.0;
#pragma unroll
      for (int R = 0; R < 4; ++R)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc[R][r] = fma(-pa, v[R][r], acc[R][r]);
          pb = fma(acc[R][r], acc[R][r], pb);
        }
      pb += __shfl_xor(pb, 16, 64);
      pb += __shfl_xor(pb, 32, 64);
      const double nb = sqrt(pb), inb = 1.0 / nb;
#pragma unroll
      for (int R = 0; R < 4; ++R)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          vp[R][r] = v[R][r];
          v[R][r] = acc[R][r] * inb;
        }
      beta = nb;
      asum += pa;
    }
    keep += asum + v[1][2];
  }
  out[blockIdx.x * 256 + threadIdx.x] = keep;
}
int main() {
  const int nblk = 512, iters = 15, npoints = 40;
  double *A, *out;
  hipMalloc(&A, (size_t)nblk * 4 * 4 * KS * 64 * 8);
  hipMalloc(&out, (size_t)nblk * 256 * 8);
  std::vector<double> h((size_t)nblk * 4 * 4 * KS * 64);
  for (size_t i = 0; i < h.size(); ++i) h[i] = ((i * 2654435761u) % 1000) * 1e-3 + ((i % (KS * 64)) / 64 == 0 ? 50.0 : 0.0);
  hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&wavecg), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(wavecg, dim3(nblk), dim3(256), 65536, 0, A, out, iters, npoints);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // 512 workgroups = 2 per CU on 256 CUs: every wave runs npoints * iters iterations
    printf("rep %d: %.3f ms -> %.3f us per iteration and wave (8 waves per CU); %.2f M points/s at %d iterations per point\n", rep, ms,
           ms * 1e3 / (npoints * iters), nblk * 4.0 * npoints / (ms * 1e-3) / 1e6, iters);
  }
  return 0;
}
