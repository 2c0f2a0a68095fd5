// letkf_post.hip -- what das_letkf / letkf.f90 do after the loop:
//   monit_dep            scale/common/common_obs_scale.f90:1851-1895   departure statistics per observation element
//   additive inflation   scale/letkf/letkf_tools.f90:804-929           anal += (add - mean(add)) * INFL_ADD * w [* q-bar]
//   addinfl_weight       :813-838 (INFL_ADD_REF_ONLY)                  exp(-d^2/2) of the nearest reflectivity obs
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "letkf_device.h"

namespace letkf {

namespace {

constexpr int kMaxId = 32;
constexpr int kIdT = 3073, kIdTv = 3074, kIdRef = 4001, kIdRe0 = 4004;   // common_obs_scale.f90:50-51, 65-66

struct UidTable {
  int n;
  int id[kMaxId];
};

// Two-level, fixed-order reduction (no floating-point atomics: the result does not depend on scheduling).
// part: [gridDim.x][nid][2] doubles, cnt: [gridDim.x][nid] ints.
__global__ void __launch_bounds__(256) monit_partial_kernel(const UidTable T, const long nn, const int* __restrict__ elm,
                                                            const double* __restrict__ dep, const int* __restrict__ qc,
                                                            double* __restrict__ part, int* __restrict__ cnt) {
  __shared__ double s_sum[256], s_sq[256];
  __shared__ int s_n[256];
  const long per = (nn + gridDim.x - 1) / gridDim.x;
  const long lo = (long)blockIdx.x * per, hi = min(nn, lo + per);
  for (int u = 0; u < T.n; ++u) {
    double a = 0.0, b = 0.0;
    int c = 0;
    for (long n = lo + threadIdx.x; n < hi; n += 256) {
      if (qc[n] != 0) continue;
      int e = elm[n];
      if (e == kIdTv) e = kIdT;          // Tv counted as T, RE0 as REF (:1871-1876)
      if (e == kIdRe0) e = kIdRef;
      if (e == T.id[u]) {
        const double d = dep[n];
        a += d;
        b = fma(d, d, b);
        ++c;
      }
    }
    s_sum[threadIdx.x] = a;
    s_sq[threadIdx.x] = b;
    s_n[threadIdx.x] = c;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) {
        s_sum[threadIdx.x] += s_sum[threadIdx.x + w];
        s_sq[threadIdx.x] += s_sq[threadIdx.x + w];
        s_n[threadIdx.x] += s_n[threadIdx.x + w];
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      part[((long)blockIdx.x * T.n + u) * 2] = s_sum[0];
      part[((long)blockIdx.x * T.n + u) * 2 + 1] = s_sq[0];
      cnt[(long)blockIdx.x * T.n + u] = s_n[0];
    }
    __syncthreads();
  }
}

__global__ void monit_final_kernel(const int nid, const int nblk, const double* __restrict__ part,
                                   const int* __restrict__ cnt, int* __restrict__ nobs, double* __restrict__ bias,
                                   double* __restrict__ rmse) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= nid) return;
  double a = 0.0, b = 0.0;
  long c = 0;
  for (int k = 0; k < nblk; ++k) {
    a += part[((long)k * nid + u) * 2];
    b += part[((long)k * nid + u) * 2 + 1];
    c += cnt[(long)k * nid + u];
  }
  nobs[u] = (int)c;
  if (c == 0) {
    bias[u] = -9.99e33;                  // undef (:1885-1887)
    rmse[u] = -9.99e33;
  } else {
    bias[u] = a / (double)c;
    rmse[u] = sqrt(b / (double)c);
  }
}

// anal(p, m, v) += add(p, mshuf, v) * infl_add * w(ij) [* qscale(p, v)], p = ij + nij1*lev  (:884-913); `add` already
// holds perturbations (mean removed by letkf_ens_to_perturbations_dev, :862-871)
__global__ void additive_kernel(const int k, const int nv, const long npts, const long nij1, double* __restrict__ anal,
                                const double* __restrict__ add, const long sp, const long sm, const long sv,
                                const double infl_add, const double* __restrict__ weight,
                                const double* __restrict__ qmean, const long q_sp, const long q_sv, const int iv_q_first,
                                const int iv_q_last, const int* __restrict__ ishuf) {
  const long tot = npts * (long)k * nv;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < tot; t += (long)gridDim.x * blockDim.x) {
    const long p = t % npts;
    const int m = (int)((t / npts) % k);
    const int v = (int)(t / (npts * k));
    const int ms = ishuf ? ishuf[m] : m;
    double x = add[p * sp + ms * sm + v * sv] * infl_add;
    if (weight) x = x * weight[p % nij1];
    if (qmean && v >= iv_q_first && v <= iv_q_last) x = x * qmean[p * q_sp + v * q_sv];
    anal[p * sp + m * sm + v * sv] += x;
  }
}

// addinfl_weight (:813-838): nearest reflectivity observation, brute force over the ctype's rows as in the reference;
// one wave per horizontal point, min by wave reduction (min is order-independent: bit-identical)
__global__ void __launch_bounds__(256) addinfl_weight_kernel(const long nij1, const double* __restrict__ rig,
                                                             const double* __restrict__ rjg, const long nob,
                                                             const double* __restrict__ ob_ri,
                                                             const double* __restrict__ ob_rj, const double dx,
                                                             const double dy, const double hori_loc,
                                                             const double cut2, double* __restrict__ w) {
  const int lane = threadIdx.x & 63;
  const long w0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
  for (long ij = w0; ij < nij1; ij += nw) {
    const double ri = rig[ij], rj = rjg[ij];
    double best = 1.0e33;
    for (long o = lane; o < nob; o += 64) {
      const double rdx = (ri - ob_ri[o]) * dx, rdy = (rj - ob_rj[o]) * dy;
      const double r2 = rdx * rdx + rdy * rdy;
      best = r2 < best ? r2 : best;
    }
    for (int msk = 1; msk < 64; msk <<= 1) best = fmin(best, __shfl_xor(best, msk, 64));
    if (lane == 0) {
      const double d = best / (hori_loc * hori_loc);
      w[ij] = d <= cut2 ? exp(-0.5 * d) : 0.0;
    }
  }
}

inline int grid_for(long n, int block, int num_cu) {
  long g = (n + block - 1) / block;
  const long cap = (long)num_cu * 16;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

size_t monit_scratch_bytes(int nid, int num_cu) { return (size_t)num_cu * 4 * nid * (2 * sizeof(double) + sizeof(int)) + 256; }

hipError_t launch_monit_dep(int nid, const int* elem_uid, long nn, const int* elm, const double* dep, const int* qc,
                            int* nobs, double* bias, double* rmse, void* scratch, int num_cu, hipStream_t st) {
  if (nid < 1 || nid > kMaxId) return hipErrorInvalidValue;
  UidTable T;
  T.n = nid;
  for (int i = 0; i < nid; ++i) T.id[i] = elem_uid[i];
  for (int i = nid; i < kMaxId; ++i) T.id[i] = -1;
  const int nblk = num_cu * 4;
  double* part = static_cast<double*>(scratch);
  int* cnt = reinterpret_cast<int*>(part + (size_t)nblk * nid * 2);
  hipLaunchKernelGGL(monit_partial_kernel, dim3(nblk), dim3(256), 0, st, T, nn, elm, dep, qc, part, cnt);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(monit_final_kernel, dim3(1), dim3(64), 0, st, nid, nblk, part, cnt, nobs, bias, rmse);
  return hipGetLastError();
}

hipError_t launch_additive(int k, int nv, long npts, long nij1, double* anal, const double* add, long sp, long sm,
                           long sv, double infl_add, const double* weight, const double* qmean, long q_sp, long q_sv,
                           int iv_q_first, int iv_q_last, const int* ishuf, int num_cu, hipStream_t st) {
  const long tot = npts * (long)k * nv;
  if (tot <= 0) return hipSuccess;
  hipLaunchKernelGGL(additive_kernel, dim3(grid_for(tot, 256, num_cu)), dim3(256), 0, st, k, nv, npts, nij1, anal, add,
                     sp, sm, sv, infl_add, weight, qmean, q_sp, q_sv, iv_q_first, iv_q_last, ishuf);
  return hipGetLastError();
}

hipError_t launch_addinfl_weight(long nij1, const double* rig, const double* rjg, long nob, const double* ob_ri,
                                 const double* ob_rj, double dx, double dy, double hori_loc, double cut2, double* w,
                                 int num_cu, hipStream_t st) {
  if (nij1 <= 0) return hipSuccess;
  hipLaunchKernelGGL(addinfl_weight_kernel, dim3(grid_for(nij1 * 64, 256, num_cu)), dim3(256), 0, st, nij1, rig, rjg,
                     nob, ob_ri, ob_rj, dx, dy, hori_loc, cut2, w);
  return hipGetLastError();
}

}  // namespace letkf
