// letkf_staged.hip -- the loop body of das_letkf (scale/letkf/letkf_tools.f90:313-527) and letkf_core
// (common/common_letkf.f90:52-257) as THREE kernels over a batch of grid points, for ensemble sizes beyond the
// one-/two-wave register kernel (letkf_wave.hip: k <= 100):
//
//   1. letkf_stage_gram_kernel   local-obs gather + weights + the symmetric matrix M of the point's eigenproblem
//   2. letkf_eig_wg_kernel (letkf_eig.hip, orders <= 208) / letkf_eig_block_kernel (letkf_kernels.hip, larger)
//   3. letkf_stage_apply_kernel  spectral functions of M applied to [dep | dep_det | x'_1 .. x'_nv], relaxation,
//                                total weight, analysis members, optional T / Pa
//
// with the matrices in a per-point workspace slab that lives in L2 / Infinity Cache between the stages (layout in
// letkf_device.h).  Splitting lets stage 2 own a whole CU's registers for one matrix.
//
// Two formulations of the same letkf_core, chosen per point:
//   PRIMAL (n >= k): M = Z^T Z + c I (k x k), Z = diag(sqrt w) Y the weighted obs-space perturbations (n x k),
//     c = (k-1)/rho: exactly common_letkf.f90:111-147.  M = V L V^T,  Pa = V L^-1 V^T,  T = V sqrt((k-1)/L) V^T.
//   DUAL (n < k): A = c I + Z^T Z is a rank-n update of c I, so every function of A follows from the n x n matrix
//     M = Z Z^T + c I = U L U^T (same non-trivial spectrum L = c + S):
//       f(A) = f(c) I + Z^T U diag((f(L) - f(c)) / S) U^T Z
//     with the divided differences in closed form (no cancellation, valid down to S = 0):
//       Pa = A^-1            : f(c) = 1/c,          g = -1 / (c L)
//       T  = sqrt(k-1) A^-1/2: f(c) = sqrt(rho),    g = -sqrt(k-1) / (sqrt(c) sqrt(L) (sqrt(c) + sqrt(L)))
//     and w-bar = Pa Z^T d = Z^T (c I + Z Z^T)^-1 d = Z^T U L^-1 U^T d (push-through).  The eigenproblem shrinks from
//     k to n (BASELINE configs[2]: 320 -> ~200, configs[4]: 1000 -> ~200) and M's conditioning is that of A.
//     No observation at all (common_letkf.f90:89-107) is the dual form with n = 0: T = sqrt(rho) I, Pa = I / c.
//
// Everything below the eigen-solve is small dense algebra with 13 right-hand sides, written as two patterns:
// wave-per-column inner products (lanes along the rows, coalesced) and thread-per-row linear combinations.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "letkf_staged_dev.h"

namespace letkf {

using namespace staged_dev;

namespace {

constexpr int kGT = 4;         // Gram register tile
constexpr int kGMaxT = 3;      // tiles per thread and pass
constexpr int kGBlock = 512;
#ifndef STAGE_APPLY_BLOCK
#define STAGE_APPLY_BLOCK 512
#endif
#ifndef STAGE_APPLY_MINWG
#define STAGE_APPLY_MINWG 1
#endif
#ifndef STAGE_APPLY_LDSCAP
#define STAGE_APPLY_LDSCAP (150 * 1024)
#endif
constexpr int kABlock = STAGE_APPLY_BLOCK;

}  // namespace

// hist: doubles of the eigen-free stage's residual history (stage_krylov_hist_doubles, 0 = none)
long staged_ws_per_point(int k, int nv, int kkout, long hist) {
  const long ldg = staged_ld(k), nb = nv + 2;
  long w = (long)(k + 1) * ldg + 2L * k + 16 + (long)nv * k + 4L * nb * k + 2L * nb;
  if (kkout) w += (long)k * ldg;
  w = (w + 15) & ~15L;                                 // (H starts on a 128-byte boundary, slab_of)
  return w + ((hist + 15) & ~15L);
}

// ------------------------------------------------------------------------------------------------ stage 1
// LDS: tile [tn][ld] | sw [max(k, ...)] (dual: sqrt(w_i), i < n < k) | wrow [3 tn]
__global__ void __launch_bounds__(kGBlock) letkf_stage_gram_kernel(const StagedArgs S, const int tn, const int ldmax) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const PointArgs& A = S.A;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int k = A.k;
  const double km1 = (double)(k - 1);
  double* Yt = smem;                                  // [tn][ld]
  double* swl = Yt + (size_t)tn * ldmax;              // [k]
  double* wrow = swl + k;                             // [3][tn]
  double* red = wrow + 3 * tn;                        // [4]

  for (long it = blockIdx.x; it < S.nbatch; it += gridDim.x) {
    const long pt = S.pt0 + it;
    Slab sl = slab_of(A.ws + (size_t)it * A.ws_per_block, k, A.nv, S.kkout);
    ObsView ov;
    ov.A = &A;
    ov.pt = pt;
    double beta = 1.0;
    if (A.mode == 0) {
      ov.o0 = A.obs_off[pt];
      ov.n = (int)(A.obs_off[pt + 1] - ov.o0);
      if (A.beta) beta = A.beta[pt];
    } else {
      ov.o0 = 0;
      ov.n = A.nobsl[pt];
    }
    const int n = ov.n;
    __syncthreads();
    if (S.gram_mfma && !(n > 0 && beta != 0.0 && !gram_mfma_takes(n, k))) continue;   // (letkf_gram.hip has done this point)
    if (A.skip_trivial && (n == 0 || beta == 0.0)) {  // done by the streaming pass (letkf_trivial.hip): no stage touches it
      if (tid == 0) {
        S.meta[2 * it] = 255;
        S.meta[2 * it + 1] = -1;
      }
      continue;
    }
    if (A.mode == 0 && beta == 0.0) {                 // letkf_tools.f90:333-359: nothing to solve
      if (tid == 0) {
        S.meta[2 * it] = 0;
        S.meta[2 * it + 1] = -1;
      }
      continue;
    }
    // inflation slot that drives the solve (first updated variable of the class, letkf_tools.f90:387-418)
    double infl_old;
    if (A.mode == 0) {
      bool qskip = false;
      if (A.q_update_top > 0.0) qskip = A.gues[pt * A.sp + k * A.sm + A.iv_p * A.sv] < A.q_update_top;
      int v0 = 0;
      while (v0 < A.nv && (!((A.var_mask >> v0) & 1u) || (qskip && v0 >= A.iv_q_first && v0 <= A.iv_q_last))) ++v0;
      infl_old = v0 < A.nv ? A.infl[pt + A.infl_sv * (long)v0] : 1.0;
    } else {
      infl_old = A.infl[pt];
    }
    const double shift = km1 / infl_old;              // common_letkf.f90:140-143
    const bool dual = n < k;
    const int m = dual ? n : k;
    if (tid == 0) {
      const int solver = m >= 2 ? (m <= S.wg_max_order ? 1 : 2) : 0;
      S.meta[2 * it] = (dual ? 2 : 1) | (solver << 8);
      S.meta[2 * it + 1] = m;
      sl.SC[3] = shift;
      sl.SC[4] = infl_old;
    }
    if (tid < 4) red[tid] = 0.0;
    __syncthreads();
    if (n == 0) {                                     // dual with an empty spectrum
      if (tid == 0) sl.SC[0] = sl.SC[1] = sl.SC[2] = 0.0;
      continue;
    }
    const int ldg = staged_ld(m);
    const int ld = ((m + 3) & ~3) + 4;                // tile row length (output index), b128-aligned rows
    const int T = (m + kGT - 1) / kGT;
    const long ntile = (long)T * (T + 1) / 2;         // tiles (ti <= tj)
    const int L = dual ? k : n;                       // contraction length

    // ---- per-observation scalars.  Dual: all of them up front (sqrt w stays in LDS, the weighted departures go
    // to the slab); primal: per obs tile below, and r = Z^T (sqrt(w) dep) accumulated.
    double p1 = 0.0, p3 = 0.0;
    if (dual) {
      for (int i = tid; i < n; i += nthr) {
        double w, d, dd, rl;
        ov.weights(i, w, d, dd, rl);
        const double sw = sqrt(w);
        swl[i] = sw;
        sl.V0[i] = sw * d;
        sl.V1[i] = sw * dd;
        p1 = fma(d * d, w, p1);
        p3 += rl;
      }
    } else {
      for (int j = tid; j < 2 * k; j += nthr) (j < k ? sl.V0 : sl.V1)[j < k ? j : j - k] = 0.0;
    }
    __syncthreads();

    double trp = 0.0;
    for (long tile0 = 0; tile0 < ntile; tile0 += (long)nthr * kGMaxT) {
      double acc[kGMaxT][kGT * kGT];
      int tis[kGMaxT], tjs[kGMaxT];
#pragma unroll
      for (int t = 0; t < kGMaxT; ++t) {
#pragma unroll
        for (int e = 0; e < kGT * kGT; ++e) acc[t][e] = 0.0;
        const long tl = tile0 + tid + (long)t * nthr;
        int tj = (int)((sqrt(8.0 * (double)tl + 1.0) - 1.0) * 0.5);
        while ((long)tj * (tj + 1) / 2 > tl) --tj;
        while ((long)(tj + 1) * (tj + 2) / 2 <= tl) ++tj;
        tjs[t] = tj;
        tis[t] = (int)(tl - (long)tj * (tj + 1) / 2);
      }
      for (int l0 = 0; l0 < L; l0 += tn) {
        const int nl = min(tn, L - l0);
        __syncthreads();
        if (!dual) {
          // rows = observations l0 .. l0+nl: Yt[i][mm] = sqrt(w_i) y_i[mm]
          if (tid < nl) {
            double w, d, dd, rl;
            ov.weights(l0 + tid, w, d, dd, rl);
            const double sw = sqrt(w);
            wrow[tid] = sw;
            wrow[tn + tid] = sw * d;
            wrow[2 * tn + tid] = sw * dd;
            if (tile0 == 0) {
              p1 = fma(d * d, w, p1);
              p3 += rl;
            }
          }
          __syncthreads();
          if (A.mode == 0) {
            for (int e = tid; e < nl * ld; e += nthr) {
              const int i = e / ld, mm = e - i * ld;
              long ms;
              const double* yr = ov.row(l0 + i, ms);
              Yt[e] = mm < k ? yr[mm] * wrow[i] : 0.0;
            }
          } else {
            for (int e = tid; e < nl * ld; e += nthr) {   // dense hdxb: consecutive threads walk down a column
              const int mm = e / nl, i = e - mm * nl;
              long ms;
              const double* yr = ov.row(l0 + i, ms);
              Yt[i * ld + mm] = mm < k ? yr[(long)mm * ms] * wrow[i] : 0.0;
            }
          }
        } else {
          // rows = members l0 .. l0+nl: Yt[mm][i] = sqrt(w_i) y_i[l0 + mm]
          if (A.mode == 0) {
            for (int e = tid; e < ld * nl; e += nthr) {   // consecutive threads along the members of one obs row
              const int i = e / nl, mm = e - i * nl;
              double v = 0.0;
              if (i < n) {
                long ms;
                const double* yr = ov.row(i, ms);
                v = yr[l0 + mm] * swl[i];
              }
              Yt[mm * ld + i] = v;
            }
          } else {
            for (int e = tid; e < ld * nl; e += nthr) {   // dense hdxb: consecutive threads along the observations
              const int mm = e / ld, i = e - mm * ld;
              double v = 0.0;
              if (i < n) {
                long ms;
                const double* yr = ov.row(i, ms);
                v = yr[(long)(l0 + mm) * ms] * swl[i];
              }
              Yt[e] = v;
            }
          }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < kGMaxT; ++t) {
          if (tile0 + tid + (long)t * nthr < ntile) {
            const double* ya = Yt + tis[t] * kGT;
            const double* yb = Yt + tjs[t] * kGT;
            for (int l = 0; l < nl; ++l) {
              double av[kGT], bv[kGT];
#pragma unroll
              for (int e = 0; e < kGT; ++e) {
                av[e] = ya[l * ld + e];
                bv[e] = yb[l * ld + e];
              }
#pragma unroll
              for (int ea = 0; ea < kGT; ++ea)
#pragma unroll
                for (int eb = 0; eb < kGT; ++eb) acc[t][ea * kGT + eb] = fma(av[ea], bv[eb], acc[t][ea * kGT + eb]);
            }
          }
        }
        if (!dual && tile0 == 0) {
          // r_j += sum_i Z[i][j] sqrt(w_i) dep_i, same for the deterministic departure (common_letkf.f90:169-195 folded)
          for (int j = tid; j < 2 * k; j += nthr) {
            const int col = j < k ? j : j - k;
            const double* wd = wrow + (j < k ? tn : 2 * tn);
            double sacc = 0.0;
            for (int i = 0; i < nl; ++i) sacc = fma(Yt[i * ld + col], wd[i], sacc);
            (j < k ? sl.V0 : sl.V1)[col] += sacc;
          }
        }
      }
      // finished tiles -> G (both triangles), shift on the diagonal
#pragma unroll
      for (int t = 0; t < kGMaxT; ++t) {
        if (tile0 + tid + (long)t * nthr < ntile) {
#pragma unroll
          for (int ea = 0; ea < kGT; ++ea)
#pragma unroll
            for (int eb = 0; eb < kGT; ++eb) {
              const int r = tis[t] * kGT + ea, c = tjs[t] * kGT + eb;
              if (r < m && c < m) {
                const double v = acc[t][ea * kGT + eb];
                if (r == c) {
                  trp += v;
                  sl.G[(size_t)c * ldg + r] = v + shift;
                } else {
                  sl.G[(size_t)c * ldg + r] = v;
                  if (tis[t] != tjs[t]) sl.G[(size_t)r * ldg + c] = v;
                }
              }
            }
        }
      }
    }
    // adaptive-inflation sums (common_letkf.f90:233-249): sum dep^2 w, sum rloc, trace(Z^T Z) = trace(Z Z^T)
    p1 = wsum(p1);
    p3 = wsum(p3);
    trp = wsum(trp);
    if ((tid & 63) == 0) {
      atomicAdd(&red[0], p1);
      atomicAdd(&red[1], p3);
      atomicAdd(&red[2], trp);
    }
    __syncthreads();
    if (tid == 0) {
      sl.SC[0] = red[0];
      sl.SC[1] = red[1];
      sl.SC[2] = red[2];
    }
    // ---- eigen-free stage (letkf_krylov.hip): the loop body without k x k outputs needs only functions of M applied to
    // nv + 2 vectors, which conjugate gradients + Lanczos deliver without an eigen-decomposition and without a bound on
    // the spectrum.  Kept away from it: orders beyond what the stage holds, and matrices that are hopeless for any
    // iteration (NaN / Inf, or a mean eigenvalue of S beyond 1e5 c: the eigen stage then reports the point).
    if (S.poly_max_n > 0 && m >= 2 && m <= S.poly_max_n && tid == 0) {
      const double tr = red[2];
      if (tr == tr && tr <= 1e5 * shift * (double)m) S.meta[2 * it] = (dual ? 2 : 1) | (3 << 8);
    }
  }
}

// ------------------------------------------------------------------------------------------------ stage 3
namespace {

// P[b][j] = sum_i C_j[i] R_b[i], j < ncols: one wave per column j of the (len x ncols) matrix Cm (leading dimension
// ldc), lanes along i; R_b = Rm + b * ldr (nbr right-hand sides)
__device__ __forceinline__ void cols_dot(const double* __restrict__ Cm, const int ldc, const int len, const int ncols,
                                         const double* __restrict__ Rm, const int ldr, const int nbr,
                                         double* __restrict__ P, const int ldp) {
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwv = blockDim.x >> 6;
  for (int j = wv; j < ncols; j += nwv) {
    const double* cj = Cm + (size_t)j * ldc;
    double acc[kMaxNb];
#pragma unroll
    for (int b = 0; b < kMaxNb; ++b) acc[b] = 0.0;
    for (int i = lane; i < len; i += 64) {
      const double c = cj[i];
#pragma unroll
      for (int b = 0; b < kMaxNb; ++b)
        if (b < nbr) acc[b] = fma(c, Rm[(size_t)b * ldr + i], acc[b]);
    }
#pragma unroll
    for (int b = 0; b < kMaxNb; ++b)
      if (b < nbr) {
        const double s = wsum(acc[b]);
        if (lane == 0) P[(size_t)b * ldp + j] = s;
      }
  }
}

// O[b][i] = sum_j Cm[i][j] K_b[j] (+ add0 * Add_b[i]), i < len: one thread per row i
__device__ __forceinline__ void rows_comb(const double* __restrict__ Cm, const int ldc, const int len, const int ncols,
                                          const double* __restrict__ K, const int ldk, const int nbr,
                                          double* __restrict__ O, const int ldo) {
  for (int i = threadIdx.x; i < len; i += blockDim.x) {
    double acc[kMaxNb];
#pragma unroll
    for (int b = 0; b < kMaxNb; ++b) acc[b] = 0.0;
    for (int j = 0; j < ncols; ++j) {
      const double c = Cm[(size_t)j * ldc + i];
#pragma unroll
      for (int b = 0; b < kMaxNb; ++b)
        if (b < nbr) acc[b] = fma(c, K[(size_t)b * ldk + j], acc[b]);
    }
#pragma unroll
    for (int b = 0; b < kMaxNb; ++b)
      if (b < nbr) O[(size_t)b * ldo + i] = acc[b];
  }
}

}  // namespace

__device__ __forceinline__ int ov_n_of(const PointArgs& A, long pt) {
  return A.mode == 0 ? (int)(A.obs_off[pt + 1] - A.obs_off[pt]) : A.nobsl[pt];
}

// LDS: lam [k] | tau [k] | pi [k] | om [k] | swl [k] | small [8 nv + 32] | P / C, q
// NTHR / MINW (r4): 256 threads at >= 4 waves per SIMD (128 registers: up to four workgroups per CU) for k <= 512 -- the stage is a
// chain of short dependent passes over strided state and slab words, and more points in flight per CU is what hides them
// (A/B: MEMBER = 100 +10 %, k = 320 +5.6 %); 512 threads, one workgroup's worth of registers, for the larger orders (k = 1000: -2.7 % with 256).
template <int NTHR, int MINW>
__global__ void __launch_bounds__(NTHR, MINW) letkf_stage_apply_kernel(const StagedArgs S, const int pcq_doubles) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const PointArgs& A = S.A;
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = nthr >> 6;
  const int k = A.k, nv = A.nv, nb = nv + 2;
  const double km1 = (double)(k - 1);
  double* lam = smem;                  // k + 2 each (up to k + 1 stored columns)
  double* tau = lam + (k + 2);
  double* pis = tau + (k + 2);
  double* om = pis + (k + 2);
  double* swl = om + (k + 2);
  double* xsm = swl + (k + 2);
  double* xmean = xsm;
  double* xdet = xsm + nv;
  double* cfac = xsm + 2 * nv;
  double* cdiag = xsm + 3 * nv;
  double* sdot = xsm + 4 * nv;
  double* sdotd = xsm + 5 * nv;
  double* varg = xsm + 6 * nv;
  double* red = xsm + 8 * nv;          // 32 scalars
  double* pcq = red + 32;              // [pcq_doubles]: the coefficient matrices P / C and q when they fit (see below)

  for (long it = blockIdx.x; it < S.nbatch; it += gridDim.x) {
    const long pt = S.pt0 + it;
    Slab sl = slab_of(A.ws + (size_t)it * A.ws_per_block, k, nv, S.kkout);
    const int meta0 = S.meta[2 * it], m = S.meta[2 * it + 1];
    const int mode = meta0 & 0xff, solver = meta0 >> 8;
    const bool das = A.mode == 0;
    const double* g0 = das ? A.gues + pt * A.sp : nullptr;
    double* a0 = das ? A.anal + pt * A.sp : nullptr;
    __syncthreads();

    if (mode == 255) continue;                         // (the streaming pass has written this point)
    if (mode == 0) {                                   // beta == 0: letkf_tools.f90:333-359
      for (int e = tid; e < nv * k; e += nthr) {
        const int v = e / k, mm = e - v * k;
        if ((A.var_mask >> v) & 1u) a0[mm * A.sm + v * A.sv] = g0[k * A.sm + v * A.sv] + g0[mm * A.sm + v * A.sv];
      }
      for (int v = tid; v < nv; v += nthr)
        if ((A.var_mask >> v) & 1u) {
          if (A.det_run) a0[(k + 1) * A.sm + v * A.sv] = g0[(k + 1) * A.sm + v * A.sv];
          if (A.rtps_out) A.rtps_out[pt + A.infl_sv * (long)v] = 1.0;
        }
      if (tid == 0) {
        if (A.status) A.status[pt] = 0;
        if (A.nsweep) A.nsweep[pt] = 0;
      }
      continue;
    }
    const bool dual = mode == 2;
    const bool poly = solver == 3;                     // eigen-free (letkf_krylov.hip has left q and the quadratic forms): sl.G still holds M itself
    const int ldg = staged_ld(m);
    // the workgroup Jacobi pads an odd order with a zero column that ends up anywhere among the stored columns
    const int mc = solver == 1 ? (m + 1) & ~1 : m;
    // P / C [nb][mc] and (dual) q [nb][n] are read as broadcasts in the innermost loops below: in LDS when they fit
    // (from the slab in L2 every one of those reads was an exposed ~0.7 us latency: 3/4 of this kernel's time)
    const int mq = (mc + 2) & ~1, nq = (ov_n_of(A, pt) + 2) & ~1;
    const bool pc_lds = (long)nb * mq <= pcq_doubles;
    const bool qq_lds = pc_lds && (long)nb * (mq + nq) <= pcq_doubles;
    double* PCp = pc_lds ? pcq : sl.PC;
    const int kq = pc_lds ? mq : k + 2;                 // row length of PC
    const bool pq_lds = poly && (long)nb * nq <= pcq_doubles;   // eigen-free: q [nb][nq] alone
    double* QQp = poly ? (pq_lds ? pcq : sl.QQ) : qq_lds ? pcq + (size_t)nb * mq : sl.QQ;
    const int qld = poly ? (pq_lds ? nq : k) : qq_lds ? nq : k;
    const double shift = sl.SC[3], infl_old = sl.SC[4];
    ObsView ov;
    ov.A = &A;
    ov.pt = pt;
    double beta = 1.0;
    if (das) {
      ov.o0 = A.obs_off[pt];
      ov.n = (int)(A.obs_off[pt + 1] - ov.o0);
      if (A.beta) beta = A.beta[pt];
    } else {
      ov.o0 = 0;
      ov.n = A.nobsl[pt];
    }
    const int n = ov.n;

    // ---------------- normalise the eigen-columns: lambda_j = |g_j|, e_j = g_j / lambda_j (in place)
    for (int j = wv; j < (poly ? 0 : mc); j += nwv) {
      double* gj = sl.G + (size_t)j * ldg;
      double ss = 0.0;
      for (int r = lane; r < m; r += 64) ss = fma(gj[r], gj[r], ss);
      ss = wsum(ss);
      const double l = sqrt(ss), il = ss > 0.0 ? 1.0 / l : 0.0;   // (ss == 0: the padding column)
      for (int r = lane; r < m; r += 64) gj[r] *= il;
      if (lane == 0) lam[j] = l;
    }
    if (dual)
      for (int i = tid; i < n; i += nthr) {
        double w, d, dd, rl;
        ov.weights(i, w, d, dd, rl);
        swl[i] = sqrt(w);
      }
    if (tid < 32) red[tid] = 0.0;
    __syncthreads();

    // ---------------- status (common_mtx.f90:66-78) and the spectra
    int st = 0;
    if (!poly) {
      double lmx = 0.0, lmn = 1e300;
      bool nan = false;                               // (a NaN in the point's observations: every comparison of the Jacobi is
                                                      //  false, it "converges" at once -- fmax / fmin would drop the NaN here)
      for (int j = lane; j < mc; j += 64) {
        lmx = fmax(lmx, lam[j]);
        if (lam[j] > 0.0 || mc == m) lmn = fmin(lmn, lam[j]);
        nan = nan || !(lam[j] == lam[j]);
      }
#pragma unroll
      for (int mk = 1; mk < 64; mk <<= 1) {
        lmx = fmax(lmx, __shfl_xor(lmx, mk, 64));
        lmn = fmin(lmn, __shfl_xor(lmn, mk, 64));
      }
      const bool any_nan = __ballot(nan) != 0ull;     // reported below as "not converged"
      if (dual) {                                     // A's spectrum = {lambda_j} and k - n copies of c
        lmx = fmax(lmx, shift);
        lmn = fmin(lmn, shift);
      }
      const bool conv = (solver == 0 || S.info[2 * it + 1] != 0) && !any_nan;
      if (!conv) st = 1;
      else if (!(lmx > 0.0)) st = 2;
      else if (lmn < lmx * 1.4901161193847656e-08) st = 3;       // sqrt(DBL_EPSILON)
    }
    const double sqc = sqrt(shift), sqkm1 = sqrt(km1);
    const double tau0 = dual ? sqrt(km1 / shift) : 0.0;            // f_T(c) = sqrt(rho)
    const double pi0 = dual ? 1.0 / shift : 0.0;                   // f_Pa(c) = rho / (k-1)
    for (int j = tid; j < (poly ? 0 : mc); j += nthr) {
      const double l = lam[j];
      if (!(l > 0.0) && mc != m) {                     // padding column: no contribution anywhere
        om[j] = tau[j] = pis[j] = 0.0;
        continue;
      }
      om[j] = 1.0 / l;
      if (dual) {
        const double sl_ = sqrt(l);
        tau[j] = -sqkm1 / (sqc * sl_ * (sqc + sl_));
        pis[j] = -1.0 / (shift * l);
      } else {
        tau[j] = sqrt(km1 / l);
        pis[j] = 1.0 / l;
      }
    }

    // ---------------- stage the perturbations X[v][mm]; right-hand sides in the solver's space TT[b][.]
    // (dual: a second copy of X in LDS -- the pcq region is free until the coefficients / the T buffers move in -- for the
    //  Z x'_v pass below, which reads all of X once per observation row)
    const bool x_lds = das && dual && !poly && (long)nv * k <= pcq_doubles;
    if (das) {
      for (int e = tid; e < nv * k; e += nthr) {
        const int v = e / k, mm = e - v * k;
        const double xv = g0[mm * A.sm + v * A.sv];
        sl.X[e] = xv;
        if (x_lds) pcq[e] = xv;
      }
      for (int v = tid; v < nv; v += nthr) {
        xmean[v] = g0[k * A.sm + v * A.sv];
        xdet[v] = A.det_run ? g0[(k + 1) * A.sm + v * A.sv] : 0.0;
      }
    }
    __syncthreads();
    const int nbr = das ? nb : 2;                      // letkf_core batch: only w-bar (and w-bar_det)
    if (poly) {
      // (the eigen-free stage built its own right-hand sides)
    } else if (!dual) {
      // TT[0] = r, TT[1] = r_det (already in V0 / V1), TT[2 + v] = x'_v
      for (int e = tid; e < 2 * k; e += nthr) sl.TT[e] = e < k ? sl.V0[e] : sl.V1[e - k];
      if (das)
        for (int e = tid; e < nv * k; e += nthr) sl.TT[2 * k + e] = sl.X[e];
    } else {
      // TT[0] = sqrt(w) dep, TT[1] = sqrt(w) dep_det, TT[2 + v][i] = (Z x'_v)_i: one wave per observation row
      for (int e = tid; e < 2 * n; e += nthr) sl.TT[(size_t)(e < n ? 0 : 1) * k + (e < n ? e : e - n)] = e < n ? sl.V0[e] : sl.V1[e - n];
      if (das)
        for (int i = wv; i < n; i += nwv) {
          long ms;
          const double* yr = ov.row(i, ms);
          double acc[kMaxNb];
#pragma unroll
          for (int v = 0; v < kMaxNb; ++v) acc[v] = 0.0;
          const double* xs = x_lds ? pcq : sl.X;
          for (int mm = lane; mm < k; mm += 64) {
            const double y = yr[(long)mm * ms];
#pragma unroll
            for (int v = 0; v < kMaxNb; ++v)
              if (v < nv) acc[v] = fma(y, xs[(size_t)v * k + mm], acc[v]);
          }
          const double sw = swl[i];
#pragma unroll
          for (int v = 0; v < kMaxNb; ++v)
            if (v < nv) {
              const double s = wsum(acc[v]);
              if (lane == 0) sl.TT[(size_t)(2 + v) * k + i] = s * sw;
            }
        }
    }
    __syncthreads();

    // ---------------- coefficients P[b][j] = e_j . TT[b]
    if (!poly) cols_dot(sl.G, ldg, m, mc, sl.TT, k, nbr, PCp, kq);
    // var_g per variable (RTPS), needed with or without observations
    if (das)
      for (int v = wv; v < nv; v += nwv) {
        double s = 0.0;
        for (int mm = lane; mm < k; mm += 64) s = fma(sl.X[(size_t)v * k + mm], sl.X[(size_t)v * k + mm], s);
        s = wsum(s);
        if (lane == 0) varg[v] = s;
      }
    __syncthreads();
    if (poly) {
      // va_v = t_v^T M^-1 t_v (q_b itself is fetched by the Z^T q pass below)
      for (int v = tid; v < nv; v += nthr) xsm[7 * nv + v] = sl.PC[v];
      __syncthreads();
    }

    // ---------------- relaxation scalars per variable (letkf_tools.f90:457-469, :1953-2002)
    if (das)
      for (int v = tid; v < nv; v += nthr) {
        const double parm = A.relax_to_inflated_prior ? A.infl[pt + A.infl_sv * (long)v] : 1.0;   // :387-391
        double cf = 1.0, cd = 0.0;
        if (A.relax_alpha != 0.0) {                    // RTPP
          cf = 1.0 - A.relax_alpha;
          cd = A.relax_alpha * sqrt(parm);
        } else if (A.relax_alpha_spread != 0.0) {      // RTPS: var_a = x'^T Pa x' = pi0 |x'|^2 + sum_j pi_j P_j^2
          const double var_g = varg[v];
          double var_a = 0.0;
          if (poly) {
            var_a = dual ? -xsm[7 * nv + v] / shift : xsm[7 * nv + v];   // sum_j pi_j P_j^2 = -(1/c) t^T M^-1 t, t = Z x'_v  |  x'^T A^-1 x'
          } else {
            for (int j = 0; j < mc; ++j) {
              const double p = PCp[(size_t)(2 + v) * kq + j];
              var_a = fma(p * p, pis[j], var_a);
            }
          }
          var_a = fma(pi0, var_g, var_a);
          if (var_g > 0.0 && var_a > 0.0)
            cf = A.relax_alpha_spread * sqrt(var_g * parm / (var_a * km1)) - A.relax_alpha_spread + 1.0;
        }
        cfac[v] = cf;
        cdiag[v] = cd;
      }
    __syncthreads();
    // ---------------- C = spectrum * P  (in place): rows 0, 1 with 1/lambda (w-bar), rows 2.. with the T spectrum
    for (int e = tid; e < (poly ? 0 : nbr * mc); e += nthr) {
      const int b = e / mc, j = e - b * mc;
      PCp[(size_t)b * kq + j] *= (b < 2) ? om[j] : tau[j];
    }
    __syncthreads();

    // ---------------- back to member space: OUT[b][mm]
    if (!dual) {
      if (!poly) rows_comb(sl.G, ldg, k, mc, PCp, kq, nbr, sl.OUT, k);   // (eigen-free: OUT is written)
    } else {
      if (!poly) rows_comb(sl.G, ldg, n, mc, PCp, kq, nbr, QQp, qld);      // q_b = U c_b  (obs space; eigen-free: q is in the slab)
      __syncthreads();
      // OUT[b][mm] = sum_i Z[i][mm] q_b[i] (+ f_T(c) x'_v[mm]).  On the matrix cores when the operands fit: the contraction
      // runs over the observation rows; for a group of 32 members lane (c, q) fetches Y[row 4 s + q][mm0 + 2 c, + 1] as one
      // 16-byte load per step s -- 256 contiguous bytes per observation row -- whose halves are the A operands of TWO 16-row
      // blocks (the group's even and odd members); B = sqrt(w_i) q_b[i] in LDS ([i][16]); 8 loads in flight per wave, every
      // one unconditional (see letkf_krylov.hip ring_steps).  (Before: a thread per member with four rows of loads ahead of
      // 52 multiply-adds -- 3.2 ms per 3456 points of C3-slab, as much as the Gram stage.)
      const int n4 = (n + 3) & ~3;
      const long qb_off = poly ? 0 : qq_lds ? (long)nb * (mq + nq) : pc_lds ? (long)nb * mq : 0;
      const bool zq_mfma = das && qb_off + 16L * n4 <= pcq_doubles;
      if (zq_mfma) {
        double* qb = pcq + qb_off;
        long* roff = reinterpret_cast<long*>(lam);        // (the spectra's arrays are used up; n < k)
        const double* qsrc = poly ? sl.QQ : QQp;
        const int qsld = poly ? k : qld;
        for (int e = tid; e < n4 * 16; e += nthr) {
          const int i = e >> 4, b = e & 15;
          qb[e] = (i < n && b < nbr) ? qsrc[(size_t)b * qsld + i] * swl[i] : 0.0;
        }
        for (int i = tid; i < n4; i += nthr) roff[i] = (long)A.obs_idx[ov.o0 + (i < n ? i : n - 1)] * A.kld;
        __syncthreads();
        const int col = lane & 15, rq = lane >> 4;
        const int T = n4 >> 2, ng = (k + 31) >> 5;
        constexpr int ZR = 8;
        const int Tp = (T + ZR - 1) / ZR * ZR;
        for (int g = wv; g < ng; g += nwv) {
          const int mm0 = 32 * g;
          int eo = mm0 + 2 * col;
          if (eo > k - 1) eo = k - 1;                     // (a row holds k + 1 doubles; such members are dropped below)
          const double* ebase = A.ensval + eo;
          d2u ring[ZR];
#pragma unroll
          for (int u = 0; u < ZR; ++u) ring[u] = *reinterpret_cast<const d2u*>(ebase + roff[4 * (u < T ? u : 0) + rq]);
          d4 ae = (d4){0.0, 0.0, 0.0, 0.0}, ao = (d4){0.0, 0.0, 0.0, 0.0};
          for (int s0 = 0; s0 < Tp; s0 += ZR) {
#pragma unroll
            for (int u = 0; u < ZR; ++u) {
              const int s_ = s0 + u;
              if (s_ < T) {
                const double bq = qb[(size_t)(4 * s_) * 16 + lane];
                ae = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[u].x, bq, ae, 0, 0, 0);
                ao = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[u].y, bq, ao, 0, 0, 0);
              }
              int sn = s_ + ZR;
              if (sn >= T) sn = 0;
              ring[u] = *reinterpret_cast<const d2u*>(ebase + roff[4 * sn + rq]);
            }
          }
          if (col < nbr) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int o = 0; o < 2; ++o) {
                const int mm = mm0 + 2 * (rq + 4 * r) + o;
                const double v_ = o ? ao[r] : ae[r];
                if (mm < k) sl.OUT[(size_t)col * k + mm] = col >= 2 ? fma(tau0, sl.X[(size_t)(col - 2) * k + mm], v_) : v_;
              }
          }
        }
      } else {
        if (poly && pq_lds) {                             // (q_b is read as broadcasts below: LDS when it fits)
          for (int e = tid; e < nb * n; e += nthr) {
            const int b = e / n, i = e - b * n;
            pcq[(size_t)b * nq + i] = sl.QQ[(size_t)b * k + i];
          }
          __syncthreads();
        }
        // one thread per member, the obs rows streamed (coalesced along mm)
        for (int mm = tid; mm < k; mm += nthr) {
          double acc[kMaxNb];
#pragma unroll
          for (int b = 0; b < kMaxNb; ++b) acc[b] = 0.0;
          for (int i0 = 0; i0 < n; i0 += 4) {               // four observation rows of loads ahead of their FMAs
            double z[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int i = i0 + u < n ? i0 + u : n - 1;
              long ms;
              const double* yr = ov.row(i, ms);
              z[u] = i0 + u < n ? yr[(long)mm * ms] * swl[i] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int i = i0 + u < n ? i0 + u : n - 1;
#pragma unroll
              for (int b = 0; b < kMaxNb; ++b)
                if (b < nbr) acc[b] = fma(z[u], QQp[(size_t)b * qld + i], acc[b]);
            }
          }
#pragma unroll
          for (int b = 0; b < kMaxNb; ++b)
            if (b < nbr) sl.OUT[(size_t)b * k + mm] = (b >= 2) ? fma(tau0, sl.X[(size_t)(b - 2) * k + mm], acc[b]) : acc[b];
        }
      }
    }
    __syncthreads();
    const double* wbar = sl.OUT;
    const double* wbard = sl.OUT + k;

    // ---------------- adaptive inflation (common_letkf.f90:233-254); T and w-bar above used the OLD rho
    double infl_new = infl_old;
    if (A.infl_adaptive && n > 0) {
      const double parm1 = sl.SC[0], parm3 = sl.SC[1], parm2 = sl.SC[2] / km1;
      const double parm4 = (parm1 - parm3) / parm2 - infl_old;
      const double tq = (infl_old * parm2 + parm3) / parm2;
      const double sigma_o = 2.0 / parm3 * (tq * tq);
      const double gain = 0.04 * 0.04 / (sigma_o + 0.04 * 0.04);
      infl_new = infl_old + gain * parm4;
    }

    if (das) {
      bool qskip = false;
      if (A.q_update_top > 0.0) qskip = xmean[A.iv_p] < A.q_update_top;
      // x'_v . w-bar, x'_v . w-bar_det
      for (int e = wv; e < 2 * nv; e += nwv) {
        const int v = e < nv ? e : e - nv;
        const double* wvv = e < nv ? wbar : wbard;
        double s = 0.0;
        for (int mm = lane; mm < k; mm += 64) s = fma(sl.X[(size_t)v * k + mm], wvv[mm], s);
        s = wsum(s);
        if (lane == 0) (e < nv ? sdot : sdotd)[v] = s;
      }
      __syncthreads();
      // ---------------- analysis members (letkf_tools.f90:472-513)
      const bool clampq = A.q_sprd_max > 0.0 && !qskip && ((A.var_mask >> A.iv_q_first) & 1u);
      for (int e = tid; e < nv * k; e += nthr) {
        const int v = e / k, mm = e - v * k;
        const bool skip = qskip && v >= A.iv_q_first && v <= A.iv_q_last;
        const double xp = sl.X[e];
        double out;
        if (skip) {
          out = xmean[v] + xp;
        } else {
          const double tx = sl.OUT[(size_t)(2 + v) * k + mm];
          const double pert = cfac[v] * tx + cdiag[v] * xp;
          out = xmean[v] + beta * (pert + sdot[v]) + (1.0 - beta) * xp;
        }
        if (clampq && v == A.iv_q_first) sl.TT[mm] = out;          // kept for the clamp (TT is free now)
        else if ((A.var_mask >> v) & 1u) a0[mm * A.sm + v * A.sv] = out;
      }
      for (int v = tid; v < nv; v += nthr) {
        const bool skip = qskip && v >= A.iv_q_first && v <= A.iv_q_last;
        if ((A.var_mask >> v) & 1u) {
          if (A.det_run) a0[(k + 1) * A.sm + v * A.sv] = skip ? xdet[v] : xdet[v] + sdotd[v] * beta;   // :489-497
          if (A.rtps_out)
            A.rtps_out[pt + A.infl_sv * (long)v] = (A.relax_alpha == 0.0 && A.relax_alpha_spread != 0.0 && !skip) ? cfac[v] : 1.0;
          if (A.infl_adaptive && !skip) A.infl[pt + A.infl_sv * (long)v] = infl_new;   // :396-398
        }
      }
      if (clampq) {                                    // :500-513, variable iv3d_q only
        __syncthreads();
        const int v = A.iv_q_first;
        if (tid < 64) {
          double sm_ = 0.0;
          for (int mm = tid; mm < k; mm += 64) sm_ += sl.TT[mm];
          sm_ = wsum(sm_);
          const double q_mean = sm_ / (double)k;
          double ss = 0.0;
          for (int mm = tid; mm < k; mm += 64) {
            const double d = sl.TT[mm] - q_mean;
            ss = fma(d, d, ss);
          }
          ss = wsum(ss);
          const double q_sprd = sqrt(ss / km1) / q_mean;
          for (int mm = tid; mm < k; mm += 64) {
            double val = sl.TT[mm];
            if (q_sprd > A.q_sprd_max) val = q_mean + (val - q_mean) * A.q_sprd_max / q_sprd;
            a0[mm * A.sm + v * A.sv] = val;
          }
        }
      }
    } else if (A.infl_adaptive && n > 0 && tid == 0) {
      A.infl[pt] = infl_new;
    }

    // ---------------- optional k x k outputs (fine boundary, parity, diagnostics)
    if (A.trans_out || A.pa_out) {
      double* To = A.trans_out ? A.trans_out + (size_t)pt * k * k : nullptr;
      double* Po = A.pa_out ? A.pa_out + (size_t)pt * k * k : nullptr;
      const double* E;                                // spectral basis in member space: E[j][mm], leading dimension lde
      int lde;
      if (!dual) {
        E = sl.G;
        lde = ldg;
      } else {
        // W[j][mm] = sum_i U[i][j] Z[i][mm]  (n x n x k): one thread per member, per eigen-column a pass over the rows
        __syncthreads();
        for (int mm = tid; mm < k; mm += nthr) {
          for (int j0 = 0; j0 < mc; j0 += 8) {
            double acc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] = 0.0;
            for (int i = 0; i < n; ++i) {
              long ms;
              const double* yr = ov.row(i, ms);
              const double z = yr[(long)mm * ms] * swl[i];
#pragma unroll
              for (int u = 0; u < 8; ++u)
                if (j0 + u < mc) acc[u] = fma(z, sl.G[(size_t)(j0 + u) * ldg + i], acc[u]);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (j0 + u < mc) sl.W[(size_t)(j0 + u) * staged_ld(k) + mm] = acc[u];
          }
        }
        __syncthreads();
        E = sl.W;
        lde = staged_ld(k);
      }
      for (long e = tid; e < (long)k * k; e += nthr) {
        const int c = (int)(e / k), r = (int)(e - (long)c * k);   // column-major, coalesced over r
        double t = 0.0, pp = 0.0;
        for (int j = 0; j < mc; ++j) {
          const double vv = E[(size_t)j * lde + r] * E[(size_t)j * lde + c];
          t = fma(vv, tau[j], t);
          pp = fma(vv, pis[j], pp);
        }
        if (r == c) {
          t += tau0;
          pp += pi0;
        }
        if (To) To[e] = A.add_wbar_to_trans ? t + wbar[r] : t;   // common_letkf.f90:218-226
        if (Po) Po[e] = pp;
      }
    }
    if (A.transm_out)
      for (int j = tid; j < k; j += nthr) A.transm_out[(size_t)pt * k + j] = wbar[j];
    if (A.transmd_out)
      for (int j = tid; j < k; j += nthr) A.transmd_out[(size_t)pt * k + j] = wbard[j];
    if (tid == 0) {
      if (A.status) A.status[pt] = st;
      if (A.nsweep) A.nsweep[pt] = solver ? S.info[2 * it] : 0;
    }
  }
}

// ------------------------------------------------------------------------------------------------ launchers
hipError_t launch_stage_gram(const StagedArgs& s, size_t lds_max, hipStream_t st) {
  const int k = s.A.k;
  // tile rows: up to 32, fewer when the rows are long (primal k = 1000: 1008 doubles per row)
  const int ldmax = ((k + 3) & ~3) + 4;
  const size_t fixed = ((size_t)k + 3 * 32 + 8) * sizeof(double);
  const size_t budget = (lds_max > 160 * 1024 ? 160 * 1024 : lds_max) - 2048;
  int tn = 32;
  while (tn > 4 && fixed + (size_t)tn * ldmax * sizeof(double) > budget) tn -= 4;
  const size_t lds = fixed + (size_t)tn * ldmax * sizeof(double);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&letkf_stage_gram_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(letkf_stage_gram_kernel, dim3((unsigned)s.nbatch), dim3(kGBlock), lds, st, s, tn, ldmax);
  return hipGetLastError();
}

// LDS doubles of stage 3 beyond its fixed part: room for P / C (primal: up to k + 2 columns) and, while it fits, for q as
// well (dual: two matrices of ~n columns); an eigen-free point keeps only q there
int stage_apply_pcq_doubles(int k, int nv) {
  const int nb = nv + 2;
  const size_t fixed = (size_t)5 * (k + 2) + 8 * (size_t)nv + 32;
  size_t pcq = (size_t)2 * nb * (k + 4);
  while ((fixed + pcq) * sizeof(double) > STAGE_APPLY_LDSCAP && pcq > 0) pcq = pcq > (size_t)nb * 64 ? pcq - (size_t)nb * 64 : 0;
  return (int)pcq;
}

hipError_t launch_stage_apply(const StagedArgs& s, hipStream_t st) {
  const int k = s.A.k;
  const size_t fixed = (size_t)5 * (k + 2) + 8 * (size_t)s.A.nv + 32;
  const size_t pcq = (size_t)stage_apply_pcq_doubles(k, s.A.nv);
  const size_t lds = (fixed + pcq) * sizeof(double);
  auto go = [&](auto kern, int nthr) -> hipError_t {
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)s.nbatch), dim3(nthr), lds, st, s, (int)pcq);
    return hipGetLastError();
  };
#ifndef STAGE_APPLY_SMALL_K
#define STAGE_APPLY_SMALL_K 512
#endif
  if (k <= STAGE_APPLY_SMALL_K) return go(&letkf_stage_apply_kernel<256, 4>, 256);
  return go(&letkf_stage_apply_kernel<kABlock, STAGE_APPLY_MINWG>, kABlock);
}

}  // namespace letkf
