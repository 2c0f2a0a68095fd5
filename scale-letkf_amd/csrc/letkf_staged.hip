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

#include "letkf_device.h"

namespace letkf {

namespace {

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}

constexpr int kGT = 4;         // Gram register tile
constexpr int kGMaxT = 3;      // tiles per thread and pass
constexpr int kGBlock = 512;
#ifndef STAGE_APPLY_BLOCK
#define STAGE_APPLY_BLOCK 512
#endif
#ifndef STAGE_APPLY_MINWG
#define STAGE_APPLY_MINWG 1
#endif
#ifndef STAGE_POLY_ROWS
#define STAGE_POLY_ROWS 8      // rows of M whose loads are issued ahead of their FMAs (poly_apply)
#endif
#ifndef STAGE_POLY_RES
#define STAGE_POLY_RES 0       // k-steps of M per wave that stay in registers across the degrees (poly_apply; A/B)
#endif
#ifndef STAGE_APPLY_LDSCAP
#define STAGE_APPLY_LDSCAP (150 * 1024)
#endif
constexpr int kABlock = STAGE_APPLY_BLOCK;
constexpr int kMaxNb = 16;     // right-hand sides: nv + 2 <= 16

struct Slab {
  double *G, *V0, *V1, *SC, *X, *TT, *PC, *QQ, *OUT, *W;
};
__device__ __forceinline__ Slab slab_of(double* base, int k, int nv, int kkout) {
  const int ldg = k | 1, nb = nv + 2;
  Slab s;
  s.G = base;
  s.V0 = s.G + (size_t)(k + 1) * ldg;   // one spare column: an odd order is padded with an inert zero column (letkf_eig.hip)
  s.V1 = s.V0 + k;
  s.SC = s.V1 + k;
  s.X = s.SC + 16;
  s.TT = s.X + (size_t)nv * k;
  s.PC = s.TT + (size_t)nb * k;
  s.QQ = s.PC + (size_t)nb * (k + 2);   // PC rows are k + 2 long (coefficients of up to k + 1 stored columns)
  s.OUT = s.QQ + (size_t)nb * k;
  s.W = kkout ? s.OUT + (size_t)nb * k : nullptr;
  return s;
}

// where the point's observations come from: the obs table through the CSR lists (das_letkf body) or a dense
// column-major hdxb(nobs, ne) block (letkf_core batch)
struct ObsView {
  const PointArgs* A;
  long pt, o0;
  int n;
  __device__ __forceinline__ void weights(int i, double& w, double& d, double& dd, double& rl) const {
    if (A->mode == 0) {
      const long e = o0 + i;
      const int iob = A->obs_idx[e];
      rl = A->rloc_l[e];
      w = 1.0 / A->rdiag_l[e];
      d = A->dep[iob];
      dd = A->det_run ? A->ensval[(long)iob * A->kld + A->k] : 0.0;
    } else {
      const long e = pt * (long)A->nobs + i;
      rl = A->rloc[e];
      w = A->rdiag_wloc ? 1.0 / A->rdiag[e] : rl / A->rdiag[e];   // common_letkf.f90:111-123
      d = A->depv[e];
      dd = A->depd ? A->depd[e] : 0.0;
    }
  }
  // address of y_i[0] and the member stride
  __device__ __forceinline__ const double* row(int i, long& ms) const {
    if (A->mode == 0) {
      ms = 1;
      return A->ensval + (long)A->obs_idx[o0 + i] * A->kld;
    }
    ms = A->nobs;
    return A->hdxb + (size_t)pt * (size_t)A->nobs * (size_t)A->k + i;
  }
};

}  // namespace

long staged_ws_per_point(int k, int nv, int kkout) {
  const long ldg = k | 1, nb = nv + 2;
  long w = (long)(k + 1) * ldg + 2L * k + 16 + (long)nv * k + 4L * nb * k + 2L * nb;
  if (kkout) w += (long)k * ldg;
  return (w + 1) & ~1L;
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
      infl_old = v0 < A.nv ? A.infl[pt + A.npts * (long)v0] : 1.0;
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
    const int ldg = m | 1;
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
    // ---- polynomial path (stage 3, "eigen-free"): M = S + c I with S = Z Z^T positive semi-definite, so M's spectrum
    // lies in [c, c + |S|] for any norm bound |S| >= lambda_max(S); the smaller of the Frobenius norm and the largest
    // absolute row sum is free here (M is in L2).  cond = (c + |S|) / c fixes the Chebyshev degree that reaches 1e-16 for
    // functions analytic away from 0 (1/x, the T and Pa spectra): rate (sqrt(cond) - 1) / (sqrt(cond) + 1) per degree.
    if (S.poly_max_n > 0 && m >= 2 && m <= S.poly_max_n && m <= nthr) {   // (dual: M = Z Z^T + c I, n x n; primal: A = Z^T Z + c I, k x k)
      __syncthreads();                                // (G complete: every thread's stores; red[] read above)
      double fs = 0.0, rs = 0.0;
      if (tid < m) {
        for (int c = 0; c < m; ++c) {
          double v = sl.G[(size_t)c * ldg + tid];
          if (c == tid) v -= shift;
          fs = fma(v, v, fs);
          rs += fabs(v);
        }
      }
      fs = wsum(fs);
#pragma unroll
      for (int mk = 1; mk < 64; mk <<= 1) rs = fmax(rs, __shfl_xor(rs, mk, 64));
      if ((tid & 63) == 0) {                          // per-wave partials (sqrt(w) in swl is no longer needed), summed in a fixed order
        swl[tid >> 6] = fs;
        swl[16 + (tid >> 6)] = rs;
      }
      __syncthreads();
      if (tid == 0) {
        double f2 = 0.0, r1 = 0.0;
        for (int w = 0; w < (nthr >> 6); ++w) {
          f2 += swl[w];
          r1 = fmax(r1, swl[16 + w]);
        }
        // (never an empty interval: all-zero weights or perturbations give S = 0)
        const double bound = fmax(fmin(sqrt(f2), r1) * (1.0 + 1e-12), 1e-6 * shift);
        const double sk = sqrt((shift + bound) / shift);
        const double rate = (sk - 1.0) / (sk + 1.0);
        int deg = 1 << 20;                             // (rate -> 1: hopeless, and log(rate) -> -0 must not reach the cast)
        if (!(rate > 0.0)) deg = 4;
        else if (rate < 0.999) deg = (int)ceil(log(1e-16) / log(rate)) + 1;
        if (deg < 4) deg = 4;
        if (f2 <= 1.7e308 && deg <= S.poly_max_deg) {   // (a NaN / Inf anywhere in M makes the sum of squares fail this test:
                                                        //  the eigen stage then reports the point, status 1)
          S.meta[2 * it] = (dual ? 2 : 1) | (3 << 8);
          S.info[2 * it] = -deg;                      // no sweeps: nsweep reports -(degree) ...
          S.info[2 * it + 1] = 1;                     // ... and nothing that could fail to converge
          sl.SC[5] = bound;
          sl.SC[6] = (double)deg;
        }
      }
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

// ---------------------------------------------------------------- the polynomial ("eigen-free") apply
// Everything stage 3 takes from the eigen-decomposition M = U L U^T of the n x n matrix M = Z Z^T + c I is a FUNCTION of M
// applied to a handful of vectors: q_b = U g(L) U^T t_b for the nb right-hand sides t_0,1 = sqrt(w) dep(_det), t_2+v = Z x'_v,
// with g = 1/L for w-bar and g = the T spectrum -sqrt(k-1) / (sqrt(c) sqrt(L) (sqrt(c) + sqrt(L))) for the members, plus the
// quadratic forms t^T M^-1 t of RTPS (var_a, letkf_tools.f90:1981-1990).  The shift c = (k-1)/rho keeps cond(M) small
// (C3 / C5: 1.5 exact, ~2.3 with the free norm bound of stage 1), so a Chebyshev expansion of g on [c, c + |S|] reaches
// 1e-16 in ~20-30 terms: 2 n^2 nb flops each -- a third of the flops of the Jacobi's ~9 sweeps, and regular, barrier-
// per-degree work instead of a latency-bound iteration.  The functions are analytic on the interval (the nearest
// singularity is L = 0), the coefficients come from interpolation at the Chebyshev nodes of the point's own interval.
// The same holds in member space for a point with n >= k: functions of the k x k matrix A = Z^T Z + c I applied to r, r_det,
// x'_v (g_T = sqrt(k-1) / sqrt(L); dual == false).  On return q_b[i] sits in qout[b * nq + i] (dual: the LDS copy that the
// Z^T q pass reads; primal: OUT itself) and va[v] = t_v^T M^-1 t_v.
// The product M T_d runs on the FP64 matrix cores (v_mfma_f64_16x16x4: A = a 16 x 4 tile of M straight from L2 -- lane l
// loads M[i0 + (l & 15)][j0 + (l >> 4)], 128 contiguous bytes per 16 lanes --, B = 4 rows of T_d from LDS, lane l reads
// T[j0 + (l >> 4)][l & 15] = 64 consecutive doubles, conflict-free; D: lane l holds rows (l >> 4) + 4 r, r < 4, of column
// l & 15).  A wave owns the 16-row blocks w, w + 8, ... (BPW of them); the recurrence state T_d, T_d-1 and the two running
// sums live in registers in D's layout, the new T_d goes to the other LDS buffer ([row][16], the right-hand sides padded to
// 16 columns of which the last ones stay zero), ONE barrier per degree.  (First versions: a thread per row with broadcast
// reads of T_d from LDS -- 62.9 ms on C3-slab with all right-hand sides per thread, 23.8 ms with the right-hand sides
// dealt to parts of the workgroup and 8 rows of loads in flight; a fixed-degree timing twin showed 0.56 ms per degree
// there, bound by the LDS broadcasts: 4 ds_read_b128 per row of M and wave.)
template <int BPW>
__device__ __forceinline__ void poly_apply(const Slab& sl, const int n, const int ldg, const int k, const int nv, const int nbr,
                                           const double shift, const double sqc, const double sqkm1, double* fw, double* ft,
                                           double* cw, double* ct, double* pcq, double* qout, const int nq, double* va,
                                           const bool dual) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = nthr >> 6;
  const double bound = sl.SC[5];
#ifdef STAGE_POLY_TIMING_DEG   // A/B twins only (make VARIANT=...): a fixed number of terms -- timing of the recurrence, results invalid
  const int deg = STAGE_POLY_TIMING_DEG, N = deg + 1;
#else
  const int deg = (int)sl.SC[6], N = deg + 1;
#endif
  const double lo = shift, hi = shift + bound;
  const double half = 0.5 * (hi - lo), mid = 0.5 * (hi + lo);
  // g at the Chebyshev nodes, then the coefficients c_i = (2 - [i = 0]) / N sum_j g(x_j) cos(pi i (j + 1/2) / N)
  for (int j = tid; j < N; j += nthr) {
    const double L = fma(half, cospi(((double)j + 0.5) / (double)N), mid);
    const double sL = sqrt(L);
    fw[j] = 1.0 / L;
    ft[j] = dual ? -sqkm1 / (sqc * sL * (sqc + sL)) : sqkm1 / sL;   // T = sqrt(rho) I + Z^T U g U^T Z  |  T = sqrt(k-1) A^-1/2
  }
  __syncthreads();
  for (int i = tid; i < N; i += nthr) {
    double sw_ = 0.0, st_ = 0.0;
    for (int j = 0; j < N; ++j) {
      const double cc = cospi((double)i * ((double)j + 0.5) / (double)N);
      sw_ = fma(fw[j], cc, sw_);
      st_ = fma(ft[j], cc, st_);
    }
    const double f = (i == 0 ? 1.0 : 2.0) / (double)N;
    cw[i] = sw_ * f;
    ct[i] = st_ * f;
  }
  const int nr16 = (n + 15) & ~15;
  const int col = lane & 15, rq = lane >> 4;
  double* cur = pcq;
  double* oth = pcq + (size_t)nr16 * 16;
  double t0[BPW][4], t1[BPW][4], yw[BPW][4], yt[BPW][4];
#pragma unroll
  for (int bi = 0; bi < BPW; ++bi) {
    const int i0 = (wv + bi * nwv) * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i0 + rq + 4 * r;
      t0[bi][r] = (row < n && col < nbr) ? sl.TT[(size_t)col * k + row] : 0.0;
      t1[bi][r] = yw[bi][r] = yt[bi][r] = 0.0;
      if (i0 < nr16) cur[(size_t)row * 16 + col] = t0[bi][r];
    }
  }
  __syncthreads();
  for (int e = tid; e < nwv * 16; e += nthr) fw[e] = 0.0;   // (the node values are used up: per-wave partials of the quadratic forms)
  const double inv = 1.0 / half;                        // M~ = (M - mid I) / half maps the interval to [-1, 1]
  constexpr int KS = 16 / BPW;                          // k-steps (of 4 rows of M^T = columns j) whose loads go out together
  // The first RS k-steps of a wave's A operands stay in registers across the degrees (the stream of M from L2 / MALL is
  // what bounds the recurrence: 320 KB per degree and CU at n = 200)
  constexpr int RS = BPW <= 2 ? STAGE_POLY_RES / BPW : 0;
  double ares[BPW][RS > 0 ? RS : 1];
#pragma unroll
  for (int s = 0; s < RS; ++s) {
    const int j = 4 * s + rq;
#pragma unroll
    for (int bi = 0; bi < BPW; ++bi) {
      const int row = (wv + bi * nwv) * 16 + col;
      ares[bi][s] = (j < n && row < n) ? sl.G[(size_t)j * ldg + row] : 0.0;
    }
  }
  const int jres = 4 * RS < nr16 ? 4 * RS : nr16;       // columns [0, jres) come from the registers
  for (int d = 1; d <= deg; ++d) {
    d4 acc[BPW];
#pragma unroll
    for (int bi = 0; bi < BPW; ++bi) acc[bi] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < RS; ++s) {
      if (4 * s < nr16) {                               // (wave-uniform)
        const double bq = cur[(size_t)(4 * s) * 16 + lane];
#pragma unroll
        for (int bi = 0; bi < BPW; ++bi) acc[bi] = __builtin_amdgcn_mfma_f64_16x16x4f64(ares[bi][s], bq, acc[bi], 0, 0, 0);
      }
    }
    for (int j0 = jres; j0 < nr16; j0 += 4 * KS) {
      double av[BPW][KS], bq[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int j = j0 + 4 * s + rq;
#pragma unroll
        for (int bi = 0; bi < BPW; ++bi) {
          const int row = (wv + bi * nwv) * 16 + col;    // (A's row sits on lane & 15)
          av[bi][s] = (j < n && row < n) ? sl.G[(size_t)j * ldg + row] : 0.0;
        }
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) bq[s] = j0 + 4 * s < nr16 ? cur[(size_t)(j0 + 4 * s) * 16 + lane] : 0.0;
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int bi = 0; bi < BPW; ++bi) acc[bi] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[bi][s], bq[s], acc[bi], 0, 0, 0);
    }
    const double cwd = cw[d], ctd = ct[d], cw0 = cw[0], ct0 = ct[0];
#pragma unroll
    for (int bi = 0; bi < BPW; ++bi) {
      const int i0 = (wv + bi * nwv) * 16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double tn;
        if (d == 1) {
          tn = (acc[bi][r] - mid * t0[bi][r]) * inv;   // T_1 = M~ t
          yw[bi][r] = cw0 * t0[bi][r];
          yt[bi][r] = ct0 * t0[bi][r];
        } else {
          tn = fma(2.0 * inv, acc[bi][r] - mid * t1[bi][r], -t0[bi][r]);   // T_d = 2 M~ T_d-1 - T_d-2
          t0[bi][r] = t1[bi][r];
        }
        t1[bi][r] = tn;
        yw[bi][r] = fma(cwd, tn, yw[bi][r]);
        yt[bi][r] = fma(ctd, tn, yt[bi][r]);
        if (i0 < nr16) oth[(size_t)(i0 + rq + 4 * r) * 16 + col] = tn;
      }
    }
    __syncthreads();
    double* sw2 = cur;
    cur = oth;
    oth = sw2;
  }
  // (every read of the two T buffers lies behind the loop's last barrier: pcq is free)
  // q_b, and the quadratic forms t_v^T M^-1 t_v: a lane's rows, then the four lanes of a column, per-wave partials
  // summed in a fixed order
  double p = 0.0;
#pragma unroll
  for (int bi = 0; bi < BPW; ++bi) {
    const int i0 = (wv + bi * nwv) * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i0 + rq + 4 * r;
      if (row < n && col < nbr) {
        p = fma(sl.TT[(size_t)col * k + row], yw[bi][r], p);
        qout[(size_t)col * nq + row] = col < 2 ? yw[bi][r] : yt[bi][r];
      }
    }
  }
  p += __shfl_xor(p, 16, 64);
  p += __shfl_xor(p, 32, 64);
  if (lane < 16) fw[wv * 16 + lane] = p;
  __syncthreads();
  for (int v = tid; v < nv; v += nthr) {
    double s_ = 0.0;
    for (int w = 0; w < nwv; ++w) s_ += fw[w * 16 + 2 + v];
    va[v] = s_;
  }
  __syncthreads();
}

}  // namespace

__device__ __forceinline__ int ov_n_of(const PointArgs& A, long pt) {
  return A.mode == 0 ? (int)(A.obs_off[pt + 1] - A.obs_off[pt]) : A.nobsl[pt];
}

// LDS: lam [k] | tau [k] | pi [k] | om [k] | swl [k] | small [8 nv + 32] | P / C, q
__global__ void __launch_bounds__(kABlock, STAGE_APPLY_MINWG) letkf_stage_apply_kernel(const StagedArgs S, const int pcq_doubles) {
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
          if (A.rtps_out) A.rtps_out[pt + A.npts * (long)v] = 1.0;
        }
      if (tid == 0) {
        if (A.status) A.status[pt] = 0;
        if (A.nsweep) A.nsweep[pt] = 0;
      }
      continue;
    }
    const bool dual = mode == 2;
    const bool poly = solver == 3;                     // eigen-free: sl.G still holds M itself
    const int ldg = m | 1;
    // the workgroup Jacobi pads an odd order with a zero column that ends up anywhere among the stored columns
    const int mc = solver == 1 ? (m + 1) & ~1 : m;
    // P / C [nb][mc] and (dual) q [nb][n] are read as broadcasts in the innermost loops below: in LDS when they fit
    // (from the slab in L2 every one of those reads was an exposed ~0.7 us latency: 3/4 of this kernel's time)
    const int mq = (mc + 2) & ~1, nq = (ov_n_of(A, pt) + 2) & ~1;
    const bool pc_lds = (long)nb * mq <= pcq_doubles;
    const bool qq_lds = pc_lds && (long)nb * (mq + nq) <= pcq_doubles;
    double* PCp = pc_lds ? pcq : sl.PC;
    const int kq = pc_lds ? mq : k + 2;                 // row length of PC
    double* QQp = poly ? pcq : qq_lds ? pcq + (size_t)nb * mq : sl.QQ;
    const int qld = (poly || qq_lds) ? nq : k;
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
      for (int j = lane; j < mc; j += 64) {
        lmx = fmax(lmx, lam[j]);
        if (lam[j] > 0.0 || mc == m) lmn = fmin(lmn, lam[j]);
      }
#pragma unroll
      for (int mk = 1; mk < 64; mk <<= 1) {
        lmx = fmax(lmx, __shfl_xor(lmx, mk, 64));
        lmn = fmin(lmn, __shfl_xor(lmn, mk, 64));
      }
      if (dual) {                                     // A's spectrum = {lambda_j} and k - n copies of c
        lmx = fmax(lmx, shift);
        lmn = fmin(lmn, shift);
      }
      const bool conv = solver == 0 || S.info[2 * it + 1] != 0;
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
    const bool x_lds = das && dual && (long)nv * k <= pcq_doubles;
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
    if (!dual) {
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
      const int nblk = (m + 15) >> 4, per = (nblk + nwv - 1) / nwv;   // 16-row blocks per wave
      if (per <= 1) poly_apply<1>(sl, m, ldg, k, nv, nbr, shift, sqc, sqkm1, lam, tau, om, pis, pcq, dual ? pcq : sl.OUT, dual ? nq : k, xsm + 7 * nv, dual);
      else if (per <= 2) poly_apply<2>(sl, m, ldg, k, nv, nbr, shift, sqc, sqkm1, lam, tau, om, pis, pcq, dual ? pcq : sl.OUT, dual ? nq : k, xsm + 7 * nv, dual);
      else poly_apply<4>(sl, m, ldg, k, nv, nbr, shift, sqc, sqkm1, lam, tau, om, pis, pcq, dual ? pcq : sl.OUT, dual ? nq : k, xsm + 7 * nv, dual);
    }

    // ---------------- relaxation scalars per variable (letkf_tools.f90:457-469, :1953-2002)
    if (das)
      for (int v = tid; v < nv; v += nthr) {
        const double parm = A.relax_to_inflated_prior ? A.infl[pt + A.npts * (long)v] : 1.0;   // :387-391
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
      if (!poly) rows_comb(sl.G, ldg, k, mc, PCp, kq, nbr, sl.OUT, k);   // (poly_apply has written OUT)
    } else {
      if (!poly) rows_comb(sl.G, ldg, n, mc, PCp, kq, nbr, QQp, qld);      // q_b = U c_b  (obs space; poly_apply left q there)
      __syncthreads();
      // OUT[b][mm] = sum_i Z[i][mm] q_b[i]: one thread per member, the obs rows streamed (coalesced along mm)
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
            A.rtps_out[pt + A.npts * (long)v] = (A.relax_alpha == 0.0 && A.relax_alpha_spread != 0.0 && !skip) ? cfac[v] : 1.0;
          if (A.infl_adaptive && !skip) A.infl[pt + A.npts * (long)v] = infl_new;   // :396-398
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
              if (j0 + u < mc) sl.W[(size_t)(j0 + u) * (k | 1) + mm] = acc[u];
          }
        }
        __syncthreads();
        E = sl.W;
        lde = k | 1;
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
// well (dual: two matrices of ~n columns); the polynomial path keeps its two [n][nb] buffers there
int stage_apply_pcq_doubles(int k, int nv) {
  const int nb = nv + 2;
  const size_t fixed = (size_t)5 * (k + 2) + 8 * (size_t)nv + 32;
  size_t pcq = (size_t)2 * nb * (k + 4);
  const size_t pcq_poly = (size_t)2 * 16 * (size_t)(((k < kABlock ? k : kABlock) + 15) & ~15);   // poly_apply: two [rows to 16][16] buffers, any order <= min(k, 512)
  if (pcq < pcq_poly) pcq = pcq_poly;
  while ((fixed + pcq) * sizeof(double) > STAGE_APPLY_LDSCAP && pcq > 0) pcq = pcq > (size_t)nb * 64 ? pcq - (size_t)nb * 64 : 0;
  return (int)pcq;
}

// largest order the polynomial path takes: two [n to 16][16] buffers, then q [nb][n + 2], in that room
int stage_poly_max_n(int k, int nv) {
  const int nb = nv + 2;
  long n = (stage_apply_pcq_doubles(k, nv) / 32) & ~15L;   // two [n to 16][16] buffers
  if (n > kABlock) n = kABlock;                            // (4 blocks of 16 rows per wave)
  while (n > 0 && (long)nb * ((n + 2) & ~1L) > stage_apply_pcq_doubles(k, nv)) n -= 16;
  return (int)(n > 0 ? n : 0);
}

hipError_t launch_stage_apply(const StagedArgs& s, hipStream_t st) {
  const int k = s.A.k;
  const size_t fixed = (size_t)5 * (k + 2) + 8 * (size_t)s.A.nv + 32;
  const size_t pcq = (size_t)stage_apply_pcq_doubles(k, s.A.nv);
  const size_t lds = (fixed + pcq) * sizeof(double);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&letkf_stage_apply_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(letkf_stage_apply_kernel, dim3((unsigned)s.nbatch), dim3(kABlock), lds, st, s, (int)pcq);
  return hipGetLastError();
}

}  // namespace letkf
