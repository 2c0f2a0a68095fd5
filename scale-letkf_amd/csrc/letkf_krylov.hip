// letkf_krylov.hip -- the eigen-free stage of the staged path (between the Gram stage and the apply stage of
// letkf_staged.hip): everything the loop body of das_letkf (scale/letkf/letkf_tools.f90:313-527) takes from the
// eigen-decomposition of letkf_core (common/common_letkf.f90:127-206) is a FUNCTION of the point's symmetric positive
// definite matrix M applied to nb = nv + 2 vectors,
//     q_b = M^-1 t_b        b = 0, 1          (w-bar and the deterministic member's weights; :169-195)
//     q_b = g_T(M) t_b      b = 2 .. nv + 1   (the transform T applied to the perturbations x'_v; :197-216)
//     va_v = t_v^T M^-1 t_v                   (RTPS, letkf_tools.f90:1981-1990)
// with, in observation space (n < k: M = Z Z^T + c I, t_0,1 = sqrt(w) dep(_det), t_2+v = Z x'_v),
//     g_T(L) = -sqrt(k-1) / (sqrt(c) sqrt(L) (sqrt(c) + sqrt(L)))   (T = sqrt(rho) I + Z^T g_T(M) Z)
// and in member space (n >= k: M = A = Z^T Z + c I, t_0,1 = r, r_det, t_2+v = x'_v), g_T(L) = sqrt((k-1) / L).
//
// Round 2 expanded both functions in Chebyshev polynomials of M on [c, c + |S|] with a norm bound |S| >= lambda_max: the
// degree grows with sqrt(cond), the norm bound overestimates cond, and above degree 64 (cond ~ 12: ensemble spread in
// observation space of ~1.6 observation errors) the point fell back to the Jacobi.  This stage needs no bound and has no
// cliff: CONJUGATE GRADIENTS on all right-hand sides at once (one product M R per iteration on the FP64 matrix cores, the
// columns otherwise independent: their own alpha, beta) solve M q = t, and the Lanczos tridiagonal T_m that the CG
// coefficients define (T_jj = 1/alpha_j + beta_j/alpha_j-1, T_j,j+1 = sqrt(beta_j+1)/alpha_j; Lanczos vectors
// v_j = (-1)^j r_j / |r_j|) gives g_T(M) t ~ |t| V_m g_T(T_m) e_1 for the same Krylov space.  Krylov methods adapt to the
// spectrum: the few dominant modes that correlated observations create cost one or two iterations each instead of
// stretching the interval of a polynomial (spread 2.4 observation errors, cond 20-57: 28-30 iterations against a degree of
// 80-140; cond 1.5: 14 against 23), the iteration ends on its own residual |r_j| <= 1e-14 |t|, and g_T(T_m) e_1 -- m <= 128
// rows per column -- is a Chebyshev expansion on the TRIDIAGONAL matrix (three multiply-adds per row and degree, any
// degree), its interval from Gershgorin's circles of T_m itself.  The residuals r_j go to the point's slab in a
// lane-private layout (every lane reads back exactly what it wrote) and are combined once g_T(T_m) e_1 is known.
// A point that does not converge in 128 iterations (or meets a non-positive curvature: M not positive definite to
// rounding, NaN) is handed to the eigen stage, which runs after this kernel: meta's solver field is rewritten.
//
// Data flow of the product: a wave owns the 16-row blocks w, w + 8, ... (BPW of them) of the result; A operand = a 16 x 4
// tile of M straight from the slab (lane l loads M[j0 + (l >> 4)][i0 + (l & 15)]: 128 contiguous bytes per 16 lanes; loads
// of the next group of tiles are issued before the matrix-core instructions of the current one), B operand = 4 rows of
// the residual block from LDS ([row][16]: the right-hand sides padded to 16 columns; 64 consecutive doubles per read),
// D: lane l holds rows (l >> 4) + 4 r, r < 4, of column l & 15.  r, p, q = M p and x live in registers in D's layout.
// Two barriers per iteration (single-reduction CG, Chronopoulos / Gear: rho = r.r and mu = r.Mr, then
// beta = rho / rho', alpha = rho / (mu - rho beta / alpha')).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "letkf_jacobi_dev.h"
#include "letkf_staged_dev.h"

namespace letkf {

using namespace staged_dev;
using jacobi_dev::dpp_shift0;

namespace {

constexpr int kKBlock = 512;
constexpr int kMmax = 128;        // CG iterations (rows of the Lanczos tridiagonal: two per lane)
constexpr int kDcap = 4094;       // highest Chebyshev degree of the tridiagonal function (cond ~ 47000); its coefficients overlay ha | hb
// A column is converged at |r_j| <= 1e-14 |t| (A/B r3, tools/r3_ab_tol.sh: against 1e-15 one iteration less at the same parity --
// 1.5e-14 benign, 5e-14 at obs-space spread 2.4 --, +2..3 %; 1e-13 gives +5 % but 1.1e-13 on C2-slab-k100 at spread 2.4)
#ifndef KRYLOV_TOL2
#define KRYLOV_TOL2 1e-28
#endif
constexpr double kTol2 = KRYLOV_TOL2;

struct KryLds {
  double* rbuf;    // [nr16cap][16]   the residual block, B operand of the product
  double* ha;      // [kMmax][16]     alpha_j per column
  double* hb;      // [kMmax][16]     beta_j
  double* hr;      // [kMmax][16]     rho_j = |r_j|^2; later the combination coefficients
  double* red;     // [2][8][16]      per-wave partial sums
  double* cT;      // [kDcap + 2]     Chebyshev coefficients of g_T on the tridiagonals' interval: OVERLAYS ha | hb (read into registers by then)
  double* fT;      // [kDcap + 2]     g_T at the nodes
  double* swl;     // [512]           sqrt(w_i) (dual)
  double* misc;    // [32]
};

// The products of this file have one shape: D (16 rows per block x 16 right-hand sides) += A (rows x contraction) B, A a
// matrix whose ROWS are contiguous in memory (M through its symmetric view, or the observation table's rows), B a block in
// LDS ([contraction index permuted in groups of 8, see rpos][16]).  Lane (c, q) fetches row c's elements 8 t + 2 q, + 1 as
// ONE 16-byte load per block and step t -- the A operands of two matrix-core instructions whose four contraction rows
// are 8 t + 2 q' + o, q' < 4 (o = 0, then 1): 64 contiguous bytes per 4 lanes of a row.  A ring of PF loads per block is
// in flight.  Every refill is unconditional: with a branch around a load hipcc can no longer count the loads in flight
// and waits for all of them (vmcnt(0)) in front of every matrix-core step (found in the ISA); a padding step fetches
// step 0 again.  wrap: the last refills fetch the FIRST steps again (the same product is repeated: CG on one matrix).
// olim: largest element offset a lane may fetch from its row (clamped beyond: such columns meet zero rows of B).
// RES (the product of the iteration at orders <= 128, r4): the ring holds the wave's WHOLE operand (T <= PF double-steps), filled
// once per point -- no refill, the matrix stays in registers from the first iteration to the last.  The counters of the
// streaming form (profiles/r04_k100_pmc_summary.json): 1.35 MB per point = 14 iterations x the 96 KB matrix through
// L2 <-> fabric at 4.8 TB/s, the rate that bounds C3's stage too.
template <int BPW, int PF, int NA, bool RES = false>
__device__ __forceinline__ void ring_steps(d2u (&ring)[PF][BPW], const double* const (&abase)[BPW], const int olim, const int T,
                                           const bool wrap, const double* __restrict__ bbuf, const int lane, d4 (&acc)[BPW]) {
  const int Tp = (T + PF - 1) / PF * PF;
  for (int t0 = 0; t0 < Tp; t0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int t = t0 + u;
      if (t < T) {                                          // (wave-uniform)
        const double b0 = bbuf[(size_t)(8 * t) * 16 + lane], b1 = bbuf[(size_t)(8 * t + 4) * 16 + lane];
#pragma unroll
        for (int bi = 0; bi < NA; ++bi) {
          acc[bi] = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[u][bi].x, b0, acc[bi], 0, 0, 0);
          acc[bi] = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[u][bi].y, b1, acc[bi], 0, 0, 0);
        }
      }
      if constexpr (RES) continue;
      int tn = t + PF;
      if (tn >= Tp) tn = wrap ? tn - Tp : 0;
      if (tn >= T) tn = 0;
      int off = 8 * tn;
      off = off < olim ? off : olim;
#pragma unroll
      for (int bi = 0; bi < NA; ++bi) ring[u][bi] = *reinterpret_cast<const d2u*>(abase[bi] + off);
    }
  }
}
template <int BPW, int PF>
__device__ __forceinline__ void ring_fill(d2u (&ring)[PF][BPW], const double* const (&abase)[BPW], const int olim, const int T) {
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    int off = 8 * (u < T ? u : 0);
    off = off < olim ? off : olim;
#pragma unroll
    for (int bi = 0; bi < BPW; ++bi) ring[u][bi] = *reinterpret_cast<const d2u*>(abase[bi] + off);
  }
}
// nact (wave-uniform) of the BPW blocks are real: the instantiation without the idle ones
template <int BPW, int PF, bool RES = false>
__device__ __forceinline__ void ring_product(const int nact, d2u (&ring)[PF][BPW], const double* const (&abase)[BPW], const int olim,
                                             const int T, const bool wrap, const double* __restrict__ bbuf, const int lane, d4 (&acc)[BPW]) {
  if constexpr (BPW == 1) {
    ring_steps<BPW, PF, 1, RES>(ring, abase, olim, T, wrap, bbuf, lane, acc);
  } else if constexpr (BPW == 2) {
    if (nact == 2) ring_steps<BPW, PF, 2, RES>(ring, abase, olim, T, wrap, bbuf, lane, acc);
    else ring_steps<BPW, PF, 1, RES>(ring, abase, olim, T, wrap, bbuf, lane, acc);
  } else {
    if (nact == 4) ring_steps<BPW, PF, 4>(ring, abase, olim, T, wrap, bbuf, lane, acc);
    else if (nact == 3) ring_steps<BPW, PF, 3>(ring, abase, olim, T, wrap, bbuf, lane, acc);
    else if (nact == 2) ring_steps<BPW, PF, 2>(ring, abase, olim, T, wrap, bbuf, lane, acc);
    else ring_steps<BPW, PF, 1>(ring, abase, olim, T, wrap, bbuf, lane, acc);
  }
}
__device__ __forceinline__ int rpos(int row) { return (row & ~7) + 4 * (row & 1) + ((row & 7) >> 1); }
template <int BPW>
struct RingDepth {
  static constexpr int value = BPW == 1 ? 16 : BPW == 2 ? 8 : 2;
};

// One point.  Returns false when the point has to go to the eigen stage.
template <int BPW, int NWV>
__device__ __forceinline__ bool krylov_point(const Slab& sl, const KryLds& L, const int n, const int ldg, const int k,
                                             const int nv, const int nbr, const double shift, const bool dual, const int dcap,
                                             int* iters_out) {
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int nwv = NWV;
  const int nr16 = (n + 15) & ~15, nblk = nr16 >> 4;
  const int col = lane & 15, rq = lane >> 4;
  double rr[BPW][4], pp[BPW][4], qq[BPW][4], xx[BPW][4];
  // (the residual block in LDS is stored with its rows permuted inside every group of 8: rpos, ring_steps)
  // ---- r_0 = t, rho_0
  {
    double ps = 0.0;
#pragma unroll
    for (int bi = 0; bi < BPW; ++bi) {
      const int i0 = (wv + bi * nwv) * 16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = i0 + rq + 4 * r;
        rr[bi][r] = (row < n && col < nbr) ? sl.TT[(size_t)col * k + row] : 0.0;
        pp[bi][r] = qq[bi][r] = xx[bi][r] = 0.0;
        ps = fma(rr[bi][r], rr[bi][r], ps);
        if (i0 < nr16) L.rbuf[(size_t)rpos(row) * 16 + col] = rr[bi][r];
      }
    }
    ps += __shfl_xor(ps, 16, 64);
    ps += __shfl_xor(ps, 32, 64);
    if (lane < 16) L.red[wv * 16 + lane] = ps;
  }
  __syncthreads();
  // ---- the product's A operands (ring_steps): row i0 + c of M is column i0 + c of the slab (M is symmetric).  The ring
  // wraps around into the NEXT iteration -- M is the same every time --, so the latency of L2 / Infinity Cache is paid once
  // per point, not once per iteration.  Rows beyond n are clamped to row n - 1 (dropped at the use), columns beyond n read
  // the neighbouring slab words (finite: zeroed by the caller) against zero rows of the residual block.
#ifndef KRYLOV_RESIDENT
#define KRYLOV_RESIDENT 1
#endif
  constexpr bool RES = KRYLOV_RESIDENT && NWV == 4 && BPW == 2;   // orders <= 128: T <= 16 double-steps
  constexpr int PF = RES ? 16 : RingDepth<BPW>::value;
  const int nact_ = (nblk - wv + nwv - 1) / nwv;           // this wave's active blocks (wave-uniform), <= BPW
  const int nact = __builtin_amdgcn_readfirstlane(nact_ < 0 ? 0 : nact_ > BPW ? BPW : nact_);
  const int T = nr16 >> 3;
  const double* abase[BPW];
#pragma unroll
  for (int bi = 0; bi < BPW; ++bi) {
    const int row = (wv + bi * nwv) * 16 + col;
    abase[bi] = sl.G + (size_t)(row < n ? row : n - 1) * ldg + 2 * rq;
  }
#ifdef KRYLOV_TIMING_SKIP   // A/B twin, timing only (results invalid): the first steps' operands come from nowhere
  const int TS = T > KRYLOV_TIMING_SKIP ? KRYLOV_TIMING_SKIP : 0;
  const double* abase_[BPW];
#pragma unroll
  for (int bi = 0; bi < BPW; ++bi) abase_[bi] = abase[bi] + 8 * TS;
#define KRY_ABASE abase_
#define KRY_T (T - TS)
#define KRY_B (L.rbuf + (size_t)8 * TS * 16)
#else
#define KRY_ABASE abase
#define KRY_T T
#define KRY_B L.rbuf
#endif
  d2u ring[PF][BPW];
  ring_fill<BPW, PF>(ring, KRY_ABASE, 1 << 30, KRY_T);
  double rho_old = 1.0, alpha_old = 1.0, tn2 = 0.0;
  bool frozen = false, failed = false;
  int mj = 0, j = 0;
  for (;; ++j) {
    double rho = 0.0;
#pragma unroll
    for (int w = 0; w < NWV; ++w) rho += L.red[w * 16 + col];
    if (j == 0) tn2 = rho;
    if (!(rho == rho)) failed = true;                       // NaN anywhere in M or t
    if (!frozen && !(rho > kTol2 * tn2)) {
      frozen = true;
      mj = j;
    }
    const bool all_frozen = __ballot(frozen) == ~0ull;
    const bool any_failed = __ballot(failed) != 0ull;
#ifdef KRYLOV_TIMING_SKIP
    frozen = failed = false;
    if (j == 14) break;                                     // (timing twin: a fixed number of iterations)
#else
    if (all_frozen || any_failed || j == kMmax) break;      // (the same in every wave: all read the same sums)
#endif

    // ---- w = M r_j
    d4 acc[BPW];
#pragma unroll
    for (int bi = 0; bi < BPW; ++bi) acc[bi] = (d4){0.0, 0.0, 0.0, 0.0};
#ifdef KRYLOV_TIMING_SKIP
    if (nact > 0) {
      for (int t = 0; t < TS; ++t) {
        const double b0 = L.rbuf[(size_t)(8 * t) * 16 + lane], b1 = L.rbuf[(size_t)(8 * t + 4) * 16 + lane];
#pragma unroll
        for (int bi = 0; bi < BPW; ++bi)
          if (bi < nact) {
            const d2u a = *reinterpret_cast<const d2u*>(L.fT + ((size_t)(bi * 16 + (t & 15)) * 64 + lane) * 2);   // (any LDS words: 16 B per lane)
            acc[bi] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b0, acc[bi], 0, 0, 0);
            acc[bi] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b1, acc[bi], 0, 0, 0);
          }
      }
    }
#endif
    if (nact > 0) ring_product<BPW, PF, RES>(nact, ring, KRY_ABASE, 1 << 30, KRY_T, true, KRY_B, lane, acc);
    // ---- r_j to the history (lane-private; issued here, behind the product's loads, so that the next product's first
    // wait does not sit on these stores), mu = r . w per column
    double wreg[BPW][4];
    double pm = 0.0;
#pragma unroll
    for (int bi = 0; bi < BPW; ++bi) {
      const int blk = wv + bi * nwv;
      const int i0 = blk * 16;
      if (blk < nblk) {
        double* hj = sl.H + (((size_t)j * nblk + blk) * 4) * 64 + lane;
#pragma unroll
        for (int r = 0; r < 4; ++r) hj[r * 64] = rr[bi][r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        wreg[bi][r] = (blk < nblk && i0 + rq + 4 * r < n) ? acc[bi][r] : 0.0;
        pm = fma(rr[bi][r], wreg[bi][r], pm);
      }
    }
    pm += __shfl_xor(pm, 16, 64);
    pm += __shfl_xor(pm, 32, 64);
    if (lane < 16) L.red[128 + wv * 16 + lane] = pm;
    __syncthreads();
    double mu = 0.0;
#pragma unroll
    for (int w = 0; w < NWV; ++w) mu += L.red[128 + w * 16 + col];
    double alpha = 0.0, beta = 0.0;
    if (!frozen) {
      beta = j == 0 ? 0.0 : rho / rho_old;
      const double denom = j == 0 ? mu : mu - rho * beta / alpha_old;
      if (!(denom > 0.0)) failed = true;                    // M is not positive definite to rounding (or NaN)
      alpha = rho / denom;
      rho_old = rho;
      alpha_old = alpha;
    }
    if (wv == 0 && lane < 16) {
      L.ha[j * 16 + lane] = alpha;
      L.hb[j * 16 + lane] = beta;
      L.hr[j * 16 + lane] = rho;
    }
    // ---- the updates
    double ps = 0.0;
#pragma unroll
    for (int bi = 0; bi < BPW; ++bi) {
      const int blk = wv + bi * nwv;
      if (blk < nblk) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pp[bi][r] = fma(beta, pp[bi][r], rr[bi][r]);
          qq[bi][r] = fma(beta, qq[bi][r], wreg[bi][r]);
          xx[bi][r] = fma(alpha, pp[bi][r], xx[bi][r]);
          rr[bi][r] = fma(-alpha, qq[bi][r], rr[bi][r]);
          ps = fma(rr[bi][r], rr[bi][r], ps);
          L.rbuf[(size_t)rpos(blk * 16 + rq + 4 * r) * 16 + col] = rr[bi][r];
        }
      }
    }
    ps += __shfl_xor(ps, 16, 64);
    ps += __shfl_xor(ps, 32, 64);
    if (lane < 16) L.red[wv * 16 + lane] = ps;
    __syncthreads();
  }
  const int iters = j;
  *iters_out = iters;
#ifndef KRYLOV_TIMING_SKIP
  if (__ballot(failed) != 0ull || !(__ballot(frozen) == ~0ull)) return false;
#else
  mj = iters;
#endif

  // ================= g_T(T_m) e_1 for the columns b >= 2, m_b = mj rows each
  // (a) the tridiagonals: wave w takes the columns 2 + w, 2 + w + NWV, ...; lane l holds rows 2l and 2l + 1
  // PACKED (r4): with at most 2 GS iterations (GS = 64 / NC lanes) the wave's NC columns run side by side, one per group of GS
  // lanes, instead of one after the other on 64 lanes of which m / 2 are busy -- at MEMBER = 100 (14 iterations, four columns per
  // wave) the recurrence below was a third of this kernel's instructions.  The wave-wide shifts stay: at a group's first and
  // last row the off-diagonal coefficient that meets the neighbouring group's value is zero.
  constexpr int NC = 16 / NWV;
  constexpr int GS = 64 / NC;
#ifndef KRYLOV_PACKED
#define KRYLOV_PACKED 1
#endif
  const bool packed = KRYLOV_PACKED && NC > 1 && iters <= 2 * GS;   // (wave-uniform)
  const int ncl = packed ? 1 : NC;                                   // passes over the columns
  const int rl = packed ? (lane & (GS - 1)) : lane;                  // row pair of this lane
  double a0[NC], a1[NC], bL[NC], bM[NC], bR[NC];
  int mb[NC];
  double gmax = 0.0;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int b = 2 + wv + NWV * (packed ? lane / GS : c);
    mb[c] = 0;
    a0[c] = a1[c] = bL[c] = bM[c] = bR[c] = 0.0;
    const int mcol = __shfl(mj, b < nbr ? b : 0, 64);      // lane b (rq = 0) carries column b's count
    if (c < ncl && b < nbr) {                               // (wave-uniform unless packed)
      mb[c] = mcol;
      const int m = mb[c];
      auto diag = [&](int jj) -> double {
        if (jj >= m) return 0.0;
        double d = 1.0 / L.ha[jj * 16 + b];
        if (jj > 0) d += L.hb[jj * 16 + b] / L.ha[(jj - 1) * 16 + b];
        return d;
      };
      auto offd = [&](int jj) -> double {                   // between rows jj and jj + 1
        if (jj < 0 || jj + 1 >= m) return 0.0;
        return sqrt(L.hb[(jj + 1) * 16 + b]) / L.ha[jj * 16 + b];
      };
      a0[c] = diag(2 * rl);
      a1[c] = diag(2 * rl + 1);
      bL[c] = offd(2 * rl - 1);
      bM[c] = offd(2 * rl);
      bR[c] = offd(2 * rl + 1);
      gmax = fmax(gmax, fmax(a0[c] + bL[c] + bM[c], a1[c] + bM[c] + bR[c]));
    }
  }
#pragma unroll
  for (int mk = 1; mk < 64; mk <<= 1) gmax = fmax(gmax, __shfl_xor(gmax, mk, 64));
  if (lane == 0) L.misc[wv] = gmax;
  __syncthreads();
  double hi = 0.0;
#pragma unroll
  for (int w = 0; w < NWV; ++w) hi = fmax(hi, L.misc[w]);
  // (b) interval, degree, Chebyshev coefficients of g_T.  Ritz values lie in [lambda_min(M), lambda_max(M)], and
  // lambda_min(M) >= c; the margin below c covers the rounding of the recurrence
  const double lo = shift * (1.0 - 1e-9);
  hi = fmax(hi * (1.0 + 1e-12), shift * (1.0 + 1e-6));
  const double sk = sqrt(hi / lo), rate = (sk - 1.0) / (sk + 1.0);
  int deg = (int)ceil(log(1e-17) / log(rate)) + 2;
  if (!(deg >= 4)) deg = 4;
  if (deg > dcap) return false;                             // (uniform: hi is the same in every thread)
  const int N = deg + 1;
  const double half = 0.5 * (hi - lo), mid = 0.5 * (hi + lo), inv = 1.0 / half;
  const double sqc = sqrt(shift), sqkm1 = sqrt((double)(k - 1));
  for (int jn = tid; jn < N; jn += nthr) {
    const double Lm = fma(half, cospi(((double)jn + 0.5) / (double)N), mid);
    const double sL = sqrt(Lm);
    L.fT[jn] = dual ? -sqkm1 / (sqc * sL * (sqc + sL)) : sqkm1 / sL;
  }
  __syncthreads();
  for (int i = tid; i < N; i += nthr) {
    // c_i = (2 - [i = 0]) / N sum_j g(x_j) cos(pi i (j + 1/2) / N); the cosines by rotation, re-seeded every 32 nodes
    const double th = (double)i / (double)N;
    const double cd = cospi(th), sd = sinpi(th);
    double s_ = 0.0, cc = 0.0, ss = 0.0;
    for (int jn = 0; jn < N; ++jn) {
      if ((jn & 31) == 0) {
        const double arg = (double)i * ((double)jn + 0.5) / (double)N;
        cc = cospi(arg);
        ss = sinpi(arg);
      }
      s_ = fma(L.fT[jn], cc, s_);
      const double c2 = fma(cc, cd, -ss * sd);
      ss = fma(ss, cd, cc * sd);
      cc = c2;
    }
    L.cT[i] = s_ * ((i == 0 ? 1.0 : 2.0) / (double)N);
  }
  __syncthreads();
  // (c) y = sum_d c_d T_d(T~) e_1, T~ = (T_m - mid) / half; then the combination coefficients y_j (-1)^j |t| / |r_j|
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int b = 2 + wv + NWV * (packed ? lane / GS : c);
    if (c < ncl && b < nbr) {
      const int m = mb[c];
      const double d0 = (a0[c] - mid) * inv, d1 = (a1[c] - mid) * inv;
      const double eL = bL[c] * inv, eM = bM[c] * inv, eR = bR[c] * inv;
      const bool r0 = 2 * rl < m, r1 = 2 * rl + 1 < m;
      double u0 = rl == 0 && m > 0 ? 1.0 : 0.0, u1 = 0.0;     // T_0 e_1
      double y0 = L.cT[0] * u0, y1 = 0.0;
      // T_1
      double nL = dpp_shift0<0x138>(u1), nR = dpp_shift0<0x130>(u0);
      double v0 = r0 ? fma(d0, u0, fma(eM, u1, eL * nL)) : 0.0;
      double v1 = r1 ? fma(d1, u1, fma(eM, u0, eR * nR)) : 0.0;
      y0 = fma(L.cT[1], v0, y0);
      y1 = fma(L.cT[1], v1, y1);
      const double d0_2 = 2.0 * d0, d1_2 = 2.0 * d1, eL2 = 2.0 * eL, eM2 = 2.0 * eM, eR2 = 2.0 * eR;
      for (int d = 2; d <= deg; ++d) {
        nL = dpp_shift0<0x138>(v1);
        nR = dpp_shift0<0x130>(v0);
        const double t0 = r0 ? fma(d0_2, v0, fma(eM2, v1, fma(eL2, nL, -u0))) : 0.0;
        const double t1 = r1 ? fma(d1_2, v1, fma(eM2, v0, fma(eR2, nR, -u1))) : 0.0;
        u0 = v0;
        u1 = v1;
        v0 = t0;
        v1 = t1;
        const double cdv = L.cT[d];
        y0 = fma(cdv, v0, y0);
        y1 = fma(cdv, v1, y1);
      }
      const double t2 = L.hr[b];                             // rho_0 = |t|^2
      const double rj0 = r0 ? L.hr[(2 * rl) * 16 + b] : 1.0, rj1 = r1 ? L.hr[(2 * rl + 1) * 16 + b] : 1.0;
      if (2 * rl < iters) L.hr[(2 * rl) * 16 + b] = r0 ? y0 * sqrt(t2 / rj0) : 0.0;
      if (2 * rl + 1 < iters) L.hr[(2 * rl + 1) * 16 + b] = r1 ? -y1 * sqrt(t2 / rj1) : 0.0;
    }
  }
  __syncthreads();
  // (d) x_T = sum_j coef_j r_j from the history
  double xt[BPW][4];
#pragma unroll
  for (int bi = 0; bi < BPW; ++bi)
#pragma unroll
    for (int r = 0; r < 4; ++r) xt[bi][r] = 0.0;
  if (col >= 2 && col < nbr) {
#pragma unroll
    for (int bi = 0; bi < BPW; ++bi) {
      const int blk = wv + bi * nwv;
      if (blk < nblk) {
        for (int j0 = 0; j0 < iters; j0 += 4) {              // four iterations' loads ahead of their multiply-adds
          double hv[4][4], cf[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int jj = j0 + u < iters ? j0 + u : iters - 1;
            cf[u] = j0 + u < iters ? L.hr[jj * 16 + col] : 0.0;
            const double* hj = sl.H + (((size_t)jj * nblk + blk) * 4) * 64 + lane;
#pragma unroll
            for (int r = 0; r < 4; ++r) hv[u][r] = hj[r * 64];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) xt[bi][r] = fma(cf[u], hv[u][r], xt[bi][r]);
        }
      }
    }
  }
  // ---- q_b (dual: QQ, read by the Z^T q pass of the apply stage; primal: OUT itself) and the quadratic forms
  double* qout = dual ? sl.QQ : sl.OUT;
  double pq = 0.0;
#pragma unroll
  for (int bi = 0; bi < BPW; ++bi) {
    const int i0 = (wv + bi * nwv) * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i0 + rq + 4 * r;
      if (row < n && col < nbr) {
        pq = fma(sl.TT[(size_t)col * k + row], xx[bi][r], pq);
        qout[(size_t)col * k + row] = col < 2 ? xx[bi][r] : xt[bi][r];
      }
    }
  }
  pq += __shfl_xor(pq, 16, 64);
  pq += __shfl_xor(pq, 32, 64);
  if (lane < 16) L.red[wv * 16 + lane] = pq;
  __syncthreads();
  for (int v = tid; v < nv; v += nthr) {
    double s_ = 0.0;
    for (int w = 0; w < nwv; ++w) s_ += L.red[w * 16 + 2 + v];
    sl.PC[v] = s_;                                          // va_v = t_v^T M^-1 t_v
  }
  return true;
}

}  // namespace

// LDS: rbuf [16 nr16cap] | ha | hb | hr [kMmax 16 each] | red [256] | fT [kDcap + 2] | swl [512] | misc [32]; cT = ha | hb
// (the staging copy of X' for the Z x' pass of an observation-space point overlays rbuf .. hr)
// BPW: 16-row blocks per wave = the orders the instantiation takes (1: m <= 128, 2: <= 256, 4: <= 512); a launch passes over
// the points of the other classes, so that each class has the registers of its own kernel
// NWV: waves of the workgroup.  8 (one workgroup per CU) for the orders above 128; orders up to 128 -- MEMBER = 100, the
// reference's most common ensemble size -- run as FOUR waves with two 16-row blocks each and TWO workgroups per CU (r4): the
// iteration is latency-bound (two barriers, two LDS reductions, 28 matrix instructions per wave), and with one workgroup per
// CU a third of the wave cycles were waiting with the vector ALU 19 % busy (profiles/r03_k100_pmc_summary.json) -- the second
// point fills them.  Same registers per wave (256: two waves per SIMD either way), LDS <= 80 KB per workgroup through a smaller
// cap of the tridiagonal function's degree (dcap: cond(T_m) ~ 5000 instead of 47000; beyond it the point goes to the eigen
// stage as before).
template <int BPW, int NWV>
__global__ void __launch_bounds__(64 * NWV, NWV == 4 ? 2 : 1) letkf_stage_krylov_kernel(const StagedArgs S, const int nr16cap, const int r0,
                                                                                         const int dcap, const int cls) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const PointArgs& A = S.A;
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int nwv = NWV;
  const int k = A.k, nv = A.nv, nb = nv + 2;
  KryLds L;
  L.rbuf = smem;
  L.ha = L.rbuf + (size_t)16 * nr16cap;
  L.hb = L.ha + kMmax * 16;
  L.hr = L.hb + kMmax * 16;
  L.red = smem + r0;                                         // (r0 >= 16 nr16cap + 3 kMmax 16)
  L.cT = L.ha;                                               // (2 kMmax 16 = kDcap + 2 doubles)
  L.fT = L.red + 256;
  L.swl = L.fT + (dcap + 2);
  L.misc = L.swl + 512;
  double* xl = smem;                                         // X' block of the Z x' product: overlays rbuf .. hr (r0 doubles)

  for (long it = blockIdx.x; it < S.nbatch; it += gridDim.x) {
    const long pt = S.pt0 + it;
    const int meta0 = S.meta[2 * it], m = S.meta[2 * it + 1];
    if ((meta0 >> 8) != 3) continue;                         // (uniform for the workgroup)
    if ((m <= 128 ? 0 : m <= 256 ? 1 : 2) != cls) continue;  // (the launch's class of orders: launch_stage_krylov)
    Slab sl = slab_of(A.ws + (size_t)it * A.ws_per_block, k, nv, S.kkout);
    const bool dual = (meta0 & 0xff) == 2;
    const double shift = sl.SC[3];
    const double* g0 = A.gues + pt * A.sp;
    ObsView ov;
    ov.A = &A;
    ov.pt = pt;
    ov.o0 = A.obs_off[pt];
    ov.n = (int)(A.obs_off[pt + 1] - ov.o0);
    const int n = ov.n;
    __syncthreads();                                         // (LDS of the previous point is free)
    // ---- right-hand sides in the solver's space: TT[b][.]
    if (!dual) {
      for (int e = tid; e < 2 * k; e += nthr) sl.TT[e] = e < k ? sl.V0[e] : sl.V1[e - k];
      for (int e = tid; e < nv * k; e += nthr) {
        const int v = e / k, mm = e - v * k;
        sl.TT[2 * k + e] = g0[mm * A.sm + v * A.sv];
      }
    } else {
      for (int i = tid; i < n; i += nthr) {
        double w, d, dd, rl;
        ov.weights(i, w, d, dd, rl);
        L.swl[i] = sqrt(w);
      }
      for (int e = tid; e < 2 * n; e += nthr) sl.TT[(size_t)(e < n ? 0 : 1) * k + (e < n ? e : e - n)] = e < n ? sl.V0[e] : sl.V1[e - n];
      // TT[2 + v][i] = sqrt(w_i) sum_mm Y[i][mm] x'_v[mm] on the matrix cores (ring_steps): A = the local observations'
      // rows of the table (members contiguous), B = X' in LDS, [member, permuted][16] with x'_v in column 2 + v, KC members
      // at a time (the block overlays the iteration's LDS arrays).  (First version: a wave per observation row with 11 wave
      // reductions each and nothing in flight behind the row being summed: ~60 us of a point's ~200 us outside the
      // iterations at n = 200, k = 320.)
      constexpr int PF = RingDepth<BPW>::value;
      const int col = lane & 15, rq = lane >> 4;
      const int nblk = (n + 15) >> 4;
      const int nact_ = (nblk - wv + nwv - 1) / nwv;
      const int nact = __builtin_amdgcn_readfirstlane(nact_ < 0 ? 0 : nact_ > BPW ? BPW : nact_);
      const double* zrow[BPW];
#pragma unroll
      for (int bi = 0; bi < BPW; ++bi) {
        const int row = (wv + bi * nwv) * 16 + col;
        long ms;
        zrow[bi] = ov.row(row < n ? row : n - 1, ms) + 2 * rq;
      }
      int KC = (r0 / 16) & ~15;
      if (KC > ((k + 15) & ~15)) KC = (k + 15) & ~15;
      d4 zacc[BPW];
#pragma unroll
      for (int bi = 0; bi < BPW; ++bi) zacc[bi] = (d4){0.0, 0.0, 0.0, 0.0};
      for (int c0 = 0; c0 < k; c0 += KC) {
        const int kc = k - c0 < KC ? k - c0 : KC, kc16 = (kc + 15) & ~15;
        __syncthreads();                                     // (the previous chunk's readers; swl / TT writes above)
        for (int e = tid; e < kc16 * 16; e += nthr) {
          const int c = e / kc16, row = e - c * kc16;        // consecutive threads walk down the members of one variable
          xl[(size_t)rpos(row) * 16 + c] = (row < kc && c >= 2 && c < nb) ? g0[(long)(c0 + row) * A.sm + (c - 2) * A.sv] : 0.0;
        }
        __syncthreads();
        if (nact > 0) {
          const double* zb[BPW];
#pragma unroll
          for (int bi = 0; bi < BPW; ++bi) zb[bi] = zrow[bi] + c0;
          const int olim = k - 1 - c0 - 2 * rq;              // (a row holds k + 1 doubles: the last pair fetched is [k - 1, k])
          d2u zring[PF][BPW];
          ring_fill<BPW, PF>(zring, zb, olim, kc16 >> 3);
          ring_product<BPW, PF>(nact, zring, zb, olim, kc16 >> 3, false, xl, lane, zacc);
        }
      }
#pragma unroll
      for (int bi = 0; bi < BPW; ++bi) {
        const int i0 = (wv + bi * nwv) * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = i0 + rq + 4 * r;
          if (row < n && col >= 2 && col < nb) sl.TT[(size_t)col * k + row] = zacc[bi][r] * L.swl[row];
        }
      }
    }
    const int ldg = staged_ld(m);
    if (tid < 16) sl.G[(size_t)m * ldg + tid] = 0.0;         // (read as padding columns of the last rows: must be finite)
    if (ldg > m + 1) {                                       // rows padded to the alignment: the padding is read against zero rows of the residual block
      const int npad = ldg - m;
      for (int e = tid; e < m * npad; e += nthr) sl.G[(size_t)(e / npad) * ldg + m + e % npad] = 0.0;
    }
    __syncthreads();
    int iters = 0;
    const bool ok = krylov_point<BPW, NWV>(sl, L, m, ldg, k, nv, nb, shift, dual, dcap, &iters);
    if (tid == 0) {
      if (ok) {
        S.info[2 * it] = -iters;                             // nsweep reports -(iterations) ...
        S.info[2 * it + 1] = 1;                              // ... and the point is converged
      } else {
        S.meta[2 * it] = (meta0 & 0xff) | ((m <= S.wg_max_order ? 1 : 2) << 8);   // to the eigen stage (runs next)
      }
    }
  }
}

int stage_krylov_max_n(int k) { return k < kKBlock ? (k + 15) & ~15 : kKBlock; }
int stage_krylov_max_iter() { return kMmax; }

long stage_krylov_hist_doubles(int k) { return (long)kMmax * 16 * stage_krylov_max_n(k); }

hipError_t launch_stage_krylov(const StagedArgs& s, size_t lds_max, hipStream_t st) {
  const int k = s.A.k;
  const int nr16cap = stage_krylov_max_n(k);
  static_assert(kDcap + 2 <= 2 * kMmax * 16, "the Chebyshev coefficients overlay ha | hb");
  const size_t over = (size_t)16 * nr16cap + 3 * (size_t)kMmax * 16;   // rbuf .. hr: what the staging copy of X' may overlay
  const size_t budget = (lds_max > 160 * 1024 ? 160 * 1024 : lds_max) - 1024;
  const size_t r0 = over;
  auto lds_of = [&](int dcap) { return ((size_t)256 + (size_t)(dcap + 2) + 512 + 32 + r0) * sizeof(double); };
  if (lds_of(kDcap) > budget) return hipErrorInvalidValue;
  auto go = [&](auto kern, int nthr, int dcap, int cls) -> hipError_t {
    const size_t lds = lds_of(dcap);
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)s.nbatch), dim3(nthr), lds, st, s, nr16cap, (int)r0, dcap, cls);
    return hipGetLastError();
  };
  // orders <= 128: two workgroups of four waves per CU where half of the CU's LDS leaves the tridiagonal function a degree
  // worth having (KRYLOV_CLS0_WAVES = 8: the one-workgroup form, for A/B twins)
#ifndef KRYLOV_CLS0_WAVES
#define KRYLOV_CLS0_WAVES 4
#endif
  const long half = (long)(budget + 1024) / 2 / (long)sizeof(double) - (long)(256 + 512 + 32 + 2) - (long)r0;
  const int dcap0 = half > kDcap ? kDcap : (int)half;
  hipError_t e;
  if (KRYLOV_CLS0_WAVES == 4 && dcap0 >= 512) e = go(&letkf_stage_krylov_kernel<2, 4>, 256, dcap0, 0);
  else e = go(&letkf_stage_krylov_kernel<1, 8>, kKBlock, kDcap, 0);
  if (e == hipSuccess && nr16cap > 128) e = go(&letkf_stage_krylov_kernel<2, 8>, kKBlock, kDcap, 1);
  if (e == hipSuccess && nr16cap > 256) e = go(&letkf_stage_krylov_kernel<4, 8>, kKBlock, kDcap, 2);
  return e;
}

}  // namespace letkf
