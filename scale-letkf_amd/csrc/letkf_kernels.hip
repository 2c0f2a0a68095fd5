// letkf_kernels.hip -- gfx950 (MI355X / CDNA4) device code of the LETKF analysis core.
//
// One workgroup solves one grid point end to end (the reference does the same work in
// common/common_letkf.f90:52-257 + scale/letkf/letkf_tools.f90:457-513, one OpenMP
// thread per point):
//
//   1. gather the point's local observations (rows of obsda_sort%ensval, member-fastest)
//      through LDS tiles, scaled by sqrt(1/rdiag)              [letkf_tools.f90:1463, common_letkf.f90:111-123]
//   2. A = Ys^T Ys + (k-1)/rho I in LDS, r = Ys^T (sqrt(w) dep) [common_letkf.f90:127-143, 169-195 folded]
//   3. eigen-decomposition of A by a wavefront-cooperative one-sided (Hestenes) Jacobi
//      iteration on G = A held in LDS: columns of G converge to v_j * lambda_j
//      (replaces common_mtx.f90:41 -> EISPACK rs/tred2/tql2, netlib.f:524)
//   4. w-bar = V L^-1 V^T r, T = V sqrt((k-1)/L) V^T, Pa = V L^-1 V^T applied directly to the
//      k x nv perturbation slab (T/Pa are only materialised when the caller asks for them)
//      + RTPP / RTPS relaxation, beta blending, deterministic member, q-spread clamp
//                                                               [common_letkf.f90:151-227, letkf_tools.f90:457-513]
//
// No k x k matrix ever touches HBM on the production path; per point the kernel streams
// n*k*8 B of obs rows + 2*k*nv*8 B of state.
//
// Template parameters:
//   BIG   false: every per-point array lives in LDS (k <= 128);  true: G, U, X live in a
//         per-workgroup HBM/L2 workspace (large-k spill path, any k).
//   RMAX  rows of a column pair cached in registers per lane in the Jacobi step
//         (8 lanes per pair: RMAX >= ceil(k/8));  0 selects the streaming variant
//         (64 lanes per pair, nothing cached) used by the BIG path.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "letkf_device.h"
#include "letkf_jacobi_dev.h"
#include "letkf_staged_dev.h"

namespace letkf {

// ------------------------------------------------------------------ small helpers
__device__ __forceinline__ double shfl_xor_d(double v, int mask) { return __shfl_xor(v, mask, 64); }

template <int W>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
  for (int m = 1; m < W; m <<= 1) v += shfl_xor_d(v, m);
  return v;
}

// DPP move of a double (VALU, no LDS crossbar): 0xB1 quad_perm[1,0,3,2], 0x4E quad_perm[2,3,0,1], 0x141 row_half_mirror
template <int CTRL>
__device__ __forceinline__ double dpp_mov_d(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// sum over aligned groups of 8 lanes: lane^1, lane^2, then the mirror image inside the 8 (after the first two
// steps every lane of a quad holds the quad sum, so "7 - lane" is as good as lane^4)
__device__ __forceinline__ double group8_sum(double v) {
  v += dpp_mov_d<0xB1>(v);
  v += dpp_mov_d<0x4E>(v);
  v += dpp_mov_d<0x141>(v);
  return v;
}
__device__ __forceinline__ double frsqrt2(double x) {   // v_rsq_f64 seed + 2 Newton steps
  double y = __builtin_amdgcn_rsq(x);
  double e = fma(-x * y, y, 1.0);
  y = fma(y * 0.5, e, y);
  e = fma(-x * y, y, 1.0);
  return fma(y * 0.5, e, y);
}
__device__ __forceinline__ double frcp2(double x) {     // v_rcp_f64 seed + 2 Newton steps
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  return fma(r, e, r);
}

// Bijective XCD-aware remap (blocks b and b+8 share an XCD and its L2): consecutive logical
// work items go to the same XCD so neighbouring grid points, which read almost the same obs
// rows, hit the same L2.  Speed only, never correctness.
__device__ __forceinline__ long xcd_remap(long orig, long n) {
  const long q = n >> 3, r = n & 7;
  const long xcd = orig & 7, j = orig >> 3;
  const long base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + j;
}

// round-robin (circle method) tournament: player m-1 fixed, step s in [0, m-1)
__device__ __forceinline__ void rr_pair(int m, int s, int pi, int& p, int& q) {
  const int mm = m - 1;
  if (pi == 0) {
    p = mm;
    q = s;
  } else {
    p = s + pi;
    if (p >= mm) p -= mm;
    q = s - pi;
    if (q < 0) q += mm;
  }
}

// Hestenes rotation that orthogonalises two columns with squared norms a, b and inner product g.
// tan(2 theta) = 2 g / (b - a);  t = 2 g sgn(d) / (|d| + sqrt(d^2 + 4 g^2)); hardware rsq/rcp seeds + Newton instead
// of the IEEE division / sqrt expansions (3 + 2 of them were most of the step's FP64 issue slots)
__device__ __forceinline__ void hestenes_cs(double a, double b, double g, double& c, double& s) {
  const double d = b - a;
  const double x = fma(d, d, 4.0 * g * g);
  const double hh = x * frsqrt2(x);
  const double t = (2.0 * g) * copysign(1.0, d) * frcp2(fabs(d) + hh);
  c = frsqrt2(fma(t, t, 1.0));
  s = c * t;
}

constexpr double kRotTol2 = 1e-30;   // rotate when gamma^2 > kRotTol2 * alpha * beta  (|cos| > 1e-15)
constexpr double kStopTol2 = 1e-20;  // converged when a whole sweep saw only |cos| <= 1e-10 (all still rotated away in it)
constexpr int kMaxSweep = 60;

// ------------------------------------------------------------------ Jacobi, 8 lanes per pair, rows in registers
template <int RMAX>
__device__ __forceinline__ int jacobi_cached(double* __restrict__ G, const int ldg, const int k, const int max_sweep,
                                             int& conv) {
  const int tid = threadIdx.x;
  const int l8 = tid & 7;
  const int grp = tid >> 3;
  const int ngrp = blockDim.x >> 3;
  const int m = k + (k & 1);
  const int npairs = m >> 1;
  int sweep = 0;
  for (; sweep < max_sweep; ++sweep) {
    int notconv = 0;
    for (int s = 0; s < m - 1; ++s) {
      for (int pi = grp; pi < npairs; pi += ngrp) {
        int p, q;
        rr_pair(m, s, pi, p, q);
        if (p < k && q < k) {
          double* gp = G + (size_t)p * ldg;
          double* gq = G + (size_t)q * ldg;
          double a[RMAX], b[RMAX];
          double al = 0.0, be = 0.0, ga = 0.0;
#pragma unroll
          for (int r = 0; r < RMAX; ++r) {
            const int row = l8 + 8 * r;
            const bool ok = row < k;
            a[r] = ok ? gp[row] : 0.0;
            b[r] = ok ? gq[row] : 0.0;
            al = fma(a[r], a[r], al);
            be = fma(b[r], b[r], be);
            ga = fma(a[r], b[r], ga);
          }
          al = group8_sum(al);
          be = group8_sum(be);
          ga = group8_sum(ga);
          const double g2 = ga * ga, ab = al * be;
          if (g2 > kStopTol2 * ab) notconv = 1;
          if (g2 > kRotTol2 * ab) {
            double c, sn;
            hestenes_cs(al, be, ga, c, sn);
#pragma unroll
            for (int r = 0; r < RMAX; ++r) {
              const int row = l8 + 8 * r;
              if (row < k) {
                gp[row] = c * a[r] - sn * b[r];
                gq[row] = sn * a[r] + c * b[r];
              }
            }
          }
        }
      }
      __syncthreads();
    }
    if (!__syncthreads_or(notconv)) {
      ++sweep;
      conv = 1;
      break;
    }
  }
  return sweep;
}

// ------------------------------------------------------------------ Jacobi, one wave per pair, streaming (any k, G anywhere)
__device__ __forceinline__ int jacobi_stream(double* __restrict__ G, const int ldg, const int k, const int max_sweep,
                                             int& conv) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int grp = tid >> 6;
  const int ngrp = blockDim.x >> 6;
  const int m = k + (k & 1);
  const int npairs = m >> 1;
  int sweep = 0;
  for (; sweep < max_sweep; ++sweep) {
    int notconv = 0;
    for (int s = 0; s < m - 1; ++s) {
      for (int pi = grp; pi < npairs; pi += ngrp) {
        int p, q;
        rr_pair(m, s, pi, p, q);
        if (p < k && q < k) {
          double* gp = G + (size_t)p * ldg;
          double* gq = G + (size_t)q * ldg;
          double al = 0.0, be = 0.0, ga = 0.0;
          for (int row = lane; row < k; row += 64) {
            const double x = gp[row], y = gq[row];
            al = fma(x, x, al);
            be = fma(y, y, be);
            ga = fma(x, y, ga);
          }
          al = group_sum<64>(al);
          be = group_sum<64>(be);
          ga = group_sum<64>(ga);
          const double g2 = ga * ga, ab = al * be;
          if (g2 > kStopTol2 * ab) notconv = 1;
          if (g2 > kRotTol2 * ab) {
            double c, sn;
            hestenes_cs(al, be, ga, c, sn);
            for (int row = lane; row < k; row += 64) {
              const double x = gp[row], y = gq[row];
              gp[row] = c * x - sn * y;
              gq[row] = sn * x + c * y;
            }
          }
        }
      }
      __syncthreads();
    }
    if (!__syncthreads_or(notconv)) {
      ++sweep;
      conv = 1;
      break;
    }
  }
  return sweep;
}

// ------------------------------------------------------------------ block Jacobi on the matrix cores (large k)
// G (k x k, column-major, leading dimension ldg) lives in the workgroup's HBM/L2 workspace.  Columns are grouped in
// blocks of 16; a round of the round-robin tournament over the blocks gives every wave one BLOCK PAIR (32 columns):
//   1. B = Y^T Y (32 x 32) with v_mfma_f64_16x16x4, Y's rows streamed from the workspace (3 tiles, k/4 steps);
//   2. the eigenvectors V of B by the in-register row-split Jacobi (letkf_jacobi_dev.h, its 32-column instance) --
//      one-sided on B: the columns come out as mu_j v_j, normalised to V;
//   3. Y <- Y V, again on the matrix cores, 16 rows at a time, in place.
// Against the streaming version (one wave per column pair): 1/11 of the L2 traffic per sweep at k = 320, the pair
// arithmetic on the matrix pipe, three 64-lane shuffle reductions per column pair gone, fewer outer sweeps.
// A sweep counts as converged when none of its block pairs needed more than the verification cycle of the inner solve.
// wscr: kBlkScr doubles of LDS per wave (512 of the inner solver + the packed 32 x 32 triangular factor).
constexpr int kBlkScr = kBlockJacobiScratch;
// (the inner solves stop at the point kernels' |cos| <= 1e-12: with round 2's 1e-10 the block iteration alone "converges" at
// cond(A) ~ 1e2 with an analysis 180 cond eps off the oracle; at 1e-12 it hands over to the scalar iteration there: 1.6 cond eps)
constexpr int kBlkFlat = 3, kBlkSweepCap = 16;
__device__ __forceinline__ int jacobi_block_mfma(double* __restrict__ G, const int ldg, const int k, const int max_sweep,
                                                 double* scr_all, int& conv) {
  using jacobi_dev::v4d;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6, nwv = blockDim.x >> 6;
  const int q = lane >> 4, c16 = lane & 15;
  double* scr = scr_all + (size_t)wv * kBlkScr;
  const int nblk = (k + 15) >> 4;
  const int nbe = nblk + (nblk & 1);                     // even number of players (one phantom block if needed)
  const int ntile = nblk;                                // 16-row tiles of Y
  int sweep = 0, flat = 0;
  for (; sweep < max_sweep; ++sweep) {
    int notconv = 0, big = 0;
    for (int rnd = 0; rnd < nbe - 1; ++rnd) {
      for (int pi = wv; pi < (nbe >> 1); pi += nwv) {
        int I, J;
        rr_pair(nbe, rnd, pi, I, J);
        if (I > J) {
          const int t_ = I;
          I = J;
          J = t_;
        }
        if (J >= nblk) continue;                         // paired with the phantom block: idle this round
        const int colI = 16 * I + c16, colJ = 16 * J + c16;
        const bool okI = colI < k, okJ = colJ < k;
        const double* gI = G + (size_t)(okI ? colI : 0) * ldg;
        const double* gJ = G + (size_t)(okJ ? colJ : 0) * ldg;
        // ---- 1. Gram of the 32 columns
        v4d tII{0.0, 0.0, 0.0, 0.0}, tIJ{0.0, 0.0, 0.0, 0.0}, tJJ{0.0, 0.0, 0.0, 0.0};
        {
          // four 4-row steps in flight (the rows come from L2: one step of MFMAs does not cover a load)
          // (every load is unconditional, from a clamped address, and zeroed afterwards by a select: written as
          // `cond ? g[rn] : 0.0` hipcc wraps each load in its own exec-mask branch and, unable to count the loads in
          // flight across the branches, waits s_waitcnt vmcnt(0) in front of every step -- found in the ISA)
          constexpr int PD = 4;
          double yI[PD], yJ[PD];
          const int kl = k - 1;
#pragma unroll
          for (int u = 0; u < PD; ++u) {
            const int rn = 4 * u + q;
            yI[u] = gI[min(rn, kl)];
            yJ[u] = gJ[min(rn, kl)];
            asm volatile("" ::: "memory");
          }
          for (int r0 = 0; r0 < k; r0 += 4 * PD) {
#pragma unroll
            for (int u = 0; u < PD; ++u) {
              // (the raw rows are routed through an empty asm right here: placed next to the load, the select makes the
              // wave wait for each load pair in turn -- four L2 latencies per four steps)
              asm volatile("" : "+v"(yI[u]), "+v"(yJ[u])::"memory");
              const int rc = r0 + 4 * u + q;
              const double cI = (okI && rc < k) ? yI[u] : 0.0, cJ = (okJ && rc < k) ? yJ[u] : 0.0;
              const int rn = rc + 4 * PD;
              yI[u] = gI[min(rn, kl)];
              yJ[u] = gJ[min(rn, kl)];
              tII = __builtin_amdgcn_mfma_f64_16x16x4f64(cI, cI, tII, 0, 0, 0);
              tIJ = __builtin_amdgcn_mfma_f64_16x16x4f64(cI, cJ, tIJ, 0, 0, 0);
              tJJ = __builtin_amdgcn_mfma_f64_16x16x4f64(cJ, cJ, tJJ, 0, 0, 0);
            }
          }
        }
        // ---- B -> "lane j owns column j" (lanes 0..31), 16 columns at a time through scr[16][32]
        double g[32];
#pragma unroll
        for (int r = 0; r < 32; ++r) g[r] = 0.0;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          jacobi_dev::wave_lds_sync();
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int a = q + 4 * reg;
            if (half == 0) {
              scr[c16 * 32 + a] = tII[reg];              // B[a][c16]
              scr[a * 32 + 16 + c16] = tIJ[reg];         // B[16 + c16][a] = IJ[a][c16]
            } else {
              scr[c16 * 32 + a] = tIJ[reg];              // B[a][16 + c16]
              scr[c16 * 32 + 16 + a] = tJJ[reg];         // B[16 + a][16 + c16]
            }
          }
          jacobi_dev::wave_lds_sync();
          if ((lane >> 4) == half) {
#pragma unroll
            for (int r = 0; r < 32; ++r) g[r] = scr[c16 * 32 + r];
          }
        }
        jacobi_dev::wave_lds_sync();
        // phantom columns (beyond k) are zero in Y: give them a unit diagonal so that they stay unit vectors of V
        {
          const int mycol = (lane < 16) ? 16 * I + lane : 16 * J + (lane - 16);
          const bool pad = lane < 32 && mycol >= k;
#pragma unroll
          for (int r = 0; r < 32; ++r)
            if (pad && r == lane) g[r] = 1.0;
        }
        // ---- 2. eigenvectors of B, through its Cholesky factor.  Run on B's own columns the one-sided iteration sees the
        // cosines of Y's columns multiplied by lambda_i / lambda_j + lambda_j / lambda_i (B is a Gram matrix: squared
        // condition) and its stop rule drowns in rounding noise from cond(A) ~ 1e3 on (tools/r3_probe_block_jacobi.py).
        // With B = R^T R the columns of R have exactly the inner products of Y's columns, at 32 rows instead of k: the
        // iteration turns R into W = R V (orthogonal columns), and V = R^T W Lambda^-1, i.e. the normalised columns of
        // R^T W -- no triangular solve.  R is kept packed in the wave's scratch (528 doubles behind the solver's 512).
        double* rp = scr + 512;
        {
          double dorig = 0.0;
#pragma unroll
          for (int r = 0; r < 32; ++r) dorig = (r == lane) ? g[r] : dorig;
          // (row i of the factor reaches the other lanes through v_readlane -- scalar operands of the update, no registers:
          // fetched from an LDS row buffer instead, hipcc keeps 31 loads in flight per step and spills 670 registers)
          auto rdl = [](const double v, const int src) {
            const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
            const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
            return __hiloint2double(hi, lo);
          };
#pragma unroll
          for (int i = 0; i < 32; ++i) {
            // a pivot below 1e-14 of its column's squared norm is rounding noise (cond(A) beyond 1e7): floor it, the
            // outer iteration goes on with a slightly wrong V for this pair instead of a NaN
            const double fl = 1e-14 * rdl(dorig, i);
            double sii = rdl(g[i], i);
            sii = sii > fl ? sii : fl;
            sii = sii > 1e-300 ? sii : 1e-300;
            const double ir = 1.0 / sqrt(sii);
            const double rij = (lane >= i && lane < 32) ? g[i] * ir : 0.0;
            const double f = (lane > i) ? rij : 0.0;
            g[i] = rij;
#pragma unroll
            for (int l = i + 1; l < 32; ++l) g[l] = fma(-rdl(rij, l), f, g[l]);
          }
          jacobi_dev::wave_lds_sync();
          const int tri = (lane * (lane + 1)) >> 1;
#pragma unroll
          for (int l = 0; l < 32; ++l)
            if (l <= lane && lane < 32) rp[tri + l] = g[l];
        }
        int inpairs = 0;
        const int insw = jacobi_dev::jacobi_split<32, 1, 8, false>(g, 32, 30, scr, &inpairs);
        if (insw > 1) notconv = 1;
        if (insw > 2) big = 1;
        // W -> R^T W, in place from the last row up (row i needs rows 0..i of W only)
#pragma unroll
        for (int i = 31; i >= 0; --i) {
          double acc = 0.0;
#pragma unroll
          for (int l = 0; l <= i; ++l) acc = fma(rp[((i * (i + 1)) >> 1) + l], g[l], acc);
          g[i] = acc;
          asm volatile("" : "+v"(g[i])::"memory");       // (one row's reads at a time: hoisted, the 528 of them spill)
        }
        // The inner solver swaps the two columns of a pair after every rotation, unconditionally: after T steps the
        // column order is a fixed permutation (odd-even transposition: reversal after 32 steps, identity after 64).
        // Undo it, so that V is the product of the ROTATIONS only -- close to the identity once the rotations are small.
        // (With the columns left scrambled, V is a rotation times a permutation and the cyclic block Jacobi loses its
        // quadratic convergence: 37 instead of ~8 outer sweeps at k = 144.)
        int origin = lane;
        {
          const int T = (2 * inpairs) & 63;
          for (int st = T - 1; st >= 0; --st) {
            if ((st & 1) == 0) origin ^= 1;                       // even step: (0,1)(2,3)...
            else if (origin >= 1 && origin <= 30) origin += (origin & 1) ? 1 : -1;   // odd step: (1,2)(3,4)...(29,30)
          }
        }
        double ss = 0.0;
#pragma unroll
        for (int r = 0; r < 32; ++r) ss = fma(g[r], g[r], ss);
        const double il = ss > 0.0 ? jacobi_dev::fast_rsqrt(ss) : 0.0;
#pragma unroll
        for (int r = 0; r < 32; ++r) g[r] *= il;         // lane j (< 32): column j of V
        // ---- V into B-operand layout: vop[s][nb] = V[4 s + q][16 nb + c16], 16 rows at a time through scr[16][32]
        double vop[8][2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          jacobi_dev::wave_lds_sync();
          if (lane < 32) {
#pragma unroll
            for (int r = 0; r < 16; ++r) scr[r * 32 + origin] = g[16 * half + r];
          }
          jacobi_dev::wave_lds_sync();
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            vop[4 * half + s4][0] = scr[(4 * s4 + q) * 32 + c16];
            vop[4 * half + s4][1] = scr[(4 * s4 + q) * 32 + 16 + c16];
          }
        }
        jacobi_dev::wave_lds_sync();
        // ---- 3. Y <- Y V, 16 rows at a time, in place (this wave owns these 32 columns for the round)
        const int rowa = c16;                            // A operand: row index inside the tile
        auto load_tile = [&](const int t, double (&a)[8]) {
          const int row = 16 * t + rowa;
#pragma unroll
          for (int s_ = 0; s_ < 8; ++s_) {
            const int col = (s_ < 4) ? 16 * I + 4 * s_ + q : 16 * J + 4 * (s_ - 4) + q;
            // unconditional, from a clamped address, and NOT zeroed: rows >= k are never stored, and a phantom column
            // (>= k) only meets the exact zeros of V's phantom rows (those columns are never rotated)
            a[s_] = G[(size_t)min(col, k - 1) * ldg + min(row, k - 1)];
          }
        };
        double a0[8], a1[8];
        load_tile(0, a0);
        for (int t = 0; t < ntile; ++t) {
          if (t + 1 < ntile) load_tile(t + 1, a1);
          v4d d0{0.0, 0.0, 0.0, 0.0}, d1{0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s_ = 0; s_ < 8; ++s_) {
            d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s_], vop[s_][0], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s_], vop[s_][1], d1, 0, 0, 0);
          }
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int row = 16 * t + q + 4 * reg;
            if (row < k) {
              if (okI) G[(size_t)colI * ldg + row] = d0[reg];
              if (okJ) G[(size_t)colJ * ldg + row] = d1[reg];
            }
          }
#pragma unroll
          for (int s_ = 0; s_ < 8; ++s_) a0[s_] = a1[s_];
        }
      }
      __syncthreads();
    }
    if (!__syncthreads_or(notconv)) {
      ++sweep;
      conv = 1;
      break;
    }
    // Stagnation: V = R^T W Lambda^-1 is orthogonal to eps cond(A) only, and what it leaves behind in the SMALL columns of Y
    // (an error of V times lambda_max / lambda_min) comes back as cosines above the stop rule sweep after sweep once
    // cond(A) is large -- the inner solves then keep reporting two cycles (one rotating, one verifying) without ever
    // reporting one.  kBlkFlat such sweeps in a row, or kBlkSweepCap sweeps in all, and the scalar one-sided iteration
    // below finishes the job: its rotations act on Y's columns directly (high relative accuracy whatever the norms).
    flat = __syncthreads_or(big) ? 0 : flat + 1;
    if (flat >= kBlkFlat || sweep + 1 >= kBlkSweepCap) {
      ++sweep;
      break;
    }
  }
  if (!conv && sweep < max_sweep) {
    __syncthreads();
    sweep += jacobi_stream(G, ldg, k, max_sweep - sweep, conv);
  }
  return sweep;
}

// ------------------------------------------------------------------ the per-point kernel
constexpr int kTile = 4;   // Gram register tile (kTile x kTile per thread)

// kMaxT = Gram tiles per thread per pass (1 covers k <= 64 with 256 threads)
template <bool BIG, int RMAX, int kMaxT>
__global__ void __launch_bounds__(BIG ? 768 : 256) letkf_point_kernel(const PointArgs A) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x;
  const int nthr = blockDim.x;
  const int k = A.k;
  const int nv = A.nv;
  const int nb = nv + 2;             // right-hand sides: r, r_det, x'_1..x'_nv
  const int ldg = A.ldg;
  const int ldy = A.ldy;
  const int tn = A.tn;
  const double km1 = (double)(k - 1);

  // ---- carve
  double* lds = smem;
  double* vec = lds;                 // 8 vectors of length k
  double* rvec = vec;                // Ys^T sqrt(w) dep
  double* rdvec = vec + k;           // same with depd
  double* lam = vec + 2 * k;
  double* wbar = vec + 3 * k;
  double* wbard = vec + 4 * k;
  double* sc1 = vec + 5 * k;         // (k-1)^.5 * lam^-.5   (T spectrum)
  double* sc2 = vec + 6 * k;         // 1/lam                (Pa spectrum)
  double* xsm = vec + 7 * k;         // 6*nv + 16 small per-variable scalars
  double* xmean = xsm;
  double* xdet = xsm + nv;
  double* cfac = xsm + 2 * nv;       // relaxation factor on T per variable
  double* cdiag = xsm + 3 * nv;      // relaxation diagonal term per variable (RTPP)
  double* sdot = xsm + 4 * nv;       // x_v . wbar
  double* sdotd = xsm + 5 * nv;      // x_v . wbar_det
  double* red = xsm + 6 * nv;        // 16 scalars
  double* wrow = red + 16;           // tn * 3: sqrt(w), sqrt(w)*dep, sqrt(w)*depd
  double* dyn = wrow + 3 * tn;
  if ((reinterpret_cast<uintptr_t>(dyn) & 15) != 0) dyn += 1;   // keep b128-able tiles 16-B aligned
  double* G;
  double* Yt;
  double* U;
  double* X;
  if constexpr (BIG) {
    double* ws = A.ws + (size_t)blockIdx.x * A.ws_per_block;
    G = ws;
    U = G + (size_t)k * ldg;
    X = U + (size_t)k * nb;
    Yt = dyn;
  } else {
    G = dyn;
    Yt = G + (size_t)k * ldg + (((size_t)k * ldg) & 1);
    U = Yt;                          // U and X alias the obs tile: it is dead after the Gram phase
    X = U + (size_t)k * nb;
  }

  const int ntile = (k + kTile - 1) / kTile;
  const int ntile2 = ntile * ntile;

  for (long it = blockIdx.x; it < A.npts; it += gridDim.x) {
    const long pt = xcd_remap(it, A.npts);

    // ---------------- per-point header
    long o0 = 0;
    int n = 0;
    double beta = 1.0;
    if (A.mode == 0) {
      o0 = A.obs_off[pt];
      n = (int)(A.obs_off[pt + 1] - o0);
      if (A.beta) beta = A.beta[pt];
    } else {
      n = A.nobsl[pt];
    }
    const double* g0 = A.gues ? A.gues + pt * A.sp : nullptr;
    double* a0 = A.anal ? A.anal + pt * A.sp : nullptr;

    if (A.mode == 0 && beta == 0.0) {            // letkf_tools.f90:333-359
      for (int e = tid; e < nv * k; e += nthr) {
        const int v = e / k, mm = e - v * k;
        if ((A.var_mask >> v) & 1u) a0[mm * A.sm + v * A.sv] = g0[k * A.sm + v * A.sv] + g0[mm * A.sm + v * A.sv];
      }
      if (A.det_run)
        for (int v = tid; v < nv; v += nthr)
          if ((A.var_mask >> v) & 1u) a0[(k + 1) * A.sm + v * A.sv] = g0[(k + 1) * A.sm + v * A.sv];
      if (A.rtps_out)
        for (int v = tid; v < nv; v += nthr)
          if ((A.var_mask >> v) & 1u) A.rtps_out[pt + A.infl_sv * (long)v] = 1.0;
      if (tid == 0) {
        if (A.status) A.status[pt] = 0;
        if (A.nsweep) A.nsweep[pt] = 0;
      }
      continue;
    }

    // variable skip mask for Q_UPDATE_TOP (letkf_tools.f90:371) and the solve's inflation slot
    bool qskip = false;
    if (A.mode == 0 && A.q_update_top > 0.0) qskip = g0[k * A.sm + A.iv_p * A.sv] < A.q_update_top;
    int v0 = 0;                                  // first variable of this class that is actually updated
    while (v0 < nv && (!((A.var_mask >> v0) & 1u) || (qskip && v0 >= A.iv_q_first && v0 <= A.iv_q_last))) ++v0;
    double* infl_p = nullptr;
    if (A.mode == 0) infl_p = (v0 < nv) ? &A.infl[pt + A.infl_sv * (long)v0] : nullptr;
    else infl_p = &A.infl[pt];
    const double infl_old = infl_p ? *infl_p : 1.0;

    __syncthreads();                             // previous point's LDS fully consumed

    // ---------------- phase 1+2: A = Ys^T Ys (+ shift), r = Ys^T sqrt(w) dep
    for (int j = tid; j < 2 * k; j += nthr) vec[j] = 0.0;        // rvec, rdvec
    if (tid < 16) red[tid] = 0.0;
    double p1 = 0.0, p3 = 0.0;                   // adaptive inflation sums (common_letkf.f90:233-249)
    int sweeps = 0, jconv = 1;

    if (n > 0) {
      for (int tile0 = 0; tile0 < ntile2; tile0 += nthr * kMaxT) {
        double acc[kMaxT][kTile * kTile];
#pragma unroll
        for (int t = 0; t < kMaxT; ++t)
#pragma unroll
          for (int e = 0; e < kTile * kTile; ++e) acc[t][e] = 0.0;
        double racc = 0.0, rdacc = 0.0;          // thread j < k: r_j ; thread k <= j < 2k: rd_j

        for (int i0 = 0; i0 < n; i0 += tn) {
          const int ni = min(tn, n - i0);
          __syncthreads();
          // weights of this tile
          if (tid < ni) {
            double w, d, dd = 0.0, rl;
            if (A.mode == 0) {
              const long e = o0 + i0 + tid;
              const int iob = A.obs_idx[e];
              rl = A.rloc_l[e];
              w = 1.0 / A.rdiag_l[e];
              d = A.dep[iob];
              if (A.det_run) dd = A.ensval[(long)iob * A.kld + k];
            } else {
              const long e = pt * (long)A.nobs + i0 + tid;
              rl = A.rloc[e];
              w = A.rdiag_wloc ? 1.0 / A.rdiag[e] : rl / A.rdiag[e];
              d = A.depv[e];
              if (A.depd) dd = A.depd[e];
            }
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
          // stage the tile: Yt[i][m] = sqrt(w_i) * y_i[m], zero padded to ldy
          if (A.mode == 0) {
            for (int e = tid; e < ni * ldy; e += nthr) {
              const int i = e / ldy, mm = e - i * ldy;
              double y = 0.0;
              if (mm < k) y = A.ensval[(long)A.obs_idx[o0 + i0 + i] * A.kld + mm] * wrow[i];
              Yt[e] = y;
            }
          } else {
            // dense column-major hdxb(nobs, ne): consecutive threads walk down a column (unit stride)
            const double* H = A.hdxb + (size_t)pt * (size_t)A.nobs * (size_t)k + i0;
            for (int e = tid; e < ni * ldy; e += nthr) {
              const int mm = e / ni, i = e - mm * ni;
              double y = 0.0;
              if (mm < k) y = H[(size_t)mm * A.nobs + i] * wrow[i];
              if (mm < ldy) Yt[i * ldy + mm] = y;
            }
          }
          __syncthreads();
          // Gram update, kTile x kTile register tiles
#pragma unroll
          for (int t = 0; t < kMaxT; ++t) {
            const int tl = tile0 + tid + t * nthr;
            if (tl < ntile2) {
              const int ti = tl / ntile, tj = tl - ti * ntile;
              const double* ya = Yt + ti * kTile;
              const double* yb = Yt + tj * kTile;
              for (int i = 0; i < ni; ++i) {
                double av[kTile], bv[kTile];
#pragma unroll
                for (int e = 0; e < kTile; ++e) {
                  av[e] = ya[i * ldy + e];
                  bv[e] = yb[i * ldy + e];
                }
#pragma unroll
                for (int ea = 0; ea < kTile; ++ea)
#pragma unroll
                  for (int eb = 0; eb < kTile; ++eb) acc[t][ea * kTile + eb] = fma(av[ea], bv[eb], acc[t][ea * kTile + eb]);
              }
            }
          }
          if (tile0 == 0) {
            for (int j = tid; j < 2 * k; j += nthr) {   // (k <= nthr/2 on the LDS path: one j per thread)
              const int col = (j < k) ? j : j - k;
              const double* wd = wrow + ((j < k) ? tn : 2 * tn);
              double sacc = 0.0;
              for (int i = 0; i < ni; ++i) sacc = fma(Yt[i * ldy + col], wd[i], sacc);
              if (2 * k <= nthr) {
                if (j < k) racc += sacc; else rdacc += sacc;
              } else {
                vec[j] += sacc;                  // large k: several j per thread, accumulate in place
              }
            }
          }
        }
        // write the finished tiles of this pass into G (column-major) with the inflation shift
        const double shift = km1 / infl_old;     // common_letkf.f90:140-143
        double trp = 0.0;                        // trace(Ys^T Ys) before the shift -> parm(2), :243-248
#pragma unroll
        for (int t = 0; t < kMaxT; ++t) {
          const int tl = tile0 + tid + t * nthr;
          if (tl < ntile2) {
            const int ti = tl / ntile, tj = tl - ti * ntile;
#pragma unroll
            for (int ea = 0; ea < kTile; ++ea)
#pragma unroll
              for (int eb = 0; eb < kTile; ++eb) {
                const int r = ti * kTile + ea, c = tj * kTile + eb;
                if (r < k && c < k) {
                  G[(size_t)c * ldg + r] = acc[t][ea * kTile + eb] + ((r == c) ? shift : 0.0);
                  if (r == c) trp += acc[t][ea * kTile + eb];
                }
              }
          }
        }
        if (A.infl_adaptive && trp != 0.0) atomicAdd(&red[2], trp);
        if (tile0 == 0 && 2 * k <= nthr) {
          if (tid < k) rvec[tid] = racc;
          else if (tid < 2 * k) rdvec[tid - k] = rdacc;
        }
      }
      // adaptive-inflation sums: p1 = sum dep^2 w, p3 = sum rloc (threads < tn hold partials)
      if (A.infl_adaptive) {
        p1 = group_sum<64>(p1);
        p3 = group_sum<64>(p3);
        if ((tid & 63) == 0) {
          atomicAdd(&red[0], p1);
          atomicAdd(&red[1], p3);
        }
      }
      __syncthreads();

      // ---------------- phase 3: eigen-decomposition (one-sided Jacobi on G = A)
      jconv = 0;
      if constexpr (RMAX > 0) sweeps = jacobi_cached<RMAX>(G, ldg, k, A.max_sweep, jconv);
      else if (A.big_block) sweeps = jacobi_block_mfma(G, ldg, k, A.max_sweep, dyn, jconv);
      else sweeps = jacobi_stream(G, ldg, k, A.max_sweep, jconv);

      // lambda_j = |g_j|, V = G / lambda
      {
        const int lane = tid & 63, grp = tid >> 6, ngrp = nthr >> 6;
        for (int j = grp; j < k; j += ngrp) {
          double* gj = G + (size_t)j * ldg;
          double ss = 0.0;
          for (int r = lane; r < k; r += 64) ss = fma(gj[r], gj[r], ss);
          ss = group_sum<64>(ss);
          const double l = sqrt(ss);
          const double il = 1.0 / l;
          for (int r = lane; r < k; r += 64) gj[r] *= il;
          if (lane == 0) lam[j] = l;
        }
      }
      __syncthreads();
    } else {
      // nobsl == 0 (common_letkf.f90:89-107): T = sqrt(rho) I, Pa = rho/(k-1) I, w-bar = 0
      for (int e = tid; e < k * ldg; e += nthr) {
        const int c = e / ldg, r = e - c * ldg;
        G[e] = (r == c) ? 1.0 : 0.0;
      }
      for (int j = tid; j < k; j += nthr) lam[j] = km1 / infl_old;
      __syncthreads();
    }

    // ---------------- status: spectrum checks (common_mtx.f90:66-78)
    int st = 0;
    {
      double lmx = 0.0, lmn = 1e300;
      for (int j = tid & 63; j < k; j += 64) {
        lmx = fmax(lmx, lam[j]);
        lmn = fmin(lmn, lam[j]);
      }
#pragma unroll
      for (int mk = 1; mk < 64; mk <<= 1) {
        lmx = fmax(lmx, shfl_xor_d(lmx, mk));
        lmn = fmin(lmn, shfl_xor_d(lmn, mk));
      }
      if (!jconv && A.max_sweep >= kMaxSweep) st = 1;   // (converging in the last permitted sweep is converged)
      else if (!(lmx > 0.0)) st = 2;
      else if (lmn < lmx * 1.4901161193847656e-08) st = 3;       // sqrt(DBL_EPSILON)
    }
    for (int j = tid; j < k; j += nthr) {
      const double l = lam[j];
      sc1[j] = sqrt(km1 / l);
      sc2[j] = 1.0 / l;
    }

    // ---------------- phase 4a: stage perturbations X[v][m], means, det member
    if (A.mode == 0) {
      for (int e = tid; e < nv * k; e += nthr) {
        const int v = e / k, mm = e - v * k;
        X[e] = g0[mm * A.sm + v * A.sv];
      }
      for (int v = tid; v < nv; v += nthr) {
        xmean[v] = g0[k * A.sm + v * A.sv];
        xdet[v] = A.det_run ? g0[(k + 1) * A.sm + v * A.sv] : 0.0;
      }
    }
    __syncthreads();

    // ---------------- phase 4b: U[j][b] = v_j . B_b,  B = [r, r_det, x'_1 .. x'_nv]
    for (int e = tid; e < k * nb; e += nthr) {
      const int j = e / nb, b = e - j * nb;
      const double* vj = G + (size_t)j * ldg;
      const double* bb = (b == 0) ? rvec : (b == 1) ? rdvec : X + (size_t)(b - 2) * k;
      double sacc = 0.0;
      for (int r = 0; r < k; ++r) sacc = fma(vj[r], bb[r], sacc);
      U[e] = sacc;
    }
    __syncthreads();

    // ---------------- phase 4c: w-bar = V (U_r / lam), w-bar_det, per-variable relaxation scalars
    for (int e = tid; e < 2 * k; e += nthr) {
      const int mm = (e < k) ? e : e - k, b = (e < k) ? 0 : 1;
      double sacc = 0.0;
      for (int j = 0; j < k; ++j) sacc = fma(G[(size_t)j * ldg + mm], U[j * nb + b] * sc2[j], sacc);
      (b == 0 ? wbar : wbard)[mm] = sacc;
    }
    if (A.mode == 0) {
      // RTPS needs var_g = |x'|^2 and var_a = x'^T Pa x' = sum_j U_jv^2 / lam_j (letkf_tools.f90:1982-1989)
      for (int v = tid; v < nv; v += nthr) {
        const double parm = A.relax_to_inflated_prior ? A.infl[pt + A.infl_sv * (long)v] : 1.0;   // :387-391
        double cf = 1.0, cd = 0.0;
        if (A.relax_alpha != 0.0) {              // RTPP :1960-1963
          cf = 1.0 - A.relax_alpha;
          cd = A.relax_alpha * sqrt(parm);
        } else if (A.relax_alpha_spread != 0.0) { // RTPS :1990-1999
          double var_g = 0.0, var_a = 0.0;
          for (int mm = 0; mm < k; ++mm) var_g = fma(X[v * k + mm], X[v * k + mm], var_g);
          for (int j = 0; j < k; ++j) var_a = fma(U[j * nb + 2 + v] * U[j * nb + 2 + v], sc2[j], var_a);
          if (var_g > 0.0 && var_a > 0.0)
            cf = A.relax_alpha_spread * sqrt(var_g * parm / (var_a * km1)) - A.relax_alpha_spread + 1.0;
        }
        cfac[v] = cf;
        cdiag[v] = cd;
        if (A.rtps_out && ((A.var_mask >> v) & 1u)) {   // work3da (letkf_tools.f90:460-462)
          const bool skipv = qskip && v >= A.iv_q_first && v <= A.iv_q_last;
          A.rtps_out[pt + A.infl_sv * (long)v] = (A.relax_alpha == 0.0 && A.relax_alpha_spread != 0.0 && !skipv) ? cf : 1.0;
        }
      }
    }
    __syncthreads();
    if (A.mode == 0) {
      for (int e = tid; e < 2 * nv; e += nthr) {
        const int v = (e < nv) ? e : e - nv;
        const double* wv = (e < nv) ? wbar : wbard;
        double sacc = 0.0;
        for (int mm = 0; mm < k; ++mm) sacc = fma(X[v * k + mm], wv[mm], sacc);
        (e < nv ? sdot : sdotd)[v] = sacc;
      }
    }
    __syncthreads();

    // ---------------- adaptive inflation (common_letkf.f90:233-254); uses the OLD rho above
    double infl_new = infl_old;
    if (A.infl_adaptive && n > 0) {
      const double parm1 = red[0], parm3 = red[1], parm2 = red[2] / km1;
      const double parm4 = (parm1 - parm3) / parm2 - infl_old;
      const double tq = (infl_old * parm2 + parm3) / parm2;
      const double sigma_o = 2.0 / parm3 * (tq * tq);
      const double gain = 0.04 * 0.04 / (sigma_o + 0.04 * 0.04);
      infl_new = infl_old + gain * parm4;
    }

    // ---------------- phase 5: analysis members  (letkf_tools.f90:472-513)
    if (A.mode == 0) {
      for (int e = tid; e < nv * k; e += nthr) {
        const int v = e / k, mm = e - v * k;
        const bool skip = qskip && v >= A.iv_q_first && v <= A.iv_q_last;
        double out;
        if (skip) {
          out = xmean[v] + X[e];
        } else {
          double tx = 0.0;                       // (T x'_v)[mm] = sum_j V[mm][j] sqrt((k-1)/lam_j) U[j][v]
          for (int j = 0; j < k; ++j) tx = fma(G[(size_t)j * ldg + mm], sc1[j] * U[j * nb + 2 + v], tx);
          const double pert = cfac[v] * tx + cdiag[v] * X[e];
          out = xmean[v] + beta * (pert + sdot[v]) + (1.0 - beta) * X[e];
        }
        if (A.q_sprd_max > 0.0 && v == A.iv_q_first && !skip) X[e] = out;   // keep for the clamp
        else if ((A.var_mask >> v) & 1u) a0[mm * A.sm + v * A.sv] = out;
      }
      if (A.det_run) {
        for (int v = tid; v < nv; v += nthr) {
          const bool skip = qskip && v >= A.iv_q_first && v <= A.iv_q_last;
          if ((A.var_mask >> v) & 1u)
            a0[(k + 1) * A.sm + v * A.sv] = skip ? xdet[v] : xdet[v] + sdotd[v] * beta;   // :489-497
        }
      }
      if (A.q_sprd_max > 0.0 && !(qskip) && ((A.var_mask >> A.iv_q_first) & 1u)) {   // :500-513, variable iv3d_q only
        __syncthreads();
        const int v = A.iv_q_first;
        if (tid < 64) {
          double sm_ = 0.0;
          for (int mm = tid; mm < k; mm += 64) sm_ += X[v * k + mm];
          sm_ = group_sum<64>(sm_);
          const double q_mean = sm_ / (double)k;
          double ss = 0.0;
          for (int mm = tid; mm < k; mm += 64) {
            const double d = X[v * k + mm] - q_mean;
            ss = fma(d, d, ss);
          }
          ss = group_sum<64>(ss);
          const double q_sprd = sqrt(ss / km1) / q_mean;
          for (int mm = tid; mm < k; mm += 64) {
            double val = X[v * k + mm];
            if (q_sprd > A.q_sprd_max) val = q_mean + (val - q_mean) * A.q_sprd_max / q_sprd;
            a0[mm * A.sm + v * A.sv] = val;
          }
        }
      }
      if (A.infl_adaptive) {                     // :396-398: every updated variable of the class gets its first slot's value
        for (int v = tid; v < nv; v += nthr) {
          const bool skip = qskip && v >= A.iv_q_first && v <= A.iv_q_last;
          if (!skip && ((A.var_mask >> v) & 1u)) A.infl[pt + A.infl_sv * (long)v] = infl_new;
        }
      }
    } else if (A.infl_adaptive && n > 0 && tid == 0) {
      A.infl[pt] = infl_new;
    }

    // ---------------- optional k x k outputs (fine boundary, parity, diagnostics)
    if (A.trans_out || A.pa_out) {
      double* To = A.trans_out ? A.trans_out + (size_t)pt * k * k : nullptr;
      double* Po = A.pa_out ? A.pa_out + (size_t)pt * k * k : nullptr;
      for (int e = tid; e < k * k; e += nthr) {
        const int c = e / k, r = e - c * k;      // column-major, coalesced over r
        double t = 0.0, pp = 0.0;
        for (int j = 0; j < k; ++j) {
          const double vv = G[(size_t)j * ldg + r] * G[(size_t)j * ldg + c];
          t = fma(vv, sc1[j], t);
          pp = fma(vv, sc2[j], pp);
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
      if (A.nsweep) A.nsweep[pt] = sweeps;
    }
  }
}

// ------------------------------------------------------------------ staged path: eigen stage for orders beyond the
// workgroup-resident Jacobi of letkf_eig.hip (letkf_staged.hip, solver 2): the block Jacobi on the slab's matrix
__global__ void __launch_bounds__(768) letkf_eig_block_kernel(const EigArgs E) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  for (long it = blockIdx.x; it < E.npts; it += gridDim.x) {
    const int m = E.meta[2 * it + 1];
    if ((E.meta[2 * it] >> 8) != 2 || m < 2) continue;          // (uniform for the workgroup)
    double* G = E.ws + (size_t)it * E.ws_per_point;
    int conv = 0;
    __syncthreads();
    const int sweeps = jacobi_block_mfma(G, staged_dev::staged_ld(m), m, E.max_sweep, smem, conv);
    if (threadIdx.x == 0) {
      E.info[2 * it] = sweeps;
      E.info[2 * it + 1] = conv;
    }
  }
}

hipError_t launch_eig_block(const EigArgs& e, int kmax, int num_cu, hipStream_t st) {
  const int nblk = (kmax + 15) / 16, nbe = nblk + (nblk & 1);
  int waves = nbe / 2;
  if (waves < 4) waves = 4;
  if (waves > 12) waves = 12;
  const size_t lds = (size_t)waves * kBlkScr * sizeof(double);
  if (lds > 48 * 1024) {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(&letkf_eig_block_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return err;
  }
  const long g = e.npts < 2L * num_cu ? (e.npts > 0 ? e.npts : 1) : 2L * num_cu;
  hipLaunchKernelGGL(letkf_eig_block_kernel, dim3((unsigned)g), dim3(64 * waves), lds, st, e);
  return hipGetLastError();
}

// ------------------------------------------------------------------ streaming passes either side of the loop
// scale/letkf/letkf_tools.f90:209-230: members 0..k-1 -= mean (slot k)
__global__ void ens_to_pert_kernel(int k, int nv, long npts, double* x, long sp, long sm, long sv) {
  const long total = npts * nv;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long v = e / npts, pt = e - v * npts;
    double* b = x + pt * sp + v * sv;
    const double mean = b[k * sm];
    for (int m = 0; m < k; ++m) b[m * sm] -= mean;
  }
}

// scale/common/common_scale.f90:1513-1552: slot k = (sum_{m<k} x_m) / k, summed in member order
__global__ void ens_mean_kernel(int k, int nv, long npts, double* x, long sp, long sm, long sv) {
  const long total = npts * nv;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long v = e / npts, pt = e - v * npts;
    double* b = x + pt * sp + v * sv;
    double s = b[0];
    for (int m = 1; m < k; ++m) s += b[m * sm];
    b[k * sm] = s / (double)k;
  }
}


// ------------------------------------------------------------------ row f3: state_trans, member<->points, spread
// scale/common/common_scale.f90:1181-1224 (inverse = false) and :1229-1280 (inverse = true); one thread per (k,i,j),
// level-fastest layout => fully coalesced, 11 reads + 5 writes per point.
__global__ void state_trans_kernel(const letkf_state_consts C, int nlev, long nxy, int nv3d, double* v, int inverse) {
  const long total = (long)nlev * nxy;
  const long stride = total;                       // distance between variables
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    double* p = v + e;
    if (inverse) {                                 // :1243-1250 positive-definite clamps
      for (int n = C.iv_q; n < nv3d; ++n) {
        const bool clamp = (n == C.iv_q) ? C.positive_definite_q != 0 : C.positive_definite_qhyd != 0;
        if (clamp) p[n * stride] = fmax(p[n * stride], 0.0);
      }
    }
    double qdry = 1.0, cvtot = 0.0;
    for (int n = C.iv_q; n < nv3d; ++n) {          // :1199-1202
      const double q = p[n * stride];
      qdry = qdry - q;
      cvtot = cvtot + q * C.tracer_cv[n - C.iv_q];
    }
    cvtot = C.cvdry * qdry + cvtot;
    const double rtot = C.rdry * qdry + C.rvap * p[C.iv_q * stride];
    if (!inverse) {
      const double cpovcv = (cvtot + rtot) / cvtot;
      const double rho = p[C.iv_rho * stride];
      const double pres = C.pre00 * pow(p[C.iv_rhot * stride] * rtot / C.pre00, cpovcv);
      const double temp = pres / (rho * rtot);
      const double u = p[C.iv_rhou * stride] / rho, vv = p[C.iv_rhov * stride] / rho, w = p[C.iv_rhow * stride] / rho;
      p[C.iv_u * stride] = u;
      p[C.iv_v * stride] = vv;
      p[C.iv_w * stride] = w;
      p[C.iv_t * stride] = temp;
      p[C.iv_p * stride] = pres;
    } else {
      const double cvovcp = cvtot / (cvtot + rtot);
      const double pr = p[C.iv_p * stride];
      const double rho = pr / (rtot * p[C.iv_t * stride]);
      const double rhot = C.pre00 / rtot * pow(pr / C.pre00, cvovcp);
      const double ru = p[C.iv_u * stride] * rho, rv = p[C.iv_v * stride] * rho, rw = p[C.iv_w * stride] * rho;
      p[C.iv_rhot * stride] = rhot;
      p[C.iv_rhow * stride] = rw;
      p[C.iv_rhov * stride] = rv;
      p[C.iv_rhou * stride] = ru;
      p[C.iv_rho * stride] = rho;
    }
  }
}

// grd_to_buf / buf_to_grd (common_mpi_scale.f90:1428-1480) fused with the level/member re-ordering of
// read_ens_mpi / write_ens_mpi: a 32 x 32 (level x point) tile is transposed through LDS so that both the
// level-fastest field and the point-fastest ensemble array are touched with unit stride.
__global__ void member_points_kernel(int dir, int nlev, int nlon, long nxy, int np, int rank, long nij1, double* v3dg,
                                     double* x, long sp, long sm_m, long sv) {
  __shared__ double tile[32][33];
  const int n = blockIdx.z;                         // variable
  const long i0 = (long)blockIdx.x * 32;            // local point tile
  const int k0 = blockIdx.y * 32;                   // level tile
  const int tx = threadIdx.x, ty = threadIdx.y;     // 32 x 8
  double* fld = v3dg + (long)n * nlev * nxy;        // v3dg(k, ilon, ilat, n): level-fastest, then j = ilon-1 + nlon*(ilat-1)
  double* xs = x + sm_m + (long)n * sv;             // slot m, variable n
  if (dir == 0) {
    for (int r = ty; r < 32; r += 8) {              // read: threads along k
      const long i = i0 + r;
      const int k = k0 + tx;
      if (i < nij1 && k < nlev) tile[r][tx] = fld[k + (long)nlev * (rank + (long)np * i)];
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {              // write: threads along the point index
      const long i = i0 + tx;
      const int k = k0 + r;
      if (i < nij1 && k < nlev) xs[(i + nij1 * k) * sp] = tile[tx][r];
    }
  } else {
    for (int r = ty; r < 32; r += 8) {
      const long i = i0 + tx;
      const int k = k0 + r;
      if (i < nij1 && k < nlev) tile[tx][r] = xs[(i + nij1 * k) * sp];
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
      const long i = i0 + r;
      const int k = k0 + tx;
      if (i < nij1 && k < nlev) fld[k + (long)nlev * (rank + (long)np * i)] = tile[r][tx];
    }
  }
  (void)nlon;
}

// scale/common/common_scale.f90:1570-1607, member-order summation
__global__ void ens_spread_kernel(int k, int nv, long npts, const double* x, long sp, long sm, long sv, double* sprd) {
  const long total = npts * nv;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long v = e / npts, pt = e - v * npts;
    const double* b = x + pt * sp + v * sv;
    const double mean = b[k * sm];
    double d = b[0] - mean;
    double s = d * d;
    for (int m = 1; m < k; ++m) {
      d = b[m * sm] - mean;
      s = s + d * d;
    }
    sprd[e] = sqrt(s / (double)(k - 1));
  }
}

// ------------------------------------------------------------------ host-callable launchers
template <bool BIG, int RMAX, int MAXT>
static hipError_t launch_one(const PointArgs& a, int grid, int block, size_t lds, hipStream_t st) {
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&letkf_point_kernel<BIG, RMAX, MAXT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((letkf_point_kernel<BIG, RMAX, MAXT>), dim3(grid), dim3(block), lds, st, a);
  return hipGetLastError();
}

hipError_t launch_point_kernel(const PointArgs& a, const LaunchPlan& p, hipStream_t st) {
  if (p.big) return launch_one<true, 0, 1>(a, p.grid, p.block, p.lds_bytes, st);
  switch (p.rmax) {
    case 4: return launch_one<false, 4, 1>(a, p.grid, p.block, p.lds_bytes, st);
    case 7: return launch_one<false, 7, 1>(a, p.grid, p.block, p.lds_bytes, st);
    case 8: return launch_one<false, 8, 1>(a, p.grid, p.block, p.lds_bytes, st);
    case 13: return launch_one<false, 13, 4>(a, p.grid, p.block, p.lds_bytes, st);
    default: return launch_one<false, 16, 4>(a, p.grid, p.block, p.lds_bytes, st);
  }
}

hipError_t launch_ens_to_pert(int k, int nv, long npts, double* x, long sp, long sm, long sv, hipStream_t st) {
  const long total = npts * nv;
  const int block = 256;
  const int grid = (int)((total + block - 1) / block < 65535L * 16 ? (total + block - 1) / block : 65535L * 16);
  hipLaunchKernelGGL(ens_to_pert_kernel, dim3(grid > 0 ? grid : 1), dim3(block), 0, st, k, nv, npts, x, sp, sm, sv);
  return hipGetLastError();
}

hipError_t launch_ens_mean(int k, int nv, long npts, double* x, long sp, long sm, long sv, hipStream_t st) {
  const long total = npts * nv;
  const int block = 256;
  const int grid = (int)((total + block - 1) / block < 65535L * 16 ? (total + block - 1) / block : 65535L * 16);
  hipLaunchKernelGGL(ens_mean_kernel, dim3(grid > 0 ? grid : 1), dim3(block), 0, st, k, nv, npts, x, sp, sm, sv);
  return hipGetLastError();
}

hipError_t launch_state_trans(const letkf_state_consts& c, int nlev, long nxy, int nv3d, double* v, int inverse,
                              hipStream_t st) {
  const long total = (long)nlev * nxy;
  const int block = 256;
  long g = (total + block - 1) / block;
  if (g > 65535L * 8) g = 65535L * 8;
  hipLaunchKernelGGL(state_trans_kernel, dim3((unsigned)(g > 0 ? g : 1)), dim3(block), 0, st, c, nlev, nxy, nv3d, v, inverse);
  return hipGetLastError();
}

hipError_t launch_member_points(int dir, int nlev, int nlon, long nxy, int nv3d, int np, int rank, long nij1,
                                double* v3dg, double* x, long sp, long sm_m, long sv, hipStream_t st) {
  dim3 grid((unsigned)((nij1 + 31) / 32), (unsigned)((nlev + 31) / 32), (unsigned)nv3d);
  hipLaunchKernelGGL(member_points_kernel, grid, dim3(32, 8), 0, st, dir, nlev, nlon, nxy, np, rank, nij1, v3dg, x, sp,
                     sm_m, sv);
  return hipGetLastError();
}

hipError_t launch_ens_spread(int k, int nv, long npts, const double* x, long sp, long sm, long sv, double* sprd,
                             hipStream_t st) {
  const long total = npts * nv;
  const int block = 256;
  long g = (total + block - 1) / block;
  if (g > 65535L * 16) g = 65535L * 16;
  hipLaunchKernelGGL(ens_spread_kernel, dim3((unsigned)(g > 0 ? g : 1)), dim3(block), 0, st, k, nv, npts, x, sp, sm, sv, sprd);
  return hipGetLastError();
}

}  // namespace letkf
