// letkf_jacobi_dev.h -- the in-register one-sided Jacobi eigensolver (row-split layout) and its small helpers, shared by
// the wave kernel (letkf_wave.hip: the k x k problem of a grid point) and the large-k workgroup kernel
// (letkf_kernels.hip: the 32 x 32 problems of its block Jacobi).  Device code only; everything is inlined.
#pragma once
#include <hip/hip_runtime.h>

namespace letkf {
namespace jacobi_dev {

typedef double v4d __attribute__((ext_vector_type(4)));

// LDS written by some lanes of this wave, read by others: DS instructions of one wave execute in
// order, so only the compiler has to be kept from reordering across the hand-off.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A grid point is solved by NW wavefronts (NW = 1: k <= 62, 4 independent points per 256-thread workgroup;
// NW = 2: 62 < k <= 100, one point per 128-thread workgroup).  Hand-offs through LDS between lanes of the point:
template <int NW>
__device__ __forceinline__ void psync() {
  if constexpr (NW == 1) wave_lds_sync();
  else __syncthreads();
}
template <int NW>
__device__ __forceinline__ bool pany(bool v) {
  if constexpr (NW == 1) return __builtin_amdgcn_ballot_w64(v) != 0ull;
  else return __syncthreads_or(v) != 0;
}

constexpr double kRotTol2W = 1e-30;   // rotate when cos^2 > 1e-30
#ifndef LETKF_EARLY_TOL2
#define LETKF_EARLY_TOL2 1e-16
#endif
#ifndef LETKF_EARLY_T2
#define LETKF_EARLY_T2 1e-12
#endif
#ifndef LETKF_STOP_TOL2
#define LETKF_STOP_TOL2 1e-24
#endif
// The iteration counts as converged when every pair visited in a full cycle had |cos| <= 1e-12 (all of them were still
// rotated away in that cycle, so what is left is second order).  Round 3: 1e-12 instead of 1e-10.  The interior points of
// a workload never saw the difference -- they end on the early-stop rule below, or overshoot by the quadratic convergence
// (C2: 4.856 sweeps either way, same time) -- but a point with fewer local observations than members has the eigenvalue
// (k-1)/rho with multiplicity k - n, pairs inside that cluster are rotated by 45 degrees at rounding-level cosines (no
// quadratic phase, no early stop), and the couplings of up to 1e-10 that the old rule left between the cluster and the
// other columns showed up as 5e-12 .. 1.5e-11 in the analysis members (tools/r3_sparse_margin.py: only points with
// 1 <= n < k, 1000 x the interior's error, 8 x inside the tolerance).  With 1e-12: 1.5e-14 on the same points for
// +0.4 sweeps there (C2-mini-sparse 6.85 -> 7.24 sweeps, +4 % time); pinned by tests/test_gpu_sparse_margin.py.
constexpr double kStopTol2W = LETKF_STOP_TOL2;
// Early stop: a full cycle in which every pair had |cos| <= 1e-8 AND every applied rotation a tangent |t| <= 1e-6 ends the
// iteration as well -- each coupling was annihilated once in that cycle and re-filled by at most k products t * cos
// <= 1e-14.  The tangent condition matters: with (near-)multiple eigenvalues a pair of almost equal columns is rotated
// by a large angle at a tiny cosine, which shuffles the couplings of size 1e-8 of those two columns to all others
// after they were visited -- without it the result is only first-order accurate (measured: 8e-10 on T for n < k,
// tools/parity_margin.py).  Such points simply fall back to the rule above (1e-12).
constexpr double kEarlyTol2W = LETKF_EARLY_TOL2;
constexpr double kEarlyT2W = LETKF_EARLY_T2;
constexpr double kEarlyTW = 1e-6;      // sqrt(kEarlyT2W): the tangent is compared as |t| (no multiply)
static_assert(LETKF_EARLY_T2 == 1e-12, "kEarlyTW is its square root");

// 1/sqrt(x) and 1/x from the hardware seeds (v_rsq_f64 / v_rcp_f64, ~2^-23) + two Newton steps each: full
// double precision without the IEEE division / sqrt expansions (~25 instructions each), which were 1/3 of
// the Jacobi step's FP64 issue slots.
__device__ __forceinline__ double fast_rsqrt(double x) {
  // (round 3: ONE third-order step instead of two Newton steps -- x y0^2 = 1 - e, so 1/sqrt(x) = y0 (1 - e)^-1/2 =
  // y0 (1 + e/2 + 3 e^2/8 + 5 e^3/16 ...); with the seed's e ~ 2.4e-7 the first dropped term is 4e-21.  5 instructions, was 8.)
  const double y = __builtin_amdgcn_rsq(x);
  const double e = fma(-x * y, y, 1.0);
  const double q = e * fma(0.375, e, 0.5);
  return fma(y, q, y);
}
// The rotation ANGLE takes the hardware seeds as they are (~2^-23 relative): a tangent that is off by delta still gives
// an exactly orthogonal (scaled) rotation -- the cosine and the scales below are computed from the tangent actually
// applied -- and leaves delta * cos of the coupling behind instead of 0: 1e-15 in the last cycle, where every cosine
// is <= 1e-8, and in the cycles before it far less than the couplings the other rotations re-fill.  (One Newton step
// each, as before round 2, was 12 FP64 instructions per step pair: C2 400.8 -> 394.7 ms, same sweep count to five
// digits, same parity.)
__device__ __forceinline__ double fast_rsqrt1(double x) { return __builtin_amdgcn_rsq(x); }
__device__ __forceinline__ double fast_rcp1(double x) { return __builtin_amdgcn_rcp(x); }
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  return fma(r, e, r);
}

// ---------------------------------------------------------------------------------------------
// One-sided (Hestenes) Jacobi on register-resident columns.
//
// Pair ordering = odd-even transposition on a line of column positions: even steps pair positions (0,1)(2,3)...,
// odd steps (1,2)(3,4)...; after its rotation a pair SWAPS places.  After k steps every one of the k(k-1)/2 column
// pairs has met exactly once, for any k, and the only partners a column ever has are its two neighbours.
// The rotation itself is a "fast" scaled rotation G' = H + coef * G (one FMA per element): the cosine is not
// multiplied into the column but accumulated in a per-column inverse scale `is`, and the squared norms alpha follow
// the rotation identities alpha' = alpha -/+ t*gamma.  Both are refreshed once per sweep.
// (Earlier layouts, measured and replaced -- see DESIGN.md 4.1: a workgroup per point with G in LDS; lane = column
// with a ds_bpermute XOR tournament; lane = column with DPP even steps and LDS-chunk odd steps.)
//
// Row-split layout: a SLOT of two lanes (m, m + 32) owns the two columns at line positions (2m, 2m + 1),
// lane m their even rows, lane m + 32 their odd rows:
//   even steps pair the two columns of a slot: no column moves at all, KR/2 FMAs for the inner product (the two
//     halves meet through v_permlane32_swap) and KR FMAs for the two half-columns;
//   odd steps pair column 2m+1 with 2m+2 of the next slot: each lane fetches one half-column from either
//     neighbour slot (wave_shr:1 / wave_shl:1 DPP moves, KR dword moves each way as before), but computes only ONE
//     inner product half (the left slot of a pair passes its rotation on through four more DPP moves).
// Per two steps: 3/2 KR FMAs + 2 KR moves less than the column-per-lane version (~390 instead of ~500
// instructions at k = 50).  The lanes of unused slots are switched off for the whole iteration: DPP reads from a
// disabled lane return 0 (bound_ctrl), which is exactly the "no partner" case at both ends of the line.
// ---------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_shift0(double v) {     // wave_shl:1 (0x130) / wave_shr:1 (0x138), 0 if no source
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// v[l & 31] + v[32 + (l & 31)] in every lane (both lanes of a slot must be active)
__device__ __forceinline__ double slot_sum(double v) {
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}

// sum of the two lanes of a slot.  One wave: lanes (m, m + 32) through v_permlane32_swap.  Two waves (k > 62): the
// same lane of the two waves, through LDS and a workgroup barrier; the two exchange buffers alternate, so that a buffer
// is rewritten only after the barrier that follows everybody's read of it.
template <int NW>
__device__ __forceinline__ double slot_sum_nw(double v, double* xs, int& ph) {
  if constexpr (NW == 1) {
    return slot_sum(v);
  } else {
    const int t = threadIdx.x & 127;
    xs[ph * 128 + t] = v;
    __syncthreads();
    const double o = xs[ph * 128 + (t ^ 64)];
    ph ^= 1;
    return v + o;
  }
}

// NW = 2 (62 < k <= 100): slot m = lane m of both waves (wave 0 the even rows, wave 1 the odd rows); everything
// else as for one wave -- the DPP shifts stay inside a wave, only the inner-product halves and the convergence vote
// cross waves (three workgroup barriers per step pair).
// EARLY = false: only the 1e-10 rule (the block Jacobi of letkf_kernels.hip reads "one cycle" as "this block pair
// was already diagonal" and must not have rotations of 1e-8 pass for that).
// INPLACE = true: two half-column arrays instead of three.  The spare array of the copy-free form below costs KR
// registers; at KR = 100 on two waves that makes 300 for the columns alone, a third of them in AGPRs, and 548 of the
// 1283 instructions of a step pair were v_accvgpr moves.  In place the even step needs one move per row (the old lower
// element is an operand of both new ones) and the odd step fetches the right neighbour's column twice (inner product,
// then update) instead of keeping it: +2 instructions per row, no AGPR traffic.
// stop_tol2: the squared cosine of the stop rule (kStopTol2W; the block Jacobi of letkf_kernels.hip, whose 32 x 32 problems are
// Gram matrices -- squared condition, a noise floor of eps cond^2 on their cosines -- keeps round 2's 1e-20).
template <int KR, int NW, int RC = 24, bool EARLY = true, bool INPLACE = false>
__device__ __forceinline__ int jacobi_split(double (&g)[KR], const int k, const int max_sweep, double* lds,
                                            int* pairs_out = nullptr, int* conv_out = nullptr, const double stop_tol2 = kStopTol2W) {
  static_assert(KR % 2 == 0, "row halves");
  constexpr int H = KR / 2;
  static_assert(RC % 2 == 0, "RC rows per conversion chunk: RC * 64 NW doubles of LDS");
  constexpr int NL = 64 * NW;
  int lane = threadIdx.x & (NL - 1);
  if constexpr (NW == 2) asm volatile("" : "+v"(lane));   // (two-wave points, see letkf_wave_kernel: nothing built from the lane number may be hoisted out of the caller's loop)
  const int slot = (NW == 1) ? (lane & 31) : (lane & 63), par = (NW == 1) ? (lane >> 5) : (lane >> 6);
  int ph = 0;
  const int ncol = (k + 1) & ~1;               // an odd k gets one zero column as an extra (inert) participant
  const int S = ncol >> 1;                     // slots in use
  double xa[H], xb[H], xf[INPLACE ? 1 : H];
  // ---- column-per-lane -> row-split: lds[r][col], r in chunks
#pragma unroll
  for (int r0 = 0; r0 < KR; r0 += RC) {
    psync<NW>();
#pragma unroll
    for (int r = r0; r < r0 + RC && r < KR; ++r) lds[(r - r0) * NL + lane] = g[r];
    psync<NW>();
#pragma unroll
    for (int rr = r0 / 2; rr < (r0 + RC) / 2 && rr < H; ++rr) {
      const double2 v2 = *reinterpret_cast<const double2*>(&lds[(2 * rr + par - r0) * NL + 2 * slot]);
      xa[rr] = v2.x;
      xb[rr] = v2.y;
    }
  }
  int sweep = 0, pairs_done = 0, conv = 0;
  // Lanes of unused slots (slot >= S) hold zero columns.  NW = 1: they sit out the whole iteration (nothing below
  // synchronises more than the wave).  NW = 2: the iteration holds WORKGROUP barriers (slot_sum_nw, the vote), and a
  // barrier must not sit in divergent code -- so every lane runs the loop, the scalar arithmetic of an unused slot works on
  // zeros (no rotation, no vote), and only the barrier-free ROW loops are switched off for it (`rows`): there a DPP read
  // from a disabled lane returns 0 (bound_ctrl), which is exactly the "no partner" case at the end of the line, and the
  // unused slot's own registers are never written.  (Rounds 1 and 2 had the barriers inside `if (slot < S)`, relying on
  // s_barrier counting waves; same instructions in the row loops, same results.)
  const bool act = slot < S;
  const bool rows = NW == 1 || act;
  if (NW == 2 || act) {
    const bool hasL = act && slot > 0, hasR = act && slot + 1 < S;
    // (lane masks of the votes are combined with scalar instructions: written as `hasR && (a || b)` hipcc builds short-circuit
    // branches inside the step pair)
    const unsigned long long hasRm = __builtin_amdgcn_ballot_w64(hasR);
    double alA = 0.0, alB = 0.0, isA = 1.0, isB = 1.0, scA = 1.0, scB = 1.0;
    // `quiet` counts consecutive step pairs in which no visited column pair exceeded the tolerance; S of them in a
    // row are one full cycle of the ordering (every column pair seen once) whatever step it started at, so the
    // iteration stops S step pairs after the last significant rotation, not at the next sweep boundary.
    int quiet = 0, quiet2 = 0, pairs = 0, vph = 0;
    bool done = false;
    for (; sweep < max_sweep && !done; ++sweep) {
      // refresh: fold the scales back, recompute the squared norms
      double a0 = 0.0, b0 = 0.0;
      if (rows) {
#pragma unroll
        for (int rr = 0; rr < H; ++rr) {
          xa[rr] *= isA;
          xb[rr] *= isB;
          a0 = fma(xa[rr], xa[rr], a0);
          b0 = fma(xb[rr], xb[rr], b0);
        }
      }
      alA = slot_sum_nw<NW>(a0, lds, ph);
      alB = slot_sum_nw<NW>(b0, lds, ph);
      isA = isB = scA = scB = 1.0;
      for (int t = 0; t < ncol && !done; t += 2) {
        // the two votes of the step pair, kept as lane masks (scalar registers): accumulated as per-lane booleans hipcc holds
        // them in a vector register and converts back and forth (5 vector instructions per step pair)
        unsigned long long notconv = 0, notconv2 = 0;
        // ---------------- even step: the slot's own two columns (A at the lower position)
        {
          double p0 = 0.0, p1 = 0.0;
          if (rows) {
#pragma unroll
            for (int rr = 0; rr < H; ++rr) {
              if (rr & 1) p1 = fma(xa[rr], xb[rr], p1);
              else p0 = fma(xa[rr], xb[rr], p0);
            }
          }
          const double ga = slot_sum_nw<NW>(p0 + p1, lds, ph) * (isA * isB);
          const double a = alA, b = alB;
          const double g2 = ga * ga, ab = a * b;
          notconv |= __builtin_amdgcn_ballot_w64(g2 > stop_tol2 * ab);
          // tan(theta) = gamma sign(h) / (|h| + sqrt(h^2 + gamma^2)), h = (beta - alpha) / 2 (the smaller root of
          // t^2 + 2 (h / gamma) t - 1 = 0).  Pairs with |cos| <= 1e-15 are NOT rotated: inside a cluster of equal eigenvalues
          // (every point with fewer observations than members) h is rounding noise as well, the angle would be 45 degrees at a
          // cosine that is noise, and such a rotation shuffles the couplings of its two columns to all others after those
          // were visited (tried without the test: tests/test_gpu_sparse_margin.py 1.5e-14 -> 2e-12).  It also keeps the
          // 0 * rsq(0) = NaN of two zero columns out.
          const bool rot = g2 > kRotTol2W * ab;
          const double h = 0.5 * (b - a);
          const double x = fma(h, h, g2);
          const double den = fma(x, fast_rsqrt1(x), fabs(h));
          double tt = ga * fast_rcp1(copysign(den, h));
          tt = rot ? tt : 0.0;
          if constexpr (EARLY)
            notconv2 |= __builtin_amdgcn_ballot_w64(g2 > kEarlyTol2W * ab) | __builtin_amdgcn_ballot_w64(fabs(tt) > kEarlyTW);
          const double w = fma(tt, tt, 1.0);
          const double c = fast_rsqrt(w);
          const double tg = tt * ga, wc = w * c;
          // rotate and swap: position 2m takes c (g_B + t g_A), position 2m+1 takes c (g_A - t g_B)
          const double cA = tt * (isA * scB), cB = -tt * (isB * scA);
          const double nisA = isB * c, nscA = scB * wc, nisB = isA * c, nscB = scA * wc;
          alA = b + tg;
          alB = a - tg;
          isA = nisA;
          scA = nscA;
          isB = nisB;
          scB = nscB;
          // the new A goes to the spare array xf, the new B is accumulated in place in xa[] (every FMA adds into the
          // register its result stays in: no copies)
          if (!rows) {
            // (an unused slot of a two-wave point: its columns stay the zeros they are)
          } else if constexpr (INPLACE) {
            // A stays in xa[], B in xb[]: xa <- xb + cA xa, xb <- xa_old + cB xb, as one instruction sequence per row
            // (from C++ hipcc renames the whole column per step)
#pragma unroll
            for (int rr = 0; rr < H; ++rr) {
              double t;
              asm("v_mov_b64 %2, %0\n\t"
                  "v_fma_f64 %0, %3, %0, %1\n\t"
                  "v_fma_f64 %1, %4, %1, %2"
                  : "+v"(xa[rr]), "+v"(xb[rr]), "=&v"(t)
                  : "v"(cA), "v"(cB));
            }
          } else {
#pragma unroll
          for (int rr = 0; rr < H; ++rr) {
            // three-address form with an early-clobber destination: left to itself the compiler accumulates into
            // xb's registers (v_fmac) and then has to copy the column out of the way of the DPP fetch below.  (No
            // "nothing to rotate" shortcut here: its copy path made the allocator shuffle 50 registers per step pair.)
            asm("v_fma_f64 %0, %1, %2, %3" : "=&v"(xf[rr]) : "v"(cA), "v"(xa[rr]), "v"(xb[rr]));
            xa[rr] = fma(cB, xb[rr], xa[rr]);
          }
          }
        }
        // ---------------- odd step: own B (now in xa[]) with the right slot's A; own A (in xf[]) with the left slot's B
        {
          double p0 = 0.0, p1 = 0.0;
          if (!rows) {
          } else if constexpr (INPLACE) {
#pragma unroll
            for (int rr = 0; rr < H; ++rr) {
              // (register-only fences keep the rows in order: without them hipcc fetches all rows of the neighbour
              // first -- a third column in registers after all)
              asm volatile("" : "+v"(xa[rr]));
              const double pr = dpp_shift0<0x130>(xa[rr]);      // A of the right slot
              if (rr & 1) {
                p1 = fma(xb[rr], pr, p1);
                asm volatile("" : "+v"(p1));
              } else {
                p0 = fma(xb[rr], pr, p0);
                asm volatile("" : "+v"(p0));
              }
            }
          } else {
#pragma unroll
          for (int rr = 0; rr < H; ++rr) {
            xb[rr] = dpp_shift0<0x130>(xf[rr]);                 // A of the right slot
            if (rr & 1) p1 = fma(xa[rr], xb[rr], p1);
            else p0 = fma(xa[rr], xb[rr], p0);
          }
          }
          const double alAr = dpp_shift0<0x130>(alA), isAr = dpp_shift0<0x130>(isA), scAr = dpp_shift0<0x130>(scA);
          const double ga = slot_sum_nw<NW>(p0 + p1, lds, ph) * (isB * isAr);
          const double a = alB, b = alAr;
          const double g2 = ga * ga, ab = a * b;
          notconv |= __builtin_amdgcn_ballot_w64(g2 > stop_tol2 * ab) & hasRm;
          const bool rot = hasR && g2 > kRotTol2W * ab;
          const double h = 0.5 * (b - a);
          const double x = fma(h, h, g2);
          const double den = fma(x, fast_rsqrt1(x), fabs(h));
          double tt = ga * fast_rcp1(copysign(den, h));
          tt = rot ? tt : 0.0;
          if constexpr (EARLY)
            notconv2 |= (__builtin_amdgcn_ballot_w64(g2 > kEarlyTol2W * ab) | __builtin_amdgcn_ballot_w64(fabs(tt) > kEarlyTW)) & hasRm;
          const double w = fma(tt, tt, 1.0);
          const double c = fast_rsqrt(w);
          const double tg = tt * ga, wc = w * c;
          // what the right slot needs to rotate its A against this slot's (old) B
          const double q1 = dpp_shift0<0x138>(-tt * scB), q2 = dpp_shift0<0x138>(isB * c),
                       q3 = dpp_shift0<0x138>(scB * wc), q4 = dpp_shift0<0x138>(a - tg);
          const double coefR = hasR ? tt * (isB * scAr) : 1.0;   // position 2m+1 takes c (g_Ar + t g_B)
          const double coefL = hasL ? q1 * isA : 1.0;            // position 2m takes c (g_Bl - t g_A)
          if (hasR) {
            alB = b + tg;
            isB = isAr * c;
            scB = scAr * wc;
          }
          if (hasL) {
            alA = q4;
            isA = q2;
            scA = q3;
          }
          // new B into xb[] (on top of the fetched column); then the old B of the LEFT slot is pulled into xa[] --
          // every lane reads its neighbour's xa[rr] and overwrites its own in the same instruction -- and the new A
          // accumulated on top of it: A is back in xa[], B in xb[], xf[] is spare again
          if (!rows) {
          } else if constexpr (INPLACE) {
            // own B (xb) takes the right slot's A on top: xb <- A_r + coefR xb; own A (xa) the left slot's old B:
            // xa <- B_l + coefL xa -- both neighbours' elements are read before either own element of the row changes
#pragma unroll
            for (int rr = 0; rr < H; ++rr) {
              asm volatile("" : "+v"(xa[rr]), "+v"(xb[rr]));
              const double pr = dpp_shift0<0x130>(xa[rr]);
              const double ql = dpp_shift0<0x138>(xb[rr]);
              // (three-address forms on the element's own register: hipcc's v_fmac accumulates into the FETCHED value's
              // register and copies the result back, one v_mov_b64 per element)
              asm("v_fma_f64 %0, %1, %0, %2" : "+v"(xb[rr]) : "v"(coefR), "v"(pr));
              asm("v_fma_f64 %0, %1, %0, %2" : "+v"(xa[rr]) : "v"(coefL), "v"(ql));
            }
          } else {
#pragma unroll
          for (int rr = 0; rr < H; ++rr) {
            xb[rr] = fma(coefR, xa[rr], xb[rr]);
            xa[rr] = dpp_shift0<0x138>(xa[rr]);
            xa[rr] = fma(coefL, xf[rr], xa[rr]);
          }
          }
        }
        ++pairs;
        if constexpr (NW == 1) {
          quiet = notconv ? 0 : quiet + 1;
          if constexpr (EARLY) quiet2 = notconv2 ? 0 : quiet2 + 1;
        } else if constexpr (!EARLY) {
          quiet = __syncthreads_or(notconv != 0ull) ? 0 : quiet + 1;
        } else {
          // two waves, two votes, ONE barrier: each wave leaves its two ballots in LDS (behind the 2 x 128 doubles of
          // the inner-product exchange; double-buffered like it)
          int* fl = reinterpret_cast<int*>(lds + 256);
          const int mine = (notconv ? 1 : 0) | (notconv2 ? 2 : 0);
          if ((threadIdx.x & 63) == 0) fl[2 * vph + ((threadIdx.x >> 6) & 1)] = mine;
          __syncthreads();
          const int both = fl[2 * vph] | fl[2 * vph + 1];
          vph ^= 1;
          quiet = (both & 1) ? 0 : quiet + 1;
          quiet2 = (both & 2) ? 0 : quiet2 + 1;
        }
        done = quiet >= S || (EARLY && quiet2 >= S);
      }
    }
    sweep = (pairs + S - 1) / S;
    pairs_done = pairs;
    conv = done ? 1 : 0;
    if (rows) {
#pragma unroll
      for (int rr = 0; rr < H; ++rr) {
        xa[rr] *= isA;
        xb[rr] *= isB;
      }
    }
  }
  sweep = __builtin_amdgcn_readfirstlane(sweep);
  if (conv_out) *conv_out = __builtin_amdgcn_readfirstlane(conv);   // 0: the sweep cap ended the iteration
  if (pairs_out) *pairs_out = __builtin_amdgcn_readfirstlane(pairs_done);   // step pairs executed (the column order
                                                                             // after them is a fixed permutation)
  // ---- row-split -> column-per-lane
#pragma unroll
  for (int r0 = 0; r0 < KR; r0 += RC) {
    psync<NW>();
#pragma unroll
    for (int rr = r0 / 2; rr < (r0 + RC) / 2 && rr < H; ++rr)
      *reinterpret_cast<double2*>(&lds[(2 * rr + par - r0) * NL + 2 * slot]) = double2{xa[rr], xb[rr]};
    psync<NW>();
#pragma unroll
    for (int r = r0; r < r0 + RC && r < KR; ++r) g[r] = lds[(r - r0) * NL + lane];
  }
  psync<NW>();
  return sweep;
}

}  // namespace jacobi_dev
}  // namespace letkf
