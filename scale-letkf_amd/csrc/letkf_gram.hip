// letkf_gram.hip -- stage 1 of the staged path on the FP64 matrix cores, for the loop body of das_letkf (CSR lists into the
// observation table; scale/letkf/letkf_tools.f90:313-527).  What it computes is letkf_staged.hip's
// letkf_stage_gram_kernel (which stays for the dense letkf_core batch and for orders beyond this kernel): the symmetric
// matrix of the point's eigenproblem -- common/common_letkf.f90:111-147 --
//   dual   (n <  k): M = Z Z^T + c I  (n x n),  Z = diag(sqrt w) Y, contraction over the k members
//   primal (n >= k): A = Z^T Z + c I  (k x k),  contraction over the n local observations; plus r = Z^T (sqrt(w) dep) and
//                    r_det (common_letkf.f90:169-195 folded) as one more block column
// the adaptive-inflation sums (:233-249) and the per-observation scalars.
//
// Shape.  Both formulations are "P P^T" of a panel P (output index x contraction index) whose entries come from the rows
// of the observation table.  The output indices are cut into blocks of 16, the contraction into double-steps of 8; one
// panel ENTRY (block b, double-step t) is 1 KB = per lane (c, q) the pair P[16 b + c][8 t + 2 q], P[..][8 t + 2 q + 1]:
// .x is the operand of the matrix-core instruction that contracts indices 8 t + 2 q', q' < 4, .y of the one for
// 8 t + 2 q' + 1 -- and the SAME register is the A operand of tile (b, .) and the B operand of tile (., b)
// (v_mfma_f64_16x16x4: A lane (c, q) = A[c][q], B lane (c, q) = B[q][c]).  A chunk of Tc double-steps of all blocks sits
// in LDS in exactly that layout (one conflict-free ds_read_b128 per operand), every wave owns up to 12 tiles of the upper
// triangle (96 accumulator registers) and keeps the row block's operands in registers while it walks along a row of
// tiles; the next chunk's global loads are issued BEFORE the current chunk's matrix-core phase and written to LDS after
// it (single buffer, two barriers per chunk).  Orders with more than 96 tiles take several passes over the contraction.
//   dual:   entry (b, t), lane (c, q) = sqrt(w_i) Y[i][8 t + 2 q, + 1], i = 16 b + c: ONE 16-byte load (members contiguous)
//   primal: members are dealt to blocks in groups of 32 -- block 2 g holds the group's even members, 2 g + 1 the odd ones
//           -- so that one 16-byte load Y[row][32 g + 2 c, + 1] feeds both; the two rows of a pair are two loads
// Before: 4 x 4 register tiles on the vector ALUs over LDS tiles of 32 rows -- 2.66 ms per 3456 points of C3-slab (n = 200,
// k = 320), a third of the whole staged analysis once stages 2 and 3 ran on the matrix cores.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "letkf_staged_dev.h"

namespace letkf {

using namespace staged_dev;

namespace {

// primal: the members are dealt to blocks of 16 in groups of 32 -- block 2 g the group's even members, 2 g + 1 the odd ones (one
// 16-byte load feeds both) -- and a last group of at most 16 members is ONE block of its own (r4: MEMBER = 100 = 3 groups + 4
// members is 7 blocks instead of 8, 35 tiles instead of 44; two 8-byte loads per entry there)
#ifndef GRAM_SINGLE_TAIL
#define GRAM_SINGLE_TAIL 1
#endif
__host__ __device__ inline int primal_pairs(int k) { return (GRAM_SINGLE_TAIL && (k & 31) >= 1 && (k & 31) <= 16) ? (k >> 5) : ((k + 31) >> 5); }
__host__ __device__ inline int primal_blocks(int k) { return 2 * primal_pairs(k) + ((GRAM_SINGLE_TAIL && (k & 31) >= 1 && (k & 31) <= 16) ? 1 : 0); }

constexpr int kGmBlock = 512;
constexpr int kGmPanelBig = 68;         // panel entries (1 KB each) the LDS buffer holds: blocks x double-steps per chunk
constexpr int kGmPanelSmall = 40;       // ... of the four-wave instantiation (orders of at most 10 blocks x 4 double-steps)
constexpr int kGmItemsMax = 9;          // panel entries (dual) / pairs of entries (primal) a wave stages per chunk (registers)
// (primal: the observation rows whose offsets and scalars are worked out at a time -- a "super-chunk" in LDS -- are 1024, 512 in
// the four-wave instantiation)
#ifndef GRAM_SMALL_WAVES
#define GRAM_SMALL_WAVES 4               // (8: no four-wave instantiation -- A/B twins)
#endif
__host__ __device__ inline bool gram_small(int n, int k) {   // the four-wave instantiation's points
  if (GRAM_SMALL_WAVES != 4) return false;
  const int nbx = n < k ? (n + 15) >> 4 : primal_blocks(k) + 1;
  return n < k ? nbx * 4 <= 36 : nbx * 4 <= kGmPanelSmall;
}

struct alignas(16) d2 {
  double x, y;
};

}  // namespace


// LDS: panel [kGmPanel][64] x 16 B | swl [1024] | roff (long) [1024] | csd [2][1024] | red [8]
// DUAL: the instantiation for the points with n < k (the other one passes over them, and vice versa): the primal stages
// two loads per item and gets by with 8 accumulator tiles, the dual needs one load and takes 12 -- one kernel for both
// held 72 staging registers beside 96 accumulators and spilled the staged loads straight to scratch, i.e. waited for
// every load in front of the matrix-core phase (found in the ISA: 2.4 ms per 3456 points of C3-slab, no faster than the
// vector-ALU kernel).
// TC4: the instantiation for the points whose chunks hold 4 double-steps (orders <= 256: up to 17 blocks); the other one takes
// the larger orders with the generic inner loop -- both in one kernel cost the loops 90-130 spilled registers (the k = 100 loop
// body 1.60 -> 1.74 M solves/s with the split).
// NWV: waves of the workgroup.  4 (r4): the small orders -- at most 10 blocks incl. the extra one: MEMBER <= 128, n <= 144 in
// observation space -- as TWO workgroups of four waves per CU (up to 12 tiles per wave, panel of 40 entries, super-chunks of
// 512 rows: 56 KB of LDS), so that one point's chunk barriers and staging waits are the other point's matrix-core time; 8: the
// one-workgroup form for everything larger.
template <bool DUAL, bool TC4, int NWV>
__global__ void __launch_bounds__(64 * NWV, NWV == 4 ? 2 : 1) letkf_stage_gram_mfma_kernel(const StagedArgs S) {
  constexpr int kGmTiles = (DUAL || NWV == 4) ? 12 : 8;     // accumulator tiles per wave and pass
  constexpr int kGmPanel = NWV == 4 ? kGmPanelSmall : kGmPanelBig;
  constexpr int kGmSuper = NWV == 4 ? 512 : 1024;
  constexpr int kGmItems = (NWV == 4 && !DUAL) ? 6 : kGmItemsMax;   // (small primal points: at most 24 items per chunk over four waves)
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const PointArgs& A = S.A;
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int nwv = NWV;
  const int col = lane & 15, rq = lane >> 4;
  const int k = A.k;
  const double km1 = (double)(k - 1);
  d2* panel = reinterpret_cast<d2*>(smem);
  double* swl = smem + (size_t)kGmPanel * 128;        // sqrt(w_i): dual of every local observation (n <= 512), primal of the super-chunk's
  long* roff = reinterpret_cast<long*>(swl + kGmSuper);   // row offsets into the observation table, likewise
  double* csd = reinterpret_cast<double*>(roff + kGmSuper);   // primal: sqrt(w) dep, sqrt(w) dep_det of the super-chunk's rows
  double* red = csd + 2 * kGmSuper;

  for (long it = blockIdx.x; it < S.nbatch; it += gridDim.x) {
    const long pt = S.pt0 + it;
    Slab sl = slab_of(A.ws + (size_t)it * A.ws_per_block, k, A.nv, S.kkout);
    ObsView ov;
    ov.A = &A;
    ov.pt = pt;
    ov.o0 = A.obs_off[pt];
    ov.n = (int)(A.obs_off[pt + 1] - ov.o0);
    const double beta = A.beta ? A.beta[pt] : 1.0;
    const int n = ov.n;
    __syncthreads();
    if (!DUAL && (n < k || beta == 0.0)) continue;    // (points without an eigenproblem are the DUAL instantiation's, n = 0 < k)
    if (A.skip_trivial && (n == 0 || beta == 0.0)) {  // done by the streaming pass (letkf_trivial.hip): no stage touches it
      if (tid == 0) {
        S.meta[2 * it] = 255;
        S.meta[2 * it + 1] = -1;
      }
      continue;
    }
    if (beta == 0.0) {                                // letkf_tools.f90:333-359: nothing to solve
      if (tid == 0) {
        S.meta[2 * it] = 0;
        S.meta[2 * it + 1] = -1;
      }
      continue;
    }
    if (n > 0 && !gram_mfma_takes(n, k)) continue;    // letkf_stage_gram_kernel's point (launched behind this kernel)
    if (n > 0) {                                      // (points without an eigenproblem: the small TC4 instantiation's)
      const int nbx = n < k ? (n + 15) >> 4 : primal_blocks(k) + 1;
      const bool small = gram_small(n, k);
      if (small != (NWV == 4)) continue;
      if (!small && (kGmPanelBig / nbx >= 4) != TC4) continue;
    } else if (!(TC4 && NWV == GRAM_SMALL_WAVES)) continue;
    // inflation slot that drives the solve (first updated variable of the class, letkf_tools.f90:387-418)
    double infl_old;
    {
      bool qskip = false;
      if (A.q_update_top > 0.0) qskip = A.gues[pt * A.sp + k * A.sm + A.iv_p * A.sv] < A.q_update_top;
      int v0 = 0;
      while (v0 < A.nv && (!((A.var_mask >> v0) & 1u) || (qskip && v0 >= A.iv_q_first && v0 <= A.iv_q_last))) ++v0;
      infl_old = v0 < A.nv ? A.infl[pt + A.infl_sv * (long)v0] : 1.0;
    }
    const double shift = km1 / infl_old;              // common_letkf.f90:140-143
    const bool dual = n < k;
    if (dual != DUAL) continue;                       // (the other instantiation's point)
    const int m = dual ? n : k;
    if (tid == 0) {
      const int solver = m >= 2 ? (m <= S.wg_max_order ? 1 : 2) : 0;
      S.meta[2 * it] = (dual ? 2 : 1) | (solver << 8);
      S.meta[2 * it + 1] = m;
      sl.SC[3] = shift;
      sl.SC[4] = infl_old;
    }
    if (tid < 8) red[tid] = 0.0;
    __syncthreads();
    if (n == 0) {                                     // dual with an empty spectrum
      if (tid == 0) sl.SC[0] = sl.SC[1] = sl.SC[2] = 0.0;
      continue;
    }
    const int ldg = staged_ld(m);
    // ---- per-observation scalars: the adaptive-inflation sums; dual: sqrt w, the weighted departures, the row offsets
    {
      double p1 = 0.0, p3 = 0.0;
      for (int i = tid; i < n; i += nthr) {
        double w, d, dd, rl;
        ov.weights(i, w, d, dd, rl);
        p1 = fma(d * d, w, p1);
        p3 += rl;
        if (dual) {
          const double sw = sqrt(w);
          swl[i] = sw;
          sl.V0[i] = sw * d;
          sl.V1[i] = sw * dd;
          roff[i] = (long)A.obs_idx[ov.o0 + i] * A.kld;
        }
      }
      p1 = wsum(p1);
      p3 = wsum(p3);
      if (lane == 0) {
        atomicAdd(&red[0], p1);
        atomicAdd(&red[1], p3);
      }
    }
    // ---- blocks, tiles, chunks
    const int NB = dual ? (m + 15) >> 4 : primal_blocks(k);     // blocks of 16 output indices
    const int NP = dual ? 0 : primal_pairs(k);                   // primal: groups of 32 members = pairs of blocks; NB - 2 NP = 0 / 1 single block
    const int NBX = dual ? NB : NB + 1;                          // + the block column [sqrt(w) dep | sqrt(w) dep_det]
    const int Lc = dual ? k : n;                                 // contraction length
    int Tc = TC4 ? 4 : kGmPanel / NBX;                           // double-steps per chunk
    if (Tc > 4) Tc = 4;
    const int nchunk = (Lc + 8 * Tc - 1) / (8 * Tc);
    const int ntri = NB * (NB + 1) / 2, ntile = dual ? ntri : ntri + NB;
    // panel entries of a chunk, dealt round-robin to the waves: dual NB x Tc loads; primal (NB / 2) x Tc pairs of loads
    // (two entries each) + Tc entries of the extra block
    const int nitem = dual ? NB * Tc : (NB - NP) * Tc + Tc;      // (pairs, then the single block, then the extra block)
    __syncthreads();                                  // (swl, roff)

    double trp = 0.0;
    for (int t0p = 0; t0p < ntile; t0p += nwv * kGmTiles) {
      const int npass = ntile - t0p < nwv * kGmTiles ? ntile - t0p : nwv * kGmTiles;
      const int tpw = (npass + nwv - 1) / nwv;        // tiles per wave in this pass
      const int tw0 = t0p + wv * tpw;                 // this wave's first tile
      int ntw = ntile - tw0 < tpw ? ntile - tw0 : tpw;
      if (tw0 >= t0p + npass) ntw = 0;
      ntw = __builtin_amdgcn_readfirstlane(ntw < 0 ? 0 : ntw);
      // tile -> (I, J): upper triangle row by row, then (primal) the tiles (b, extra block)
      int tI[kGmTiles], tJ[kGmTiles];
#pragma unroll
      for (int ti = 0; ti < kGmTiles; ++ti) {
        int tl = tw0 + ti, I = 0, J = 0;
        if (ti < ntw) {
          if (tl < ntri) {
            while (tl >= NB - I) {                    // (a few iterations: scalar)
              tl -= NB - I;
              ++I;
            }
            J = I + tl;
          } else {
            I = tl - ntri;
            J = NB;
          }
        }
        tI[ti] = __builtin_amdgcn_readfirstlane(I);
        tJ[ti] = __builtin_amdgcn_readfirstlane(J);
      }
      d4 acc[kGmTiles];
#pragma unroll
      for (int ti = 0; ti < kGmTiles; ++ti) acc[ti] = (d4){0.0, 0.0, 0.0, 0.0};

      // staged loads of one chunk: up to kGmItems items per wave; an item = one entry (dual) or a pair of entries from
      // two 16-byte loads (primal; the extra block's entries need no load)
      d2u st0[kGmItems], st1[DUAL ? 1 : kGmItems];
      const int rows_ch = 8 * Tc;                     // contraction indices per chunk
      // primal: offsets and scalars of the observation rows [s0, s0 + kGmSuper) -- every thread a row, once per
      // kGmSuper rows (a chain of dependent loads: index -> weight, departure; per 32-row chunk it cost more than the chunk)
      auto super_rows = [&](int s0) {
        for (int j = tid; j < kGmSuper; j += nthr) {
          const int i = s0 + j;
          double w = 0.0, d = 0.0, dd = 0.0, rl;
          if (i < n) ov.weights(i, w, d, dd, rl);
          const double sw = sqrt(w);
          swl[j] = sw;                                // (0 beyond the last observation)
          csd[j] = sw * d;
          csd[kGmSuper + j] = sw * dd;
          roff[j] = (long)A.obs_idx[ov.o0 + (i < n ? i : n - 1)] * A.kld;
        }
      };
      auto issue = [&](int ch) {
#pragma unroll
        for (int u = 0; u < kGmItems; ++u) {
          const int item = wv + u * nwv;
          st0[u] = (d2u){0.0, 0.0};
          if constexpr (!DUAL) st1[u] = (d2u){0.0, 0.0};
          if (item < nitem) {                         // (wave-uniform)
            if constexpr (DUAL) {
              const int b = item / Tc, t = item - b * Tc;
              const int i = 16 * b + col;
              int e0 = (ch * Tc + t) * 8 + 2 * rq;
              e0 = e0 < k - 1 ? e0 : k - 1;           // (a row holds k + 1 doubles)
              st0[u] = *reinterpret_cast<const d2u*>(A.ensval + roff[i < n ? i : n - 1] + e0);
            } else if (item < NP * Tc) {
              const int g = item / Tc, t = item - g * Tc;
              int mm = 32 * g + 2 * col;
              mm = mm < k - 1 ? mm : k - 1;
              const int r0 = (ch * rows_ch) % kGmSuper + 8 * t + 2 * rq;   // super-chunk-relative rows r0, r0 + 1
              st0[u] = *reinterpret_cast<const d2u*>(A.ensval + roff[r0] + mm);
              st1[u] = *reinterpret_cast<const d2u*>(A.ensval + roff[r0 + 1] + mm);
            } else if (item < (NB - NP) * Tc) {       // the single last block: member 32 NP + col of rows r0, r0 + 1
              const int t = item - NP * Tc;
              int mm = 32 * NP + col;
              mm = mm < k ? mm : k;                   // (a row holds k + 1 doubles)
              const int r0 = (ch * rows_ch) % kGmSuper + 8 * t + 2 * rq;
              st0[u].x = A.ensval[roff[r0] + mm];
              st1[u].x = A.ensval[roff[r0 + 1] + mm];
            }
          }
        }
      };
      auto commit = [&](int ch) {
#pragma unroll
        for (int u = 0; u < kGmItems; ++u) {
          const int item = wv + u * nwv;
          if (item < nitem) {
            if constexpr (DUAL) {
              const int b = item / Tc, t = item - b * Tc;
              const int i = 16 * b + col;
              const int e0 = (ch * Tc + t) * 8 + 2 * rq;
              const double sw = i < n ? swl[i] : 0.0;
              d2 v;
              v.x = e0 < k ? st0[u].x * sw : 0.0;     // (e0 >= k - 1: the clamped load fetched [k - 1, k])
              v.y = e0 + 1 < k ? st0[u].y * sw : 0.0;
              panel[(size_t)(b * Tc + t) * 64 + lane] = v;
            } else if (item >= NP * Tc && item < (NB - NP) * Tc) {
              const int t = item - NP * Tc;
              const int mm = 32 * NP + col;
              const int r0 = (ch * rows_ch) % kGmSuper + 8 * t + 2 * rq;
              d2 v;
              v.x = mm < k ? st0[u].x * swl[r0] : 0.0;
              v.y = mm < k ? st1[u].x * swl[r0 + 1] : 0.0;
              panel[(size_t)((2 * NP) * Tc + t) * 64 + lane] = v;
            } else if (item < NP * Tc) {
              const int g = item / Tc, t = item - g * Tc;
              const int mm = 32 * g + 2 * col;
              const int r0 = (ch * rows_ch) % kGmSuper + 8 * t + 2 * rq;
              const double s0 = swl[r0], s1 = swl[r0 + 1];   // (0 beyond the last observation)
              d2 ve, vo;
              if (mm < k - 1) {
                ve.x = st0[u].x * s0;
                vo.x = st0[u].y * s0;
                ve.y = st1[u].x * s1;
                vo.y = st1[u].y * s1;
              } else {                                // (the clamped loads fetched [k - 1, k])
                ve.x = mm == k - 1 ? st0[u].x * s0 : 0.0;
                ve.y = mm == k - 1 ? st1[u].x * s1 : 0.0;
                vo.x = vo.y = 0.0;
              }
              panel[(size_t)((2 * g) * Tc + t) * 64 + lane] = ve;
              panel[(size_t)((2 * g + 1) * Tc + t) * 64 + lane] = vo;
            } else {                                  // extra block: column 0 = sqrt(w) dep, 1 = sqrt(w) dep_det
              const int t = item - (NB - NP) * Tc;
              const int r0 = (ch * rows_ch) % kGmSuper + 8 * t + 2 * rq;
              d2 v;
              v.x = col == 0 ? csd[r0] : col == 1 ? csd[kGmSuper + r0] : 0.0;
              v.y = col == 0 ? csd[r0 + 1] : col == 1 ? csd[kGmSuper + r0 + 1] : 0.0;
              panel[(size_t)(NB * Tc + t) * 64 + lane] = v;
            }
          }
        }
      };

      // (kGmSuper is a multiple of every chunk length 8 Tc <= 32: a chunk never straddles two super-chunks)
      if constexpr (!DUAL) {
        __syncthreads();                              // (a previous pass may still read swl / roff / csd)
        super_rows(0);
        __syncthreads();
      }
      issue(0);
      for (int ch = 0; ch < nchunk; ++ch) {
        __syncthreads();                              // (the previous chunk's readers are done with the panel)
        commit(ch);
        if constexpr (!DUAL) {
          // the next chunk opens a new super-chunk: its rows' offsets and scalars replace the current ones -- behind
          // this chunk's commit (which read them), in front of the next issue
          if (ch + 1 < nchunk && ((ch + 1) * rows_ch) % kGmSuper == 0) {
            __syncthreads();
            super_rows((ch + 1) * rows_ch);
          }
        }
        __syncthreads();
        if (ch + 1 < nchunk) issue(ch + 1);           // in flight during the matrix-core phase below
        if constexpr (TC4) {
          // the common shape (orders <= 256): no condition on the double-step inside a tile, and TWO tiles side by side -- the
          // eight instructions of a tile accumulate into one register quad, a dependent chain; two chains interleave
#pragma unroll
          for (int ti = 0; ti < kGmTiles; ti += 2) {
            if (ti + 1 < ntw) {                       // (wave-uniform)
              const int I0 = tI[ti], J0 = tJ[ti], I1 = tI[ti + 1], J1 = tJ[ti + 1];
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                const d2 a0 = panel[(size_t)(I0 * 4 + t) * 64 + lane];
                const d2 b0 = panel[(size_t)(J0 * 4 + t) * 64 + lane];
                const d2 a1 = panel[(size_t)(I1 * 4 + t) * 64 + lane];
                const d2 b1 = panel[(size_t)(J1 * 4 + t) * 64 + lane];
                acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, b0.x, acc[ti], 0, 0, 0);
                acc[ti + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, b1.x, acc[ti + 1], 0, 0, 0);
                acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, b0.y, acc[ti], 0, 0, 0);
                acc[ti + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, b1.y, acc[ti + 1], 0, 0, 0);
              }
            } else if (ti < ntw) {
              const int I = tI[ti], J = tJ[ti];
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                const d2 a = panel[(size_t)(I * 4 + t) * 64 + lane];
                const d2 b = panel[(size_t)(J * 4 + t) * 64 + lane];
                acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.x, acc[ti], 0, 0, 0);
                acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b.y, acc[ti], 0, 0, 0);
              }
            }
          }
        } else {
          int curI = -1;
          d2 pI[4];
#pragma unroll
          for (int ti = 0; ti < kGmTiles; ++ti) {
            if (ti < ntw) {                           // (wave-uniform)
              const int I = tI[ti], J = tJ[ti];
              if (I != curI) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                  if (t < Tc) pI[t] = panel[(size_t)(I * Tc + t) * 64 + lane];
                curI = I;
              }
#pragma unroll
              for (int t = 0; t < 4; ++t)
                if (t < Tc) {
                  const d2 pJ = panel[(size_t)(J * Tc + t) * 64 + lane];
                  acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(pI[t].x, pJ.x, acc[ti], 0, 0, 0);
                  acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(pI[t].y, pJ.y, acc[ti], 0, 0, 0);
                }
            }
          }
        }
      }
      // ---- finished tiles -> G (both triangles), shift on the diagonal; primal: the extra tiles are r, r_det
      // (D: lane (c, q) holds rows q + 4 r of column c of the tile)
#pragma unroll
      for (int ti = 0; ti < kGmTiles; ++ti) {
        if (ti < ntw) {
          const int I = tI[ti], J = tJ[ti];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int li = rq + 4 * r;                // local row of block I, local column col of block J
            const double v = acc[ti][r];
            if (dual) {
              const int gi = 16 * I + li, gj = 16 * J + col;
              if (gi < m && gj < m) {
                if (gi == gj) {
                  trp += v;
                  sl.G[(size_t)gj * ldg + gi] = v + shift;
                } else {
                  sl.G[(size_t)gi * ldg + gj] = v;
                  if (I != J) sl.G[(size_t)gj * ldg + gi] = v;
                }
              }
            } else {
              const int gi = I < 2 * NP ? 32 * (I >> 1) + 2 * li + (I & 1) : 32 * NP + li;   // member of (block, local index)
              if (J == NB) {
                if (gi < k && col < 2) (col == 0 ? sl.V0 : sl.V1)[gi] = v;
              } else {
                const int gj = J < 2 * NP ? 32 * (J >> 1) + 2 * col + (J & 1) : 32 * NP + col;
                if (gi < k && gj < k) {
                  if (gi == gj) {
                    trp += v;
                    sl.G[(size_t)gj * ldg + gi] = v + shift;
                  } else {
                    sl.G[(size_t)gi * ldg + gj] = v;
                    if (I != J) sl.G[(size_t)gj * ldg + gi] = v;
                  }
                }
              }
            }
          }
        }
      }
    }
    // adaptive-inflation sums (common_letkf.f90:233-249): sum dep^2 w, sum rloc, trace(Z^T Z) = trace(Z Z^T)
    trp = wsum(trp);
    if (lane == 0) atomicAdd(&red[2], trp);
    __syncthreads();
    if (tid == 0) {
      sl.SC[0] = red[0];
      sl.SC[1] = red[1];
      sl.SC[2] = red[2];
      // eigen-free stage (letkf_krylov.hip), as letkf_stage_gram_kernel decides it
      const double tr = red[2];
      if (S.poly_max_n > 0 && m >= 2 && m <= S.poly_max_n && tr == tr && tr <= 1e5 * shift * (double)m)
        S.meta[2 * it] = (dual ? 2 : 1) | (3 << 8);
    }
  }
}

hipError_t launch_stage_gram_mfma(const StagedArgs& s, hipStream_t st) {
  auto go = [&](auto kern, int nwv) -> hipError_t {
    const size_t lds = ((size_t)(nwv == 4 ? kGmPanelSmall : kGmPanelBig) * 128 + 4 * (size_t)(nwv == 4 ? 512 : 1024) + 8) * sizeof(double);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)s.nbatch), dim3(64 * nwv), lds, st, s);
    return hipGetLastError();
  };
  hipError_t e = go(&letkf_stage_gram_mfma_kernel<true, true, 4>, 4);   // small orders; n < k first (and every point without an eigenproblem)
  if (e == hipSuccess) e = go(&letkf_stage_gram_mfma_kernel<false, true, 4>, 4);
  if (e == hipSuccess && !gram_small(s.A.k, s.A.k)) {                   // (MEMBER <= 128: every point is small)
    e = go(&letkf_stage_gram_mfma_kernel<true, true, 8>, 8);
    if (e == hipSuccess) e = go(&letkf_stage_gram_mfma_kernel<false, true, 8>, 8);
  }
  if (e == hipSuccess && s.A.k > 256) {                                 // orders beyond 17 blocks: chunks of fewer double-steps
    e = go(&letkf_stage_gram_mfma_kernel<true, false, 8>, 8);
    if (e == hipSuccess) e = go(&letkf_stage_gram_mfma_kernel<false, false, 8>, 8);
  }
  return e;
}

}  // namespace letkf
