// letkf_eig.hip -- eigen-decomposition stage of the staged (three-kernel) path for matrix orders 63 .. 208:
// one WORKGROUP per symmetric positive-definite matrix, the matrix resident in the CU's registers (+ part of it in LDS)
// for the whole iteration.  Replaces common/common_mtx.f90:41 (mtx_eigen -> EISPACK rs, common/netlib.f:524) for the
// k x k problem of a grid point (63 <= k) and for the n x n observation-space problem when a point has fewer local
// observations than members (letkf_staged.hip explains the latter).
//
// Algorithm: the same one-sided (Hestenes) Jacobi with odd-even transposition ordering and rotate-and-swap as the
// one-wave kernel (letkf_jacobi_dev.h), scaled out to a workgroup:
//   * a SLOT owns the two columns at line positions (2s, 2s+1); slot s = lane s of a wavefront (SB = 1: 64 slots,
//     m <= 128) or lane (s & 63) of slot-block (s >> 6) (SB = 2: 128 slots);
//   * the rows are split 8 ways: wave (part p, slot-block b) holds rows [p RP, (p+1) RP) of the 2 x 64 columns of its
//     slots: 8 SB waves per matrix, RP <= 26 doubles of each column per lane -- the register budget of 16 waves per CU
//     (128 VGPRs each) is what sets this shape.  The lower column of a slot (A) lives in registers, of the upper
//     column (B) the first RBR rows in registers and the rest in LDS ([row][thread]: conflict-free);
//   * even steps pair the two columns of a slot: no data moves; odd steps pair the upper column of slot s with the
//     lower column of slot s+1: the register-resident rows travel by DPP wave_shl:1 / wave_shr:1 inside a wave, the
//     LDS rows of the left neighbour are read at thread - 1; only the one slot pair that straddles the two
//     slot-blocks goes through a small LDS mailbox;
//   * inner products are reduced across the 8 parts through LDS and a workgroup barrier; every part sums the 8
//     partials in the same order, so all parts compute bit-identical rotations (no broadcast needed);
//   * squared norms follow the rotation identities and are refreshed once per sweep; convergence = a full cycle of
//     the ordering in which no pair had |cos| > 1e-10 (the rule of the one-wave kernel, kStopTol2W).
// Barriers per step pair: 2 (SB = 1) or 3 (SB = 2).  No lane ever diverges around a barrier: unused slots carry
// zero columns and take part in everything.
//
// Input / output: G (m x m, column-major, leading dimension ldg) in the point's workspace slab; on return column j
// holds lambda_j v_j (any order) -- the convention of the other Jacobi variants (letkf_kernels.hip).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "letkf_device.h"
#include "letkf_jacobi_dev.h"
#include "letkf_staged_dev.h"

#ifndef EIG_RBR2
#define EIG_RBR2 50
#endif
#ifndef EIG_PF
#define EIG_PF 2
#endif
#ifndef EIG_PM
#define EIG_PM 2
#endif

namespace letkf {

namespace {

typedef __attribute__((address_space(3))) double lds_double;

using jacobi_dev::dpp_shift0;
using jacobi_dev::fast_rcp1;
using jacobi_dev::fast_rsqrt;
using jacobi_dev::fast_rsqrt1;

struct Rot {
  double tt, c, tg;   // tangent, cosine, t * gamma
  bool notconv;      // |cos| > 1e-10
  bool notconv2;     // |cos| > 1e-8 or |t| > 1e-6: the early-stop rule of letkf_jacobi_dev.h (kEarlyTol2W, kEarlyT2W)
};

// rotation that orthogonalises (lower column: norm^2 a, upper column: norm^2 b, inner product g); letkf_jacobi_dev.h
__device__ __forceinline__ Rot make_rot(const double a, const double b, const double ga, const bool enable) {
  Rot r;
  const double g2 = ga * ga, ab = a * b;
  r.notconv = enable && g2 > jacobi_dev::kStopTol2W * ab;
  const bool rot = enable && g2 > jacobi_dev::kRotTol2W * ab;
  const double d = b - a;
  const double x = fma(d, d, 4.0 * g2);
  const double hh = x * fast_rsqrt1(x);
  double tt = (2.0 * ga) * copysign(1.0, d) * fast_rcp1(fabs(d) + hh);
  tt = rot ? tt : 0.0;
  r.notconv2 = enable && (g2 > jacobi_dev::kEarlyTol2W * ab || tt * tt > jacobi_dev::kEarlyT2W);
  const double w = fma(tt, tt, 1.0);
  r.tt = tt;
  r.c = fast_rsqrt(w);
  r.tg = tt * ga;
  return r;
}

// hipcc hoists every LDS read of an unrolled row loop to its top (17 doubles in flight, 34 VGPRs -- with 128 VGPRs per
// lane that spills the columns themselves); a compiler-level fence every 8 rows keeps the loads near their use (8 in flight)
#define EIG_ROW_FENCE(rr) \
  if (((rr) & 7) == 7) asm volatile("" ::: "memory")
// the single-lane mailbox loops: rows in pairs, EIG_FIX_ROWS rows of loads in flight
#ifndef EIG_FIX_ROWS
#define EIG_FIX_ROWS 8
#endif
#define EIG_FIX_FENCE(rr) \
  if (((rr) + 2) % EIG_FIX_ROWS == 0) asm volatile("" ::: "memory")

// The two row updates, written as instruction sequences that work IN PLACE.  From C++ hipcc puts every new element of
// the lower column into a fresh register (the old one is still an operand of the upper column's update), i.e. it
// renames the whole column per step, and the copies / live ranges that follow cost ~90 spilled registers at 52 rows
// per lane (found with -Rpass-analysis and in the ISA).
// The rotations are the SCALED ("fast") ones of letkf_jacobi_dev.h: a column is kept as stored * is (is = the product
// of the cosines since the last refresh, sc = 1 / is carried along), so a rotated column is H + coef * G -- one FMA per
// element instead of a multiply and an FMA.
//   even step:  a <- b + cA a ;  b <- a_old + cB b
__device__ __forceinline__ void rot_inplace(double& a, double& b, const double cA, const double cB) {
  double t;
  asm("v_mov_b64 %2, %0\n\t"
      "v_fma_f64 %0, %3, %0, %1\n\t"
      "v_fma_f64 %1, %4, %1, %2"
      : "+v"(a), "+v"(b), "=&v"(t)
      : "v"(cA), "v"(cB));
}
//   odd step:   x <- y + f x   (on the element's own register: hipcc's v_fmac accumulates into y's register and copies back)
__device__ __forceinline__ void xpay_inplace(double& x, const double f, const double y) {
  asm("v_fma_f64 %0, %1, %0, %2" : "+v"(x) : "v"(f), "v"(y));
}

#ifdef LETKF_WAVE_PROF
__device__ unsigned long long g_eig_prof[8];
#define EP_DECL() unsigned long long epa[4] = {0, 0, 0, 0}; unsigned long long ept0 = __builtin_readcyclecounter()
#define EP_SYNC(i) do { const unsigned long long t0_ = __builtin_readcyclecounter(); __syncthreads(); epa[i] += __builtin_readcyclecounter() - t0_; } while (0)
#define EP_FLUSH() do { if (lane == 0) { for (int i = 0; i < 4; ++i) atomicAdd(&g_eig_prof[i], epa[i]); atomicAdd(&g_eig_prof[4], __builtin_readcyclecounter() - ept0); } } while (0)
#else
#define EP_DECL() do {} while (0)
#define EP_SYNC(i) __syncthreads()
#define EP_FLUSH() do {} while (0)
#endif
constexpr int kPF = EIG_PF;   // LDS rows in flight ahead of their use

template <int NP>
__device__ __forceinline__ double sum_parts(const double* p, const int stride) {
  double s = p[0];
#pragma unroll
  for (int i = 1; i < NP; ++i) s += p[i * stride];
  return s;
}

}  // namespace

size_t eig_wg_lds_bytes(int NP, int RP, int RBR, int SB) {
  const size_t NT = 64 * (size_t)NP * SB, NS = 64 * (size_t)SB;
  return 8 * ((size_t)(RP - RBR) * NT + 2 * (size_t)NP * NS + 8 * NS + 2 * (size_t)NP * RP) + 2 * 16 * sizeof(int) + 64;
}

// NP row parts x SB slot-blocks = NP * SB waves per matrix; 2 waves per SIMD (256 VGPRs per lane) either way:
// <4, 32, 32, 1>: m <= 128, 4 waves, two matrices in flight per CU; <4, 52, 18, 2>: m <= 208, 8 waves, one per CU.
template <int NP, int RP, int RBR, int SB>
__global__ void __launch_bounds__(64 * NP * SB, SB == 1 ? 2 : 1) letkf_eig_wg_kernel(const EigArgs E) {
  constexpr int kParts = NP;
  constexpr int NT = 64 * NP * SB, NS = 64 * SB, RBL = RP - RBR, MMAX = kParts * RP;
  static_assert(RBR >= 0 && RBR <= RP, "rows of B in registers");
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* blds = sm;                       // [RBL][NT]   rows RBR.. of the upper columns
  double* pe = blds + (size_t)RBL * NT;    // [8][NS]     partial inner products, even steps (and the norm refresh)
  double* po = pe + kParts * NS;           // [8][NS]     ... odd steps
  double* pn = po + kParts * NS;           // [NS][4]     squared norms and scales (lower, upper) of every slot after its even step
  double* st = pn + 4 * NS;                // [NS][4]     the slots' norms and scales BETWEEN step pairs (see the even step)
  double* bbP = st + 4 * NS;               // [8][RP]     mailbox: lower column of slot 64 (for slot 63)
  double* bbQ = bbP + kParts * RP;         // [8][RP]     mailbox: upper column of slot 63 (for slot 64)
  int* flags = reinterpret_cast<int*>(bbQ + kParts * RP);   // [2][16] convergence votes of the waves

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: the branches on part / sb below are wave-uniform)
  const int part = wave / SB, sb = wave % SB;
  const int slot = sb * 64 + lane;
  const bool bndR = SB == 2 && sb == 0 && lane == 63;   // slot 63: its right neighbour sits in the other slot-block
  const bool bndL = SB == 2 && sb == 1 && lane == 0;    // slot 64
  const int row0 = part * RP;
  // LDS rows of a thread are NT * 8 bytes apart and a DS instruction reaches 64 KB from its address register: left to
  // itself hipcc gives every row beyond that its own (hoisted) address register -- 20 VGPRs at 36 rows.  One laundered
  // base per KROW rows instead, every access an immediate offset from one of them.
  static_assert(SB == 1 || RP % 2 == 0, "mailbox rows are read in pairs");
  constexpr int KROW = 65536 / (NT * 8), NBASE = RBL > 0 ? (RBL + KROW - 1) / KROW : 1;
  lds_double* bt[NBASE];   // the thread's own rows
#pragma unroll
  for (int i = 0; i < NBASE; ++i) {
    unsigned o = (unsigned)(uintptr_t)(lds_double*)(blds + (size_t)i * KROW * NT + tid);
    asm volatile("" : "+v"(o));
    bt[i] = (lds_double*)(uintptr_t)o;
  }
#define BT(r) bt[(r) / KROW][((r) % KROW) * NT]

  EP_DECL();
  for (long it = blockIdx.x; it < E.npts; it += gridDim.x) {
    const long pt = E.pt0 + it;
    const int m = E.meta[2 * it + 1];
    const int solver = E.meta[2 * it] >> 8;              // 1: this kernel
    if (solver != 1 || m < 2 || m > MMAX) continue;     // (uniform for the workgroup)
    const int ldg = staged_dev::staged_ld(m);
    double* __restrict__ G = E.ws + (size_t)it * E.ws_per_point;
    const int ncol = (m + 1) & ~1, S = ncol >> 1;
    const bool hasL = slot > 0 && slot < S, hasR = slot + 1 < S;

    // ---- load: lower column 2s -> a[], upper column 2s+1 -> breg[] / blds
    double a[RP], breg[RBR > 0 ? RBR : 1];
    {
      // (unconditional loads from clamped addresses, zeroed by a select: a conditional load costs an exec-mask
      // branch each, and hipcc spills the hoisted addresses of all 2 RP of them)
      const int ca = 2 * slot, cb = 2 * slot + 1;
      const bool oka = ca < m, okb = cb < m;
      const double* ga = G + (size_t)(oka ? ca : 0) * ldg;
      const double* gb = G + (size_t)(okb ? cb : 0) * ldg;
#pragma unroll
      for (int rr = 0; rr < RP; ++rr) {
        const int row = row0 + rr;
        const int rc = row < m ? row : m - 1;
        const double va = ga[rc], vb = gb[rc];
        a[rr] = (oka && row < m) ? va : 0.0;
        const double vbz = (okb && row < m) ? vb : 0.0;
        if (rr < RBR) breg[rr] = vbz;
        else BT(rr - RBR) = vbz;
        EIG_ROW_FENCE(rr);
      }
    }
    // A slot's squared norms (alA, alB) and scales (column = stored * is) are the same in all its row parts, change only
    // in the two short scalar phases of a step pair and are needed nowhere else: they LIVE IN LDS (st between step pairs,
    // pn between the even and the odd step, written by part 0) instead of in eight registers across the row loops --
    // four more rows of the upper column in registers.
    if (part == 0) {
      st[4 * slot] = 0.0;
      st[4 * slot + 1] = 0.0;
      st[4 * slot + 2] = 1.0;
      st[4 * slot + 3] = 1.0;
    }
    int quiet = 0, quiet2 = 0, pairs = 0, sweep = 0;
    bool done = false, odd_notconv_prev = false, odd_notconv2_prev = false;
    int vph = 0;
    // Lanes of unused slots hold zero columns and are SWITCHED OFF inside the row loops (never around a barrier): the
    // scaled rotations below write x <- y + f x, so a fetched neighbour can no longer be multiplied away, but a DPP
    // read from a disabled lane returns 0 (bound_ctrl) and their LDS rows stay the zeros of the load -- "no partner".
    const bool act = slot < S;
    __syncthreads();

    for (; sweep < E.max_sweep && !done; ++sweep) {
      // ---- refresh: fold the scales back into the columns, recompute the squared norms (inside a sweep they are
      // carried by the rotation identities)
      {
        if (sweep > 0) __syncthreads();                   // (st was written by part 0 in the last odd step)
        const double isA = st[4 * slot + 2], isB = st[4 * slot + 3];
        double sa = 0.0, sbq = 0.0, sab = 0.0;
#pragma unroll
        for (int rr = 0; rr < RP; ++rr) {
          a[rr] *= isA;
          double vb;
          if (rr < RBR) {
            breg[rr < RBR ? rr : 0] *= isB;
            vb = breg[rr < RBR ? rr : 0];
          } else {
            vb = BT(rr - RBR) * isB;
            BT(rr - RBR) = vb;
          }
          sa = fma(a[rr], a[rr], sa);
          sbq = fma(vb, vb, sbq);
          sab = fma(a[rr], vb, sab);
          EIG_ROW_FENCE(rr);
        }
        pe[part * NS + slot] = sa;
        po[part * NS + slot] = sbq;
        __syncthreads();
        if (part == 0) {
          st[4 * slot] = sum_parts<NP>(pe + slot, NS);
          st[4 * slot + 1] = sum_parts<NP>(po + slot, NS);
          st[4 * slot + 2] = 1.0;
          st[4 * slot + 3] = 1.0;
        }
        __syncthreads();
        pe[part * NS + slot] = sab;                       // the first even step's inner product (see the odd step)
      }
      for (int t = 0; t < ncol && !done; t += 2) {
        // ================= even step: the slot's own two columns.  Their inner product was accumulated while the
        // previous odd step (or the norm refresh) had both columns in its hands: no pass of its own
        EP_SYNC(0);                                       // (1)
        bool notconv, notconv2;
        {
          const double alA = st[4 * slot], alB = st[4 * slot + 1], isA = st[4 * slot + 2], isB = st[4 * slot + 3];
          const double ga = sum_parts<NP>(pe + slot, NS) * (isA * isB);
          const Rot r = make_rot(alA, alB, ga, true);
          notconv = r.notconv;
          notconv2 = r.notconv2;
          // rotate and swap: position 2s takes c (B + t A), position 2s+1 takes c (A - t B)
          // (coefficients t isA / isB and -t isB / isA through ONE reciprocal; carrying 1 / is along as the register
          // kernels do costs four more registers per lane, which this kernel does not have)
          const double rab = r.tt * jacobi_dev::fast_rcp(isA * isB);
          const double cA = (isA * isA) * rab, cB = -(isB * isB) * rab;
          if (part == 0) {
            pn[4 * slot] = alB + r.tg;
            pn[4 * slot + 1] = alA - r.tg;
            pn[4 * slot + 2] = isB * r.c;
            pn[4 * slot + 3] = isA * r.c;
          }
          // LDS rows: the loads run kPF rows ahead of their use, written out in the source -- hipcc cannot move a row's
          // read above the previous row's write (every row has its own address register: no alias information), and
          // left alone it emits read, s_waitcnt lgkmcnt(0), use for every single row
          if (act) {
            double pf[kPF];
#pragma unroll
            for (int u = 0; u < kPF; ++u)
              if (RBR + u < RP) pf[u] = BT(u);
#pragma unroll
            for (int rr = 0; rr < RP; ++rr) {
              if (rr < RBR) {
                rot_inplace(a[rr], breg[rr < RBR ? rr : 0], cA, cB);
              } else {
                double vb = pf[(rr - RBR) % kPF];
                if (rr + kPF < RP) pf[(rr - RBR) % kPF] = BT(rr + kPF - RBR);
                rot_inplace(a[rr], vb, cA, cB);
                BT(rr - RBR) = vb;
              }
            }
          }
        }
        if constexpr (SB == 2) {
          // mailbox for the slot pair (63, 64) that straddles the two slot-blocks (written by an unused slot 64 as
          // well: its zeros are slot 63's "no partner")
          if (bndL) {
#pragma unroll
            for (int rr = 0; rr < RP; ++rr) bbP[part * RP + rr] = a[rr];
          }
          if (bndR) {
#pragma unroll
            for (int rr = 0; rr < RP; ++rr)
              bbQ[part * RP + rr] = rr < RBR ? breg[rr < RBR ? rr : 0] : BT(rr - RBR);
          }
          EP_SYNC(1);                                     // (2)
        }
        // ================= odd step: upper column of slot s (Q_s) with the lower column of slot s+1 (P_{s+1}).
        // The row loops below carry no branch, so that hipcc can issue the LDS reads of a fence group together (with a
        // conditional mailbox read per row every LDS read was followed by its own s_waitcnt lgkmcnt(0)).  The slot pair
        // that straddles the slot-blocks: in the row loops the two lanes concerned see "no partner" (their DPP fetch
        // returns 0), and each makes up for it in a short loop of its own afterwards -- everything is linear in the
        // missing column.  (Before: an unconditional mailbox term in every lane, i.e. 2 RP broadcast reads per wave and
        // step pair for two lanes of the workgroup; the LDS pipe is this kernel's bottleneck.)
        {
          double p0 = 0.0, p1 = 0.0;
          if (act) {
            double pf[kPF];
#pragma unroll
            for (int u = 0; u < kPF; ++u)
              if (RBR + u < RP) pf[u] = BT(u);
#pragma unroll
            for (int rr = 0; rr < RP; ++rr) {
              // (register-only fences: without them hipcc fetches ALL rows of the neighbour first -- a third column in
              // registers; volatile asm statements keep their order, so each row's fetch waits for the previous row's FMA)
              asm volatile("" : "+v"(a[rr]));
              const double pr = dpp_shift0<0x130>(a[rr]);    // lane + 1 (0 for lane 63 and for a switched-off lane)
              double q;
              if (rr < RBR) {
                q = breg[rr < RBR ? rr : 0];
              } else {
                q = pf[(rr - RBR) % kPF];
                if (rr + kPF < RP) pf[(rr - RBR) % kPF] = BT(rr + kPF - RBR);
              }
              if (rr & 1) {
                p1 = fma(q, pr, p1);
                asm volatile("" : "+v"(p1));
              } else {
                p0 = fma(q, pr, p0);
                asm volatile("" : "+v"(p0));
              }
            }
          }
          if constexpr (SB == 2) {
            if (bndR && act) {                                     // slot 63: Q_63 . P_64, the lower column of slot 64 from the mailbox
              const double2* mb = reinterpret_cast<const double2*>(bbP + part * RP);
#pragma unroll
              for (int rr = 0; rr < RP; rr += 2) {
                const double2 mv = mb[rr / 2];
                const double q0 = rr < RBR ? breg[rr < RBR ? rr : 0] : BT(rr - RBR);
                const double q1 = rr + 1 < RBR ? breg[rr + 1 < RBR ? rr + 1 : 0] : BT(rr + 1 - RBR);
                p0 = fma(q0, mv.x, p0);
                p1 = fma(q1, mv.y, p1);
                EIG_FIX_FENCE(rr);
              }
            }
          }
          const double podd = p0 + p1;
          po[part * NS + slot] = podd;
          // vote: this step pair's even step and the previous step pair's odd step
          const int anyv = (__any(notconv || odd_notconv_prev) ? 1 : 0) | (__any(notconv2 || odd_notconv2_prev) ? 2 : 0);
          if (lane == 0) flags[16 * vph + wave] = anyv;
        }
        EP_SYNC(2);                                       // (3)
        {
          // as the left member of the pair (s, s+1), and again as the right member of (s-1, s)
          const int sr = slot + 1 < NS ? slot + 1 : slot, sl = slot > 0 ? slot - 1 : 0;
          const double alA = pn[4 * slot], alB = pn[4 * slot + 1], isA = pn[4 * slot + 2], isB = pn[4 * slot + 3];
          const double nPr = pn[4 * sr], isPr = pn[4 * sr + 2];                // |P_{s+1}|^2 and its scale
          const double nQl = pn[4 * sl + 1], isQl = pn[4 * sl + 3];            // |Q_{s-1}|^2 and its scale
          const double gR = sum_parts<NP>(po + slot, NS) * (isB * isPr);
          const double gL = sum_parts<NP>(po + sl, NS) * (isQl * isA);
          const double scPr = jacobi_dev::fast_rcp(isPr), scQl = jacobi_dev::fast_rcp(isQl);
          const Rot rR = make_rot(alB, nPr, gR, hasR);     // lower = Q_s, upper = P_{s+1}
          const Rot rL = make_rot(nQl, alA, gL, hasL);     // lower = Q_{s-1}, upper = P_s
          odd_notconv_prev = rR.notconv;
          odd_notconv2_prev = rR.notconv2;
          // new upper column of slot s (position 2s+1) = c (P_{s+1} + t Q_s); new lower column (position 2s) = c (Q_{s-1} - t P_s)
          const double coefR = hasR ? rR.tt * (isB * scPr) : 1.0;
          const double coefL = hasL ? -rL.tt * (isA * scQl) : 1.0;
          if (part == 0) {
            st[4 * slot] = hasL ? nQl - rL.tg : alA;
            st[4 * slot + 1] = hasR ? nPr + rR.tg : alB;
            st[4 * slot + 2] = hasL ? isQl * rL.c : isA;
            st[4 * slot + 3] = hasR ? isPr * rR.c : isB;
          }
          int anyf = 0;
#pragma unroll
          for (int w = 0; w < NP * SB; ++w) anyf |= flags[16 * vph + w];
          anyf = __builtin_amdgcn_readfirstlane(anyf);     // (the same words in every lane: quiet / done live in SGPRs)
          vph ^= 1;
          double fe0 = 0.0, fe1 = 0.0;
          if (act) {
            double pfq[kPF];
#pragma unroll
            for (int u = 0; u < kPF; ++u)
              if (RBR + u < RP) pfq[u] = BT(u);
#pragma unroll
            for (int rr = 0; rr < RP; ++rr) {
              // the neighbour's rows are fetched a second time (once for the inner product, once here); laundering a[rr]
              // keeps hipcc from merging the two fetches across the barrier, and orders this row behind the previous one
              asm volatile("" : "+v"(a[rr]));
              const double pr = dpp_shift0<0x130>(a[rr]);
              // (the left neighbour's element comes from the lane next door for the LDS rows too -- it has just loaded it:
              // two vector moves instead of a second LDS read; the LDS pipe is this kernel's bottleneck)
              double q;
              if (rr < RBR) {
                q = breg[rr < RBR ? rr : 0];
              } else {
                q = pfq[(rr - RBR) % kPF];
                if (rr + kPF < RP) pfq[(rr - RBR) % kPF] = BT(rr + kPF - RBR);
              }
              const double ql = dpp_shift0<0x138>(q);        // lane - 1 (0 for lane 0)
              if (rr < RBR) {
                xpay_inplace(breg[rr < RBR ? rr : 0], coefR, pr);
              } else {
                xpay_inplace(q, coefR, pr);
                BT(rr - RBR) = q;
              }
              xpay_inplace(a[rr], coefL, ql);
              // both new columns of the slot are in registers here: the next even step's inner product, for free
              {
                const double qn = rr < RBR ? breg[rr < RBR ? rr : 0] : q;
                if (rr & 1) {
                  fe1 = fma(a[rr], qn, fe1);
                  asm volatile("" : "+v"(fe1));              // (keeps the rows in order: see the odd step's inner product)
                } else {
                  fe0 = fma(a[rr], qn, fe0);
                  asm volatile("" : "+v"(fe0));
                }
              }
            }
          }
          if constexpr (SB == 2) {
            if (bndR && act) {                                     // slot 63: new upper column = (P_64 +) coefR Q_63
              const double2* mb = reinterpret_cast<const double2*>(bbP + part * RP);
#pragma unroll
              for (int rr = 0; rr < RP; rr += 2) {
                const double2 mv = mb[rr / 2];
                if (rr < RBR) breg[rr < RBR ? rr : 0] += mv.x;
                else BT(rr - RBR) = BT(rr - RBR) + mv.x;
                if (rr + 1 < RBR) breg[rr + 1 < RBR ? rr + 1 : 0] += mv.y;
                else BT(rr + 1 - RBR) = BT(rr + 1 - RBR) + mv.y;
                fe0 = fma(a[rr], mv.x, fe0);
                fe1 = fma(a[rr + 1], mv.y, fe1);
                EIG_FIX_FENCE(rr);
              }
            }
            if (bndL && act) {                                     // slot 64: new lower column = (Q_63 +) coefL P_64
              const double2* mb = reinterpret_cast<const double2*>(bbQ + part * RP);
#pragma unroll
              for (int rr = 0; rr < RP; rr += 2) {
                const double2 mv = mb[rr / 2];
                a[rr] += mv.x;
                a[rr + 1] += mv.y;
                const double q0 = rr < RBR ? breg[rr < RBR ? rr : 0] : BT(rr - RBR);
                const double q1 = rr + 1 < RBR ? breg[rr + 1 < RBR ? rr + 1 : 0] : BT(rr + 1 - RBR);
                fe0 = fma(mv.x, q0, fe0);
                fe1 = fma(mv.y, q1, fe1);
                EIG_FIX_FENCE(rr);
              }
            }
          }
          pe[part * NS + slot] = fe0 + fe1;
          ++pairs;
          quiet = (anyf & 1) ? 0 : quiet + 1;
          quiet2 = (anyf & 2) ? 0 : quiet2 + 1;
          done = quiet >= S + 1 || quiet2 >= S + 1;
        }
      }
    }
    __syncthreads();
    // ---- store: columns back in place (lambda_j v_j, permuted order), scales folded in
    {
      const int ca = 2 * slot, cb = 2 * slot + 1;
      const double isA = st[4 * slot + 2], isB = st[4 * slot + 3];
#pragma unroll
      for (int rr = 0; rr < RP; ++rr) {
        const int row = row0 + rr;
        const double vb = (rr < RBR ? breg[rr < RBR ? rr : 0] : BT(rr - RBR)) * isB;
        if (row < m) {                                     // all ncol columns: the zero column that pads an odd order
          if (ca < ncol) G[(size_t)ca * ldg + row] = a[rr] * isA;   // may sit anywhere among them after the swaps (the slab
          if (cb < ncol) G[(size_t)cb * ldg + row] = vb;            // has room for m + 1 columns)
        }
      }
    }
    if (tid == 0) {
      E.info[2 * it] = (pairs + S - 1) / S;               // sweeps
      E.info[2 * it + 1] = done ? 1 : 0;                  // converged
    }
    (void)pt;
    __syncthreads();
  }
  EP_FLUSH();
}

template <int NP, int RP, int RBR, int SB>
static hipError_t launch_eig_one(const EigArgs& e, int grid, hipStream_t st) {
  const size_t lds = eig_wg_lds_bytes(NP, RP, RBR, SB);
  auto kern = letkf_eig_wg_kernel<NP, RP, RBR, SB>;
  if (lds > 48 * 1024) {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds);
    if (err != hipSuccess) return err;
  }
#ifdef LETKF_WAVE_PROF
  unsigned long long z[8] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_eig_prof), z, sizeof z);
#endif
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NP * SB), lds, st, e);
#ifdef LETKF_WAVE_PROF
  (void)hipStreamSynchronize(st);
  (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(g_eig_prof), sizeof z);
  fprintf(stderr, "EIG_PROF <%d,%d,%d,%d> grid=%d wave-cycles total=%llu barrier1=%.1f%% barrier2=%.1f%% barrier3=%.1f%%\n", NP, RP, RBR, SB,
          grid, z[4], 100.0 * z[0] / (z[4] + 1.0), 100.0 * z[1] / (z[4] + 1.0), 100.0 * z[2] / (z[4] + 1.0));
#endif
  return hipGetLastError();
}

#ifndef EIG_RBR2
#define EIG_RBR2 50
#endif
int eig_wg_max_order() { return 208; }

// mcap: upper bound of the matrix orders in this batch (the instantiation is chosen once per launch; points whose
// order does not fit skip themselves and are left to the block Jacobi)
hipError_t launch_eig_wg(const EigArgs& e, int mcap, int num_cu, hipStream_t st) {
  const int grid = (int)(e.npts < 4L * num_cu ? (e.npts > 0 ? e.npts : 1) : 4L * num_cu);
  if (mcap <= 128) return launch_eig_one<4, 32, 32, 1>(e, grid, st);
  // (an <8, 26, RBR, 2> shape -- 16 waves, a quarter of the LDS rows per lane -- was tried for this range: 128 VGPRs per
  // lane do not hold the two columns and the prefetch rings, 130-200 B/lane of scratch, C3-slab 171 ms against 107 ms)
#ifdef EIG_SHAPE6
  return launch_eig_one<6, 35, EIG_SHAPE6, 2>(e, grid, st);   // 12 waves = 3 per SIMD (168 VGPRs), EIG_SHAPE6 of the 35 rows of B in registers
#else
  return launch_eig_one<4, 52, EIG_RBR2, 2>(e, grid, st);
#endif
}

}  // namespace letkf
