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

#ifndef EIG_RBR2
#define EIG_RBR2 16
#endif
#ifndef EIG_PF
#define EIG_PF 4
#endif
#ifndef EIG_PM
#define EIG_PM 4
#endif

namespace letkf {

namespace {

using jacobi_dev::dpp_shift0;
using jacobi_dev::fast_rcp1;
using jacobi_dev::fast_rsqrt;
using jacobi_dev::fast_rsqrt1;

struct Rot {
  double c, s, tg;   // cosine, sine, t * gamma
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
  r.c = fast_rsqrt(w);
  r.s = r.c * tt;
  r.tg = tt * ga;
  return r;
}

// hipcc hoists every LDS read of an unrolled row loop to its top (17 doubles in flight, 34 VGPRs -- with 128 VGPRs per
// lane that spills the columns themselves); a compiler-level fence every 8 rows keeps the loads near their use (8 in flight)
#define EIG_ROW_FENCE(rr) \
  if (((rr) & 7) == 7) asm volatile("" ::: "memory")

// The two row updates, written as instruction sequences that work IN PLACE.  From C++ hipcc puts every new element of
// the lower column into a fresh register (the old one is still an operand of the upper column's update), i.e. it
// renames the whole column per step, and the copies / live ranges that follow cost ~90 spilled registers at 52 rows
// per lane (found with -Rpass-analysis and in the ISA).
//   even step:  a <- c b + s a ;  b <- c a - s b
__device__ __forceinline__ void rot_inplace(double& a, double& b, const double c, const double s) {
  double t;
  asm("v_mov_b64 %2, %0\n\t"
      "v_mul_f64 %0, %4, %0\n\t"
      "v_fmac_f64 %0, %3, %1\n\t"
      "v_mul_f64 %1, -%4, %1\n\t"
      "v_fmac_f64 %1, %3, %2"
      : "+v"(a), "+v"(b), "=&v"(t)
      : "v"(c), "v"(s));
}
//   odd step:   x <- f x + g y
__device__ __forceinline__ void axpby_inplace(double& x, const double f, const double g, const double y) {
  asm("v_mul_f64 %0, %1, %0\n\t"
      "v_fmac_f64 %0, %2, %3"
      : "+v"(x)
      : "v"(f), "v"(g), "v"(y));
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
constexpr int kPM = EIG_PM;   // mailbox rows in flight

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
  return 8 * ((size_t)(RP - RBR) * NT + 2 * (size_t)NP * NS + 2 * NS + 2 * (size_t)NP * RP) + 2 * 16 * sizeof(int) + 64;
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
  double* pn = po + kParts * NS;           // [NS][2]     squared norms (lower, upper) of every slot after its even step
  double* bbP = pn + 2 * NS;               // [8][RP]     mailbox: lower column of slot 64 (for slot 63)
  double* bbQ = bbP + kParts * RP;         // [8][RP]     mailbox: upper column of slot 63 (for slot 64)
  int* flags = reinterpret_cast<int*>(bbQ + kParts * RP);   // [2][16] convergence votes of the waves

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: the branches on part / sb below are wave-uniform)
  const int part = wave / SB, sb = wave % SB;
  const int slot = sb * 64 + lane;
  const int tidm1 = tid > 0 ? tid - 1 : 0;
  const bool bndR = SB == 2 && sb == 0 && lane == 63;   // slot 63: its right neighbour sits in the other slot-block
  const bool bndL = SB == 2 && sb == 1 && lane == 0;    // slot 64
  const int row0 = part * RP;

  EP_DECL();
  for (long it = blockIdx.x; it < E.npts; it += gridDim.x) {
    const long pt = E.pt0 + it;
    const int m = E.meta[2 * it + 1];
    const int solver = E.meta[2 * it] >> 8;              // 1: this kernel
    if (solver != 1 || m < 2 || m > MMAX) continue;     // (uniform for the workgroup)
    const int ldg = m | 1;
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
        else blds[(size_t)(rr - RBR) * NT + tid] = vbz;
        EIG_ROW_FENCE(rr);
      }
    }
    double alA = 0.0, alB = 0.0;
    int quiet = 0, quiet2 = 0, pairs = 0, sweep = 0;
    bool done = false, odd_notconv_prev = false, odd_notconv2_prev = false;
    int vph = 0;
    __syncthreads();

    for (; sweep < E.max_sweep && !done; ++sweep) {
      // ---- refresh the squared norms (they are carried by the rotation identities inside a sweep)
      {
        double sa = 0.0, sbq = 0.0, sab = 0.0;
#pragma unroll
        for (int rr = 0; rr < RP; ++rr) {
          const double vb = rr < RBR ? breg[rr < RBR ? rr : 0] : blds[(size_t)(rr - RBR) * NT + tid];
          sa = fma(a[rr], a[rr], sa);
          sbq = fma(vb, vb, sbq);
          sab = fma(a[rr], vb, sab);
          EIG_ROW_FENCE(rr);
        }
        pe[part * NS + slot] = sa;
        po[part * NS + slot] = sbq;
        __syncthreads();
        alA = sum_parts<NP>(pe + slot, NS);
        alB = sum_parts<NP>(po + slot, NS);
        __syncthreads();
        pe[part * NS + slot] = sab;                       // the first even step's inner product (see the odd step)
      }
      for (int t = 0; t < ncol && !done; t += 2) {
        // ================= even step: the slot's own two columns.  Their inner product was accumulated while the
        // previous odd step (or the norm refresh) had both columns in its hands: no pass of its own
        EP_SYNC(0);                                       // (1)
        bool notconv, notconv2;
        {
          const double ga = sum_parts<NP>(pe + slot, NS);
          const Rot r = make_rot(alA, alB, ga, true);
          notconv = r.notconv;
          notconv2 = r.notconv2;
          // rotate and swap: position 2s takes c B + s A, position 2s+1 takes c A - s B
          const double nA = alB + r.tg, nB = alA - r.tg;
          alA = nA;
          alB = nB;
          // LDS rows: the loads run kPF rows ahead of their use, written out in the source -- hipcc cannot move a row's
          // read above the previous row's write (every row has its own address register: no alias information), and
          // left alone it emits read, s_waitcnt lgkmcnt(0), use for every single row
          double pf[kPF];
#pragma unroll
          for (int u = 0; u < kPF; ++u)
            if (RBR + u < RP) pf[u] = blds[(size_t)u * NT + tid];
#pragma unroll
          for (int rr = 0; rr < RP; ++rr) {
            if (rr < RBR) {
              rot_inplace(a[rr], breg[rr < RBR ? rr : 0], r.c, r.s);
            } else {
              double vb = pf[(rr - RBR) % kPF];
              if (rr + kPF < RP) pf[(rr - RBR) % kPF] = blds[(size_t)(rr + kPF - RBR) * NT + tid];
              rot_inplace(a[rr], vb, r.c, r.s);
              blds[(size_t)(rr - RBR) * NT + tid] = vb;
            }
          }
          if (part == 0) {
            pn[2 * slot] = alA;
            pn[2 * slot + 1] = alB;
          }
        }
        if constexpr (SB == 2) {
          // mailbox for the slot pair (63, 64) that straddles the two slot-blocks
          if (bndL) {
#pragma unroll
            for (int rr = 0; rr < RP; ++rr) bbP[part * RP + rr] = a[rr];
          }
          if (bndR) {
#pragma unroll
            for (int rr = 0; rr < RP; ++rr)
              bbQ[part * RP + rr] = rr < RBR ? breg[rr < RBR ? rr : 0] : blds[(size_t)(rr - RBR) * NT + tid];
          }
          EP_SYNC(1);                                     // (2)
        }
        // ================= odd step: upper column of slot s (Q_s) with the lower column of slot s+1 (P_{s+1}).
        // The row loops below carry no branch, so that hipcc can issue the LDS reads of a fence group together (with a
        // conditional mailbox read per row every LDS read was followed by its own s_waitcnt lgkmcnt(0): ~260 exposed
        // LDS latencies per step pair, 2/3 of the kernel's time).  The slot pair that straddles the slot-blocks gets
        // its partner through an UNCONDITIONAL term: every wave reads its mailbox row (one address for the whole wave)
        // and multiplies it by a coefficient that is zero in all lanes but the one concerned.
        const double* mbx = (SB == 2 && sb == 1) ? bbQ + part * RP : bbP + part * RP;   // (wave-uniform)
        const double selP = bndR ? 1.0 : 0.0;
        {
          double p0 = 0.0, p1 = 0.0;
          double pf[kPF], pm[kPM];
#pragma unroll
          for (int u = 0; u < kPF; ++u)
            if (RBR + u < RP) pf[u] = blds[(size_t)u * NT + tid];
          if constexpr (SB == 2) {
#pragma unroll
            for (int u = 0; u < kPM; ++u)
              if (u < RP) pm[u] = mbx[u];
          }
#pragma unroll
          for (int rr = 0; rr < RP; ++rr) {
            // (register-only fences: without them hipcc fetches ALL rows of the neighbour first -- a third column in
            // registers; volatile asm statements keep their order, so each row's fetch waits for the previous row's FMA)
            asm volatile("" : "+v"(a[rr]));
            double pr = dpp_shift0<0x130>(a[rr]);          // lane + 1 (0 for lane 63)
            if constexpr (SB == 2) {                        // slot 63: its partner's column is in the mailbox
              pr = fma(selP, pm[rr % kPM], pr);
              if (rr + kPM < RP) pm[rr % kPM] = mbx[rr + kPM];
            }
            double q;
            if (rr < RBR) {
              q = breg[rr < RBR ? rr : 0];
            } else {
              q = pf[(rr - RBR) % kPF];
              if (rr + kPF < RP) pf[(rr - RBR) % kPF] = blds[(size_t)(rr + kPF - RBR) * NT + tid];
            }
            if (rr & 1) {
              p1 = fma(q, pr, p1);
              asm volatile("" : "+v"(p1));
            } else {
              p0 = fma(q, pr, p0);
              asm volatile("" : "+v"(p0));
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
          const double gR = sum_parts<NP>(po + slot, NS);
          const double gL = sum_parts<NP>(po + (slot > 0 ? slot - 1 : 0), NS);
          const double nPr = pn[2 * (slot + 1 < NS ? slot + 1 : slot)];       // |P_{s+1}|^2
          const double nQl = pn[2 * (slot > 0 ? slot - 1 : 0) + 1];           // |Q_{s-1}|^2
          const Rot rR = make_rot(alB, nPr, gR, hasR);     // lower = Q_s, upper = P_{s+1}
          const Rot rL = make_rot(nQl, alA, gL, hasL);     // lower = Q_{s-1}, upper = P_s
          odd_notconv_prev = rR.notconv;
          odd_notconv2_prev = rR.notconv2;
          // new upper column of slot s (position 2s+1) = c P_{s+1} + s Q_s ; new lower column (position 2s) = c Q_{s-1} - s P_s
          const double cp = hasR ? rR.c : 0.0, cq = hasR ? rR.s : 1.0;
          const double cql = hasL ? rL.c : 0.0, ca = hasL ? -rL.s : 1.0;
          if (hasR) alB = nPr + rR.tg;
          if (hasL) alA = nQl - rL.tg;
          int anyf = 0;
#pragma unroll
          for (int w = 0; w < NP * SB; ++w) anyf |= flags[16 * vph + w];
          vph ^= 1;
          double fe0 = 0.0, fe1 = 0.0;
          double pfq[kPF], pfl[kPF], pm[kPM];
#pragma unroll
          for (int u = 0; u < kPF; ++u)
            if (RBR + u < RP) {
              pfq[u] = blds[(size_t)u * NT + tid];
              pfl[u] = blds[(size_t)u * NT + tidm1];
            }
          if constexpr (SB == 2) {
#pragma unroll
            for (int u = 0; u < kPM; ++u)
              if (u < RP) pm[u] = mbx[u];
          }
#pragma unroll
          for (int rr = 0; rr < RP; ++rr) {
            // the neighbour's rows are fetched a second time (once for the inner product, once here); laundering a[rr]
            // keeps hipcc from merging the two fetches across the barrier, and orders this row behind the previous one
            asm volatile("" : "+v"(a[rr]));
            double pr = dpp_shift0<0x130>(a[rr]);
            double q, ql;
            if (rr < RBR) {
              q = breg[rr < RBR ? rr : 0];
              ql = dpp_shift0<0x138>(q);                   // lane - 1 (0 for lane 0)
            } else {
              q = pfq[(rr - RBR) % kPF];
              ql = pfl[(rr - RBR) % kPF];
              if (rr + kPF < RP) {
                pfq[(rr - RBR) % kPF] = blds[(size_t)(rr + kPF - RBR) * NT + tid];
                pfl[(rr - RBR) % kPF] = blds[(size_t)(rr + kPF - RBR) * NT + tidm1];
              }
            }
            if constexpr (SB == 2) {
              const double mb = pm[rr % kPM];
              if (rr + kPM < RP) pm[rr % kPM] = mbx[rr + kPM];
              pr = fma(selP, mb, pr);                      // slot 63 <- lower column of slot 64
              ql = bndL ? mb : ql;                         // slot 64 <- upper column of slot 63
            }
            if (rr < RBR) {
              axpby_inplace(breg[rr < RBR ? rr : 0], cq, cp, pr);
            } else {
              axpby_inplace(q, cq, cp, pr);
              blds[(size_t)(rr - RBR) * NT + tid] = q;
            }
            axpby_inplace(a[rr], ca, cql, ql);
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
          pe[part * NS + slot] = fe0 + fe1;
          ++pairs;
          quiet = (anyf & 1) ? 0 : quiet + 1;
          quiet2 = (anyf & 2) ? 0 : quiet2 + 1;
          done = quiet >= S + 1 || quiet2 >= S + 1;
        }
        if constexpr (SB == 2) {
          // the LDS rows of slot 63's upper column were just rewritten by its own wave while slot 64 (other wave) read
          // the mailbox copy: nothing to wait for.  But the rows of the left neighbour read at thread - 1 cross the wave
          // boundary only for slot 64, which uses the mailbox instead.
        }
      }
    }
    __syncthreads();
    // ---- store: columns back in place (lambda_j v_j, permuted order)
    {
      const int ca = 2 * slot, cb = 2 * slot + 1;
#pragma unroll
      for (int rr = 0; rr < RP; ++rr) {
        const int row = row0 + rr;
        const double vb = rr < RBR ? breg[rr < RBR ? rr : 0] : blds[(size_t)(rr - RBR) * NT + tid];
        if (row < m) {                                     // all ncol columns: the zero column that pads an odd order
          if (ca < ncol) G[(size_t)ca * ldg + row] = a[rr];   // may sit anywhere among them after the swaps (the slab
          if (cb < ncol) G[(size_t)cb * ldg + row] = vb;      // has room for m + 1 columns)
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
#define EIG_RBR2 16
#endif
int eig_wg_max_order() { return 208; }

// mcap: upper bound of the matrix orders in this batch (the instantiation is chosen once per launch; points whose
// order does not fit skip themselves and are left to the block Jacobi)
hipError_t launch_eig_wg(const EigArgs& e, int mcap, int num_cu, hipStream_t st) {
  const int grid = (int)(e.npts < 4L * num_cu ? (e.npts > 0 ? e.npts : 1) : 4L * num_cu);
  if (mcap <= 128) return launch_eig_one<4, 32, 32, 1>(e, grid, st);
  // (an <8, 26, RBR, 2> shape -- 16 waves, a quarter of the LDS rows per lane -- was tried for this range: 128 VGPRs per
  // lane do not hold the two columns and the prefetch rings, 130-200 B/lane of scratch, C3-slab 171 ms against 107 ms)
  return launch_eig_one<4, 52, EIG_RBR2, 2>(e, grid, st);
}

}  // namespace letkf
