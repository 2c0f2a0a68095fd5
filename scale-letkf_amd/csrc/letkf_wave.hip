// letkf_wave.hip -- wavefront-per-grid-point LETKF kernel for k <= 62, two wavefronts for 65 <= k <= 100 (gfx950).
//
// One 64-lane wavefront solves one grid point; the 4 waves of a workgroup are fully independent (no workgroup
// barrier anywhere), so a CU keeps 8 points in flight.  Each wave walks a run of consecutive points.
//
// Per point:
//   Gram         A = Ys^T Ys + (k-1)/rho I (common/common_letkf.f90:127-143) on the FP64 matrix cores, observation
//                rows straight from the obs table into MFMA operand layout; accumulator tiles -> lane j = column j
//   warm start   G0 = A Q, Q = eigenvectors of the previous point of the run (global workspace slot)
//   eigen-solve  (reference: common_mtx.f90:41 -> EISPACK rs, netlib.f:524) one-sided (Hestenes) Jacobi in registers,
//                row-split layout, odd-even transposition ordering, DPP exchanges only (jacobi_split).  Columns
//                converge to lambda_j v_j, so lambda_j = |g_j| and V needs no accumulation.
//   apply        (lane j holds v_j)
//                U[j][b] = v_j . B_b        B = [Ys^T d, Ys^T d_det, x'_1 .. x'_nv]  (LDS broadcast reads)
//                Out = V (D U)              through 8-column LDS transposition chunks, lane m gets row m
//                -> w-bar, w-bar_det, T x'_v;  RTPP/RTPS, beta, det member, q clamp as
//                   scale/letkf/letkf_tools.f90:457-513.  T / Pa themselves are only formed on request.

#include <hip/hip_runtime.h>
#include <vector>
#include <stdint.h>

#include <cstdlib>
#include <type_traits>

#include "letkf_device.h"
#include "letkf_search_dev.h"
#include "letkf_jacobi_dev.h"

namespace letkf {

namespace {

using namespace jacobi_dev;


__device__ __forceinline__ double wshfl_xor(double v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ double wshfl(double v, int src) { return __shfl(v, src, 64); }

// DPP cross-lane move of a double (two 32-bit VALU movs, no LDS crossbar).  CTRL is a gfx9 dpp_ctrl code:
// 0xB1 quad_perm[1,0,3,2] (lane^1), 0x4E quad_perm[2,3,0,1] (lane^2), 0x1B quad_perm[3,2,1,0] (lane^3),
// 0x141 row_half_mirror (lane^7), 0x140 row_mirror (lane^15).
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  // quad_perm / mirror patterns have a valid source in every lane: no `old` value is needed, and asking for one
  // costs an extra v_mov_b32 per dword (seen in the ISA: 106 of the 264 instructions of an even Jacobi step)
  if constexpr (CTRL < 0x130) {
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  } else {
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
  }
  return __hiloint2double(hi, lo);
}

// Reductions over the 64 lanes (every lane active).  (r4) The xor butterfly through __shfl_xor compiles to two ds_bpermute per
// stage -- six dependent LDS round trips per sum, and the analysis members take 11 .. 33 sums per point (found in the ISA of
// letkf_trio.hip: 636 ds_bpermute; the phase was 14 % of its wave time).  Same tree without LDS: inside a row of 16 lanes by DPP
// (lane ^ 1, lane ^ 2, then the mirrors: lanes of a quad / of eight hold the same partial sum by then, so lane ^ 7 and lane ^ 15
// deliver what lane ^ 4 and lane ^ 8 would), the four rows by v_readlane -- ((r0 + r1) + (r2 + r3)), the butterfly's own
// association: bitwise the same result.
#ifndef LETKF_WAVE_SUM_SHFL
#define LETKF_WAVE_SUM_SHFL 0
#endif
__device__ __forceinline__ double readlane_d(double v, int src);
__device__ __forceinline__ double wave_sum(double v) {
  if constexpr (LETKF_WAVE_SUM_SHFL) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += wshfl_xor(v, m);
    return v;
  } else {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
  }
}
__device__ __forceinline__ double wave_max(double v) {
  if constexpr (LETKF_WAVE_SUM_SHFL) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = fmax(v, wshfl_xor(v, m));
    return v;
  } else {
    v = fmax(v, dpp_mov<0xB1>(v));
    v = fmax(v, dpp_mov<0x4E>(v));
    v = fmax(v, dpp_mov<0x141>(v));
    v = fmax(v, dpp_mov<0x140>(v));
    return fmax(fmax(readlane_d(v, 0), readlane_d(v, 16)), fmax(readlane_d(v, 32), readlane_d(v, 48)));
  }
}
__device__ __forceinline__ double wave_min(double v) {
  if constexpr (LETKF_WAVE_SUM_SHFL) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = fmin(v, wshfl_xor(v, m));
    return v;
  } else {
    v = fmin(v, dpp_mov<0xB1>(v));
    v = fmin(v, dpp_mov<0x4E>(v));
    v = fmin(v, dpp_mov<0x141>(v));
    v = fmin(v, dpp_mov<0x140>(v));
    return fmin(fmin(readlane_d(v, 0), readlane_d(v, 16)), fmin(readlane_d(v, 32), readlane_d(v, 48)));
  }
}

// a value known to be identical in every lane -> SGPR pair (frees VGPRs, lets FMAs take a scalar operand)
__device__ __forceinline__ double uniform(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// lane `src` (wave-uniform) of v -> SGPR pair
__device__ __forceinline__ double readlane_d(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// Scheduling pin: makes every accumulator an in/out operand of an empty asm with a memory clobber.  The FMAs that
// produce the accumulators must then retire before it and the next LDS loads issue after it, which stops the
// compiler from issuing all unrolled broadcast loads first and spilling them (measured: 5 KB of scratch per lane).
template <int NB>
__device__ __forceinline__ void pin_acc(double (&c)[NB]) {
  if constexpr (NB == 2) {
    asm volatile("" : "+v"(c[0]), "+v"(c[1])::"memory");
  } else if constexpr (NB == 13) {
    asm volatile(""
                 : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]),
                   "+v"(c[8]), "+v"(c[9]), "+v"(c[10]), "+v"(c[11]), "+v"(c[12])::"memory");
  } else {
#pragma unroll
    for (int b = 0; b < NB; ++b) asm volatile("" : "+v"(c[b])::"memory");
  }
}

// sum / max / min over all lanes of the point.  With two waves the partials meet in a 4-double LDS scratch; `slot`
// alternates between consecutive calls (each call has its own barrier, so slot s is free again two calls later).
template <int NW, int OP>
__device__ __forceinline__ double preduce(double v, double* red, int& slot) {
  v = (OP == 0) ? wave_sum(v) : (OP == 1) ? wave_max(v) : wave_min(v);
  if constexpr (NW == 1) {
    return v;
  } else {
    if ((threadIdx.x & 63) == 0) red[2 * slot + (threadIdx.x >> 6)] = v;
    __syncthreads();
    const double a = red[2 * slot], b = red[2 * slot + 1];
    slot ^= 1;
    return (OP == 0) ? a + b : (OP == 1) ? fmax(a, b) : fmin(a, b);
  }
}

__device__ __forceinline__ long xcd_remap_w(long orig, long n) {
  const long q = n >> 3, r = n & 7;
  const long xcd = orig & 7, j = orig >> 3;
  const long base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + j;
}

// Dynamic run scheduling (PointArgs::sched, PointArgs::plan).  The runs are handed out in UNITS of `ub` consecutive run
// ids (1 for runs of 16 points or more; short runs are bundled so that a draw covers ~16 points), and the units [0, n) are
// cut into 8 contiguous ranges, one per XCD -- the same ranges as the static dealing of round 1, so that neighbouring
// columns still meet in one L2.  Every wave (two-wave points: every workgroup) starts with a unit that is its own by its
// position in the grid -- no 2048 waves queueing at 8 counters when the kernel starts -- and then draws the next one from
// its XCD's counter (a device-scope load, so that a finished range costs no read-modify-write, then one atomicAdd by one
// lane); when that range is used up it takes units from the range that has most left.  A domain whose observations sit
// in one place (a radar disc: the columns outside have no observation and cost a hundredth of a column inside) left
// whole XCDs idle under the static dealing (C2-disc: 259 -> 147 ms).
// f units of every range (ub = 1 only) are handed out last and in quarters, so that the waves do not end a whole run (C2:
// a column of 60 points, 14 ms) apart: every t-th one, a sample spread over the range, so that the quarters carry the
// range's average work wherever its observations sit.  Which points start a quarter -- i.e. start cold -- is a fixed
// function of the launch shape, so results stay bitwise reproducible from run to run.
// A drawn unit is coded as 8 * unit id + (0: whole, 4 + s: quarter s of its run), -1: nothing left.
// Everything here is wave-uniform and written so that it stays in scalar registers (the plan comes from the kernel
// arguments; the one division is a multiplication by a host-made reciprocal): the first version did this arithmetic in
// lane 0's vector registers, and the vector registers it needed around every draw cost the kernel 250 B/lane of scratch
// and 50 GB of spill traffic per C2 launch.
__host__ __device__ __forceinline__ int sched_unit(const SchedPlan& P, const int x, const int i) {
  const int whole = P.whole[x], f = P.f[x], t = P.t[x], base = P.base[x];
  if (i < whole) {
    const int head = f * (t - 1);
    if (i < head) {
      const int g = (int)(((unsigned long long)(unsigned)i * P.magic[x]) >> 40);   // i / (t - 1)
      return 8 * (base + g * t + (i - g * (t - 1)));
    }
    return 8 * (base + f * t + (i - head));
  }
  return 8 * (base + ((i - whole) >> 2) * t + (t - 1)) + 4 + ((i - whole) & 3);
}
// the same word in every lane, as a scalar
__device__ __forceinline__ int sched_peek(const unsigned* c) {
  return __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ int sched_take(unsigned* c) {
  int i = 0;
  if ((threadIdx.x & 63) == 0) i = (int)atomicAdd(c, 1u);
  return __builtin_amdgcn_readfirstlane(i);
}
// called by a whole wave; slot = the wave's (workgroup's) position among those of its XCD, or -1 after its first unit
__device__ __forceinline__ int sched_next(const SchedPlan& P, unsigned* cnt, const int xcd, const int slot) {
  if (slot >= 0 && slot < P.nstat[xcd]) return sched_unit(P, xcd, slot);
  {
    const int dyn = P.whole[xcd] + 4 * P.f[xcd] - P.nstat[xcd];
    if (dyn > 0 && sched_peek(&cnt[16 * xcd]) < dyn) {
      const int i = sched_take(&cnt[16 * xcd]);
      if (i < dyn) return sched_unit(P, xcd, P.nstat[xcd] + i);
    }
  }
  // own range used up: help where most is left (so that all ranges end together, each with its quartered runs last)
#pragma unroll 1
  for (int tries = 0; tries < 64; ++tries) {
    int best = -1, most = 0;
#pragma unroll 1
    for (int x = 0; x < 8; ++x) {
      const int left = P.whole[x] + 4 * P.f[x] - P.nstat[x] - sched_peek(&cnt[16 * x]);
      if (left > most) {
        most = left;
        best = x;
      }
    }
    if (best < 0) return -1;
    const int i = sched_take(&cnt[16 * best]);
    if (i < P.whole[best] + 4 * P.f[best] - P.nstat[best]) return sched_unit(P, best, P.nstat[best] + i);
  }
  return -1;
}

// host: the plan for a grid of `grid` workgroups with ppw wave-slots each, `resident` wave-slots in flight per XCD
static void sched_make_plan(SchedPlan& P, const long npts, const long stride, const int run_len, const int grid, const int ppw,
                            const int resident_per_xcd, const int ub_of = 1) {
  const long S = stride > 1 ? stride : 1, rl = run_len > 1 ? run_len : 1;
  const long nruns = S * ((npts / S + rl - 1) / rl);
  // short runs are bundled (a draw should cover ~16 points), as long as that leaves every wave-slot of the grid four units
  int ub = rl >= 16 ? 1 : (int)((16 + rl - 1) / rl);
  const long most = nruns / (4L * grid * ppw);
  if (ub > most) ub = most < 1 ? 1 : (int)most;
  // ub_of > 1 (letkf_trio.hip: three runs are walked in step): whole multiples of it
  if (ub_of > 1) ub = (ub + ub_of - 1) / ub_of * ub_of;
  const long n = (nruns + ub - 1) / ub;
  // whole runs of 8 points or more may be quartered at the end of a range: as many as the XCD has wave-slots in flight
  const int fs = (ub == 1 && rl >= 8) ? ((long)grid * ppw / 8 < resident_per_xcd ? (int)((long)grid * ppw / 8) : resident_per_xcd) : 0;
  const int q = (int)(n >> 3), r = (int)(n & 7);
  for (int x = 0; x < 8; ++x) {
    const int len = q + (x < r ? 1 : 0);
    P.base[x] = x * q + (x < r ? x : r);
    P.f[x] = fs < len ? fs : len;
    P.whole[x] = len - P.f[x];
    P.t[x] = P.f[x] > 0 ? len / P.f[x] : 1;
    const long mine = (long)((grid - x + 7) >> 3) * ppw;              // wave-slots of the grid that sit on XCD x
    const long units = (long)P.whole[x] + 4L * P.f[x];
    P.nstat[x] = (int)(mine < units ? mine : units);
    P.magic[x] = P.t[x] > 1 ? ((1ull << 40) / (unsigned long long)(P.t[x] - 1)) + 1ull : 0ull;
  }
  P.ub = ub;
  P.nruns = nruns;
}

constexpr int kTnW = 8;               // obs rows per LDS tile (per wave)
constexpr int kChunk = 8;             // columns per LDS transposition chunk
constexpr int kVld = kChunk + 2;      // row stride of the transposition buffer (doubles, even)


// Out[b] (lane m: row m of V C) += sum over the wave's columns j of V[m][j] * C[j][b], b < NB.
// V[:, j] is lane j's register column vcol[], C[j][:] is lane j's crow[].  Done in chunks of
// kChunk columns through LDS:  vbuf[KR][kVld], cbuf[kChunk][NBP].
template <int KR, int NB, int NW>
__device__ __forceinline__ void rows_times_c(const double (&vcol)[KR], const double (&crow)[NB], double (&out)[NB],
                                             const int k, double* vbuf, double* cbuf) {
  constexpr int NBP = (NB + 1) & ~1;
  int lane = threadIdx.x & (64 * NW - 1);
  if constexpr (NW == 2) asm volatile("" : "+v"(lane));   // (two-wave points: see the point loop of letkf_wave_kernel)
#pragma unroll
  for (int b = 0; b < NB; ++b) out[b] = 0.0;
  const int ncol = (k + 1) & ~1;                 // columns live in lanes [0, ncol) (see jacobi_split)
  for (int j0 = 0; j0 < ncol; j0 += kChunk) {
    psync<NW>();
    if (lane >= j0 && lane < j0 + kChunk) {
      const int jj = lane - j0;
#pragma unroll
      for (int r = 0; r < KR; ++r) vbuf[r * kVld + jj] = vcol[r];
#pragma unroll
      for (int b = 0; b < NB; ++b) cbuf[jj * NBP + b] = crow[b];
      if (NBP > NB) cbuf[jj * NBP + NB] = 0.0;
    }
    psync<NW>();
    const int mrow = lane < KR ? lane : KR - 1;
    double vv[kChunk];
#pragma unroll
    for (int jj = 0; jj < kChunk; jj += 2) {
      const double2 t2 = *reinterpret_cast<const double2*>(&vbuf[mrow * kVld + jj]);
      vv[jj] = t2.x;
      vv[jj + 1] = t2.y;
    }
    const int nj = min(kChunk, ncol - j0);
#pragma unroll
    for (int jj = 0; jj < kChunk; ++jj) {
      if (jj < nj) {
#pragma unroll
        for (int b = 0; b < NBP; b += 2) {
          const double2 c2 = *reinterpret_cast<const double2*>(&cbuf[jj * NBP + b]);   // wave-uniform: broadcast
          out[b] = fma(vv[jj], c2.x, out[b]);
          if (b + 1 < NB) out[b + 1] = fma(vv[jj], c2.y, out[b + 1]);
        }
      }
      pin_acc<NB>(out);
    }
  }
  psync<NW>();
}

// ---------------------------------------------------------------------------------------------
// Warm start of the eigensolve: G0 = A Q with Q the eigenvector matrix of the PREVIOUS point of this wave's run.
// One-sided Jacobi on A Q (any orthogonal Q) still ends with columns lambda_j v_j of A -- the accumulated rotation
// is simply Q^T V -- but neighbouring grid points see almost the same observations, so Q nearly diagonalises A and
// the slow linear phase of the iteration (6 of the 9 sweeps at k = 50) is skipped: 9.1 -> 6.0 sweeps on the C2
// workload for x- or y-neighbours, no worse than the cold start for an unrelated Q.  Q's departure from
// orthogonality is the previous point's final residual (< 1e-10 measured BEFORE its last sweep rotated it away),
// so errors do not accumulate along a run.
//
// Lane j needs (A Q)[:, j] = sum_i A[:, i] Q[i][j]: its own Q column comes back from the wave's global workspace
// slot a chunk at a time (it cannot stay in registers: 256 VGPRs hold g, h and nothing else); the columns of A are
// written to LDS kWC at a time and read back as wave-uniform (broadcast) ds_read_b128.  2 k^2 FMAs + k^2/2 LDS reads
// per lane.  History: this LDS version was first measured at 1.6 sweeps' worth of time because the column-per-lane
// Jacobi's odd steps kept the LDS return path ~70 % busy; a v_readlane_b32 x2 + SGPR-operand FMA version
// (tools/ubench_readlane.hip: 5.6 ns per triple against 2.35 ns per bare v_fma_f64) cost one sweep; with the
// row-split Jacobi (no LDS in its steps) the LDS broadcast is the cheapest again.
// ---------------------------------------------------------------------------------------------
// early stop of the eigensolver (letkf_jacobi_dev.h): one-wave points, and two-wave points of the 64-column
// instantiation (both ballots ride one barrier: k = 64 1.00 M against 0.93 M solves/s).  Not the 80- and 100-column
// ones: at k = 100 the same change costs 8 % (A/B on one box) -- their register allocation is fragile, see the Gram.
#ifndef LETKF_INPLACE_NW
#define LETKF_INPLACE_NW(kr, nw) ((nw) == 2 && (kr) > 64)   // two half-column arrays instead of three (letkf_jacobi_dev.h)
#endif
#ifndef LETKF_TWO_PER_SIMD
#define LETKF_TWO_PER_SIMD(kr, nw) false   // (tried for the in-place instantiations: 256 registers in all, 2 KB/lane of scratch, k = 100 510 k -> 448 k)
#endif
#ifndef LETKF_EARLY_NW
#define LETKF_EARLY_NW(kr, nw) true
#endif
#ifndef LETKF_GRAM_DEPTH
#define LETKF_GRAM_DEPTH 3
#endif
constexpr int kGramDepth = LETKF_GRAM_DEPTH;   // 4-obs Gram steps in flight (measured on C2: 3 -> 477 ms, 4 -> 487, 5 -> 492)
constexpr int kWC = 4;   // columns of A per LDS chunk (small: the unrolled chunk body is ~100 instructions per column)
// pins out[R0 .. R0+7] (those below KR): see pin_acc
template <int KR, int R0>
__device__ __forceinline__ void pin_rows8(double (&o)[KR]) {
  if constexpr (R0 + 8 <= KR) {
    asm volatile(""
                 : "+v"(o[R0]), "+v"(o[R0 + 1]), "+v"(o[R0 + 2]), "+v"(o[R0 + 3]), "+v"(o[R0 + 4]), "+v"(o[R0 + 5]),
                   "+v"(o[R0 + 6]), "+v"(o[R0 + 7])::"memory");
  } else if constexpr (R0 < KR) {
#pragma unroll
    for (int r = R0; r < KR; r += 2) asm volatile("" : "+v"(o[r]), "+v"(o[r + 1])::"memory");
  }
}
template <int KR, int NW>
__device__ __forceinline__ void warm_start_product(double (&g)[KR], const double* __restrict__ uws, const int k,
                                                   double* cb) {
  constexpr int NL = 64 * NW;
  constexpr int NG = (KR + 7) / 8;
  static_assert(NG <= 13, "pin_rows8 dispatch below");
  int lane = threadIdx.x & (NL - 1);
  if constexpr (NW == 2) asm volatile("" : "+v"(lane));   // (two-wave points: see the point loop of letkf_wave_kernel)
  const int ncol = (k + 1) & ~1;
  double out[KR];
#pragma unroll
  for (int r = 0; r < KR; ++r) out[r] = 0.0;
  double un[kWC];
#pragma unroll
  for (int q = 0; q < kWC; ++q) un[q] = (q < ncol) ? uws[(size_t)q * NL] : 0.0;
#pragma unroll 1
  for (int i0 = 0; i0 < ncol; i0 += kWC) {
    double u[kWC];
#pragma unroll
    for (int q = 0; q < kWC; ++q) u[q] = un[q];
    psync<NW>();
    if (lane >= i0 && lane < i0 + kWC) {
      double* mine = cb + (lane - i0) * KR;
#pragma unroll
      for (int r = 0; r < KR; r += 2) *reinterpret_cast<double2*>(&mine[r]) = double2{g[r], g[r + 1]};
    }
    psync<NW>();
#pragma unroll
    for (int q = 0; q < kWC; ++q) un[q] = (i0 + kWC + q < ncol) ? uws[(size_t)(i0 + kWC + q) * NL] : 0.0;
    // 8-row groups, the loads of group g+1 issued before the FMAs of group g
    double2 cur[4], nxt[4];
    auto ld = [&](const int q, const int gi, double2 (&d)[4]) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (8 * gi + 2 * e < KR) d[e] = *reinterpret_cast<const double2*>(&cb[q * KR + 8 * gi + 2 * e]);   // broadcast
    };
    ld(0, 0, cur);
#pragma unroll
    for (int q = 0; q < kWC; ++q) {
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {
        const int nq = (gi + 1 < NG) ? q : q + 1, ng = (gi + 1 < NG) ? gi + 1 : 0;
        if (nq < kWC) ld(nq, ng, nxt);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 8 * gi + 2 * e;
          if (r < KR) {
            out[r] = fma(cur[e].x, u[q], out[r]);
            out[r + 1] = fma(cur[e].y, u[q], out[r + 1]);
          }
        }
        if (gi == 0) pin_rows8<KR, 0>(out);
        if (gi == 1) pin_rows8<KR, 8>(out);
        if (gi == 2) pin_rows8<KR, 16>(out);
        if (gi == 3) pin_rows8<KR, 24>(out);
        if (gi == 4) pin_rows8<KR, 32>(out);
        if (gi == 5) pin_rows8<KR, 40>(out);
        if (gi == 6) pin_rows8<KR, 48>(out);
        if (gi == 7) pin_rows8<KR, 56>(out);
        if (gi == 8) pin_rows8<KR, 64>(out);
        if (gi == 9) pin_rows8<KR, 72>(out);
        if (gi == 10) pin_rows8<KR, 80>(out);
        if (gi == 11) pin_rows8<KR, 88>(out);
        if (gi == 12) pin_rows8<KR, 96>(out);
#pragma unroll
        for (int e = 0; e < 4; ++e) cur[e] = nxt[e];
      }
    }
  }
  psync<NW>();
#pragma unroll
  for (int r = 0; r < KR; ++r) g[r] = out[r];
}

// The same product on the FP64 matrix cores, for one-wave points with KR <= 50 (A fits the wave's LDS slice whole:
// 4 x 20 KB per workgroup, two workgroups still share a CU -- tools/ubench_lds_occ.hip).  The LDS-broadcast version
// above is latency-bound (4 ds_read_b128 in flight against 8 FMAs: 23 cycles per instruction, 15 % of the kernel's
// wave time on C2, measured with the PROF build) and it competes with the other wave's Jacobi for the vector ALU;
// here every lane first parks its column of A in LDS -- which frees the 2 KR registers of g for the accumulators --
// and the product runs as KS = KR/4 steps of 16 v_mfma_f64_16x16x4 with
//   A operand, row block I : A[4c + I][4s + q]   = 4 consecutive doubles of LDS column 4s+q (A is symmetric)
//   B operand, col block J : Q[4s + q][4c + J]   = 4 consecutive doubles of workspace row 4s+q (coalesced 32-B loads)
// (lane = 16 q + c; the blocks interleave rows / columns with stride 4 instead of covering 16 contiguous ones, so that
// both operands of a step are one 32-byte read per lane).  Accumulator (I, J), register `reg` of lane (q, c) is
// G0[16 reg + 4q + I][4c + J]; the tiles go back to "lane j owns column j" through the same LDS region.
typedef __attribute__((address_space(1))) double gdouble;
typedef double v2d_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) v2d_t gdouble2;

template <int KR>
__device__ __forceinline__ void warm_start_product_mfma(double (&g)[KR], const double* __restrict__ qslot, double* cb) {
  static_assert(KR % 2 == 0 && KR <= 50, "A must fit the LDS slice");
  constexpr int KS = (KR + 3) / 4;             // contraction steps of 4
  constexpr int PD = (KS < 6) ? KS : 6;        // workspace row-quads in flight
  const int wlane = threadIdx.x & 63;
  const int q = wlane >> 4, c = wlane & 15;
  struct Quad {
    double2 lo, hi;
  };
  unsigned long long qaddr = reinterpret_cast<unsigned long long>(qslot + q * 64 + 4 * c);
  asm volatile("" : "+v"(qaddr));
  const gdouble2* qp = (const gdouble2*)qaddr;   // integer -> global pointer: no generic pointer in between
  auto ldq = [&](const int s) {
    Quad t{double2{0.0, 0.0}, double2{0.0, 0.0}};
    if (4 * s + 3 < KR || 4 * s + q < KR) {    // rows >= KR do not exist in the slot
      // (global address space spelled out: behind the laundering asm the pointer is generic, hipcc emits flat_load,
      // flat operations count in lgkmcnt as well and return out of order, so every LDS wait in this loop became
      // lgkmcnt(0) and drained the workspace loads in flight)
      const gdouble2* a = qp + (size_t)s * 128;
      const v2d_t lo = a[0], hi = a[1];
      t.lo = double2{lo.x, lo.y};
      t.hi = double2{hi.x, hi.y};
    }
    return t;
  };
  auto lda = [&](const int s) {
    Quad t{double2{0.0, 0.0}, double2{0.0, 0.0}};
    if (4 * s + 3 < KR || 4 * s + q < KR) {
      const double* a = cb + (4 * s + q) * KR + 4 * c;   // rows 4c+I >= KR read finite-or-not garbage: only output
      t.lo = *reinterpret_cast<const double2*>(a);        // rows >= KR see it, and those are dropped below
      t.hi = *reinterpret_cast<const double2*>(a + 2);
    }
    return t;
  };
  Quad qr[PD];
#pragma unroll
  for (int s = 0; s < PD; ++s) qr[s] = ldq(s);
  wave_lds_sync();
  if (wlane < KR) {
    double* mine = cb + wlane * KR;
#pragma unroll
    for (int r = 0; r < KR; r += 2) *reinterpret_cast<double2*>(&mine[r]) = double2{g[r], g[r + 1]};
  }
  wave_lds_sync();
  v4d acc[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
  Quad ar[2];
  ar[0] = lda(0);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    if (s + 1 < KS) ar[(s + 1) & 1] = lda(s + 1);
    const Quad a4 = ar[s & 1], b4 = qr[s % PD];
    const double av[4] = {a4.lo.x, a4.lo.y, a4.hi.x, a4.hi.y};
    const double bv[4] = {b4.lo.x, b4.lo.y, b4.hi.x, b4.hi.y};
#pragma unroll
    for (int I = 0; I < 4; ++I)
#pragma unroll
      for (int J = 0; J < 4; ++J) acc[4 * I + J] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[I], bv[J], acc[4 * I + J], 0, 0, 0);
    if (s + PD < KS) qr[s % PD] = ldq(s + PD);
  }
  wave_lds_sync();
#pragma unroll
  for (int J = 0; J < 4; ++J) {
    const int col = 4 * c + J;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int row0 = 16 * reg + 4 * q;
      if (col < KR && row0 < KR) *reinterpret_cast<double2*>(&cb[col * KR + row0]) = double2{acc[J][reg], acc[4 + J][reg]};
      if (col < KR && row0 + 2 < KR) *reinterpret_cast<double2*>(&cb[col * KR + row0 + 2]) = double2{acc[8 + J][reg], acc[12 + J][reg]};
    }
  }
  wave_lds_sync();
  {
    const double* mine = cb + (wlane < KR ? wlane : 0) * KR;
#pragma unroll
    for (int r = 0; r < KR; r += 2) {
      const double2 v2 = *reinterpret_cast<const double2*>(&mine[r]);
      g[r] = wlane < KR ? v2.x : 0.0;
      g[r + 1] = wlane < KR ? v2.y : 0.0;
    }
  }
  wave_lds_sync();
}

}  // namespace

#if !defined(LETKF_WAVE_UNIT2) && !defined(LETKF_WAVE_UNIT3)
// include/letkf_amd.h, letkf_sched_plan_check: every run exactly once (whole or as four quarters)?
int sched_plan_check(long npts, long stride, int run_len, int grid, int ppw, int resident_per_xcd, int ub_of) {
  if (npts < 0 || grid < 1 || ppw < 1 || run_len < 1 || ub_of < 1) return -1;
  SchedPlan P;
  sched_make_plan(P, npts, stride, run_len, grid, ppw, resident_per_xcd, ub_of);
  if (P.nruns > (1L << 27)) return -2;
  if (P.ub % ub_of != 0) return -9;                            // (letkf_trio.hip: units are whole multiples of three runs)
  std::vector<unsigned char> seen((size_t)P.nruns, 0);   // bit 7: whole, bits 0-3: quarters
  for (int x = 0; x < 8; ++x) {
    if (P.nstat[x] < 0 || P.nstat[x] > P.whole[x] + 4 * P.f[x]) return -3;
    for (int i = 0; i < P.whole[x] + 4 * P.f[x]; ++i) {
      const int code = sched_unit(P, x, i);
      const long u = code >> 3;
      for (long rid = u * P.ub; rid < (u + 1) * P.ub && rid < P.nruns; ++rid) {
        if (rid < 0) return -4;
        const unsigned char bit = (code & 4) ? (unsigned char)(1u << (code & 3)) : (unsigned char)0x80;
        if ((code & 4) && P.ub != 1) return -5;
        if (seen[(size_t)rid] & (bit | ((code & 4) ? 0x80 : 0x0f))) return -6;    // handed out twice
        seen[(size_t)rid] |= bit;
      }
      if (u * P.ub >= P.nruns) return -7;                                          // a unit beyond the last run
    }
  }
  for (long rid = 0; rid < P.nruns; ++rid)
    if (seen[(size_t)rid] != 0x80 && seen[(size_t)rid] != 0x0f) return -8;         // missed, or quartered in part
  return 0;
}
#endif   // LETKF_WAVE_UNIT2

// smallest k the instantiation <KR, NW> is dispatched for (launch_wave_kernel walks the instances in this order)
__host__ __device__ constexpr int wave_kmin(int KR, int NW) {
  return NW == 1 ? (KR == 16 ? 1 : KR == 20 ? 17 : KR == 32 ? 17 : KR == 48 ? 33 : KR == 50 ? 49 : KR == 64 ? 51 : 1)
                 : (KR == 64 ? 63 : KR == 80 ? 65 : KR == 100 ? 81 : 1);
}

// Small ensembles (KR <= 20: C1, the k = 20 workloads) are issue- and latency-bound with 20 of 64 lanes carrying a column (one wave
// per SIMD -> two: x 1.47 on C2-k20, measured with padded LDS), so their slice is cut to what such a point needs -- the Jacobi's
// conversion chunk is KR rows (not 24), the apply phase's output buffer 32 rows (not 64): 12.5 KB per wave, 50 KB per workgroup,
// room for THREE workgroups per CU.  Measured (r4, A/B in one gpurun call, -DLETKF_SMALL_OCC=3 against 2): the register budget
// of three waves per SIMD (168) costs this kernel 464 B/lane of scratch instead of 136 -- C2-k20 -6 %, C1 +4 %.  The default
// stays at two; the slice stays small.
__host__ __device__ constexpr bool wave_small(int KR, int NW) { return NW == 1 && KR <= 20; }
__host__ __device__ constexpr int wave_base_doubles(int KR, int NW) { return wave_small(KR, NW) ? 1280 : 1536 * NW; }
__host__ __device__ constexpr int wave_jacobi_rc(int KR, int NW) { return wave_small(KR, NW) ? KR : 24; }   // rows per conversion chunk
__host__ __device__ constexpr int wave_ob_rows(int KR) { return KR <= 20 ? 32 : 64; }                        // rows of the MAPPLY output buffer
#ifndef LETKF_SMALL_OCC
#define LETKF_SMALL_OCC 2
#endif
__host__ __device__ constexpr int wave_occupancy(int KR, int NW) { return NW == 1 ? (wave_small(KR, NW) ? LETKF_SMALL_OCC : 2) : 1; }

// per-wave LDS slice (doubles)
__host__ __device__ inline int wave_slice_doubles(int KR, int nv, int NW) {
  const int nb = nv + 2;
  int tile = kTnW * 64;                       // obs tile, also reused as vbuf (KR * kVld) and kk-output C chunk
  const int vb = KR * kVld;
  if (vb > tile) tile = vb;
  int bmat = ((nb + 1) & ~1) * KR;            // B vectors [KR][NBP]; reused as the T/Pa C chunk (kChunk * KR)
  if (kChunk * KR > bmat) bmat = kChunk * KR;
  const int cb = kChunk * ((nb + 1) & ~1);
  const int small = 3 * kTnW + 8 * nv + 16;
  if (tile + bmat < wave_base_doubles(KR, NW)) bmat = wave_base_doubles(KR, NW) - tile;   // Gram transposition buffer abuf[64 NW][18] and
                                                         // the Jacobi exchange slots (64 NW * RC doubles) span tile + bmat
  int tot = tile + bmat + cb + small + 8;                // + 4 doubles of reduction scratch (two-wave points)
  if (NW == 1 && KR <= 50 && tot < (KR - 1) * KR + 64) tot = (KR - 1) * KR + 64;   // A whole: warm_start_product_mfma
  if (NW == 1 && KR < 32 && tot < 32 * KR + 64 * ((KR + 3) / 4) + 128 + 16 * wave_ob_rows(KR)) tot = 32 * KR + 64 * ((KR + 3) / 4) + 128 + 16 * wave_ob_rows(KR);   // + [rows][16] output buffer
  if (NW == 1 && KR <= 50 && tot < 32 * KR + 64 * ((KR + 3) / 4) + 128)               // half of V + padded B + spectra: the apply phase on the matrix cores
    tot = 32 * KR + 64 * ((KR + 3) / 4) + 128;
  return (tot + 1) & ~1;
}

// KKOUT: also materialise T / Pa (fine boundary, parity, diagnostics) -- a separate instantiation so that the
// production kernel carries neither the code nor the registers for it.
// FUSED = 2: obs_local walked inside the kernel (mode 2); FUSED = 3: the vertical half of obs_local on the column's horizontal
// survivors (mode 3) -- instantiations of their own: carried by the list-driven kernel
// the extra code cost 15 % of its speed (registers / instruction cache), measured.
// Profiling build (make PROF=1): per-phase wave time from s_memtime, kept in SGPRs, summed over all waves.
#ifdef LETKF_WAVE_PROF
#define PROF_DECL unsigned long long prof_t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long prof_last = __builtin_amdgcn_s_memtime(); const unsigned long long prof_t0 = prof_last; int prof_units = 0;
#define PROF_MARK(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); prof_t[i] += t_ - prof_last; prof_last = t_; }
// [10]: ~earliest wave start, [11]: latest wave end (s_memtime), [12 + u]: waves that did u units (u capped at 11),
// [24]: units done in all (= the plan's units if every one was drawn exactly once)
#define PROF_FLUSH if (A.prof && wlane == 0) { for (int i_ = 0; i_ < 10; ++i_) atomicAdd(&A.prof[i_], prof_t[i_]); \
    atomicMax(&A.prof[10], ~prof_t0); atomicMax(&A.prof[11], (unsigned long long)__builtin_amdgcn_s_memtime()); \
    atomicAdd(&A.prof[12 + (prof_units < 11 ? prof_units : 11)], 1ull); atomicAdd(&A.prof[24], (unsigned long long)prof_units); }
#define PROF_UNIT ++prof_units;
#else
#define PROF_DECL
#ifdef LETKF_MARK_FENCE   // A/B knob (make VARIANT=...): compiler-level fences where the PROF twin reads its clock
#define PROF_MARK(i) { if ((LETKF_MARK_FENCE >> (i)) & 1) asm volatile("" ::: "memory"); }
#else
#define PROF_MARK(i)
#endif
#define PROF_FLUSH
#define PROF_UNIT
#endif

// acc += (ys of lane A of this lane's row of 16) * y: ONE instruction -- FP64 instructions take exactly one DPP control on
// gfx950, row_newbcast (tools/ubench_newbcast.hip: correct in every row, ~10 cycles of a SIMD per instruction beside the matrix
// instructions).  NOP: two wait states in front (a DPP read of a register a vector instruction has just written; the compiler's
// hazard recogniser does not see into inline asm) -- set on the first one of a group.
template <int A, bool NOP>
__device__ __forceinline__ void fmac_row_bcast(double& acc, const double ys, const double y) {
#define LETKF_FMAC_BCAST(N)                                                                                                      \
  if constexpr (A == N) {                                                                                                        \
    if constexpr (NOP) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(ys), "v"(y)); \
    else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(ys), "v"(y)); \
  }
  LETKF_FMAC_BCAST(0) LETKF_FMAC_BCAST(1) LETKF_FMAC_BCAST(2) LETKF_FMAC_BCAST(3) LETKF_FMAC_BCAST(4) LETKF_FMAC_BCAST(5)
#undef LETKF_FMAC_BCAST
}

// Checked build (make CHECKED=1 -> lib/libletkf_amd_checked.so, run by tests/test_gpu_checked.py): every index the column-survivor
// mode derives from device data -- the run's column, the column's survivor range, the wave's slot, each list entry -- is tested
// against the bound the HOST sized the buffers by before it is used; a violation is recorded (code, workgroup, value, bound) in
// PointArgs::prof and the access is left out, so that the entry returns an error instead of the process dying on a memory fault
// with nothing to tell which access it was.  Absent from the normal build.
#ifdef LETKF_CHECKED
#define LETKF_CHECK(ok, code, val, lim) letkf_check_fail(A.prof, (ok), (code), (long)(val), (long)(lim))
__device__ __forceinline__ bool letkf_check_fail(unsigned long long* rec, const bool ok, const int code, const long val, const long lim) {
  if (!ok && rec && atomicCAS(&rec[0], 0ull, (unsigned long long)code) == 0ull) {
    rec[1] = blockIdx.x;
    rec[2] = (unsigned long long)val;
    rec[3] = (unsigned long long)lim;
  }
  return ok;
}
#else
#define LETKF_CHECK(ok, code, val, lim) true
#endif

template <int KR, int NV, bool KKOUT, int NW, int FUSED>
__global__ void __launch_bounds__(NW == 1 ? 256 : 128, NW == 1 ? wave_occupancy(KR, NW) : (LETKF_TWO_PER_SIMD(KR, NW) ? 2 : 1)) letkf_wave_kernel(const PointArgs A) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int NB = NV + 2;
  constexpr int NBP = (NB + 1) & ~1;
  constexpr int NL = 64 * NW;                 // lanes per point
  int lane = threadIdx.x & (NL - 1);          // lane of the point: column index in the eigen phase, member index after
  int wlane = threadIdx.x & 63;               // lane inside the wavefront (MFMA operand layout)
  const int wvp = (NW == 1) ? 0 : (threadIdx.x >> 6);   // wave inside the point
  const int wv = (NW == 1) ? (threadIdx.x >> 6) : 0;    // point slot inside the workgroup
  const int k = A.k;
  const int nv = A.nv;                        // == NV on the das path, 0 on the letkf_core batch path
  const double km1 = (double)(k - 1);

  double* slice = smem + (size_t)wv * wave_slice_doubles(KR, NV, NW);
  int tile_sz = kTnW * 64;
  if (KR * kVld > tile_sz) tile_sz = KR * kVld;
  int bmat_sz = NBP * KR;
  if (kChunk * KR > bmat_sz) bmat_sz = kChunk * KR;
  if (tile_sz + bmat_sz < wave_base_doubles(KR, NW)) bmat_sz = wave_base_doubles(KR, NW) - tile_sz;
  double* vbuf = slice;                       // [KR][kVld]          (after the Gram phase)
  double* bmat = slice + tile_sz;             // [KR][NBP]
  double* cbuf = bmat + bmat_sz;              // [kChunk][NBP]
  double* wrow = cbuf + kChunk * NBP;         // 3 * kTnW
  double* xsm = wrow + 3 * kTnW;              // 8 * NV + 16
  double* xmean = xsm;
  double* xdet = xsm + NV;
  double* red = xsm + 8 * NV + 16;            // 4 doubles (+ pad): partials of two-wave reductions
  int rslot = 0;

  // Each wave walks a RUN of run_len consecutive points (warm-started eigensolves, see warm_start_product); the 4
  // runs of a workgroup are consecutive too, and workgroups are dealt over the XCDs so that neighbouring points
  // (which gather almost the same obs rows) hit the same L2
  constexpr bool WARM = true;
  constexpr int PPW = (NW == 1) ? 4 : 1;      // points in flight per workgroup
  const int run_len = WARM ? A.run_len : 1;
  // the points as an array [nA][S], p = a S + b: a run walks a at fixed b (S = 1: consecutive points; S = nij1 with
  // gues3d's point order: up a column), run number = chunk * S + b -- neighbouring runs are neighbouring columns
  const long S = A.warm_stride, nA = A.npts / S;
  // this wave's slot of the warm-start workspace: [KR][NL] doubles, lane-fastest
  double* uws = (WARM && run_len > 1) ? A.warm_ws + ((size_t)blockIdx.x * PPW + wv) * ((size_t)KR * NL) + lane : nullptr;
  PROF_DECL
#ifdef LETKF_WAVE_PROF
  long Bstat = blockIdx.x;
#endif
  bool first_draw = true;
  int pend = 0;                                // runs of the drawn unit that are still to do
  long next_rid = 0;
  for (;;) {
   long rid;
   int ir0 = 0, ir1 = run_len;
#ifdef LETKF_WAVE_PROF
   if (!A.sched) {
     // static (PROF twin only, LETKF_AMD_STATIC_SCHED): blocks of PPW consecutive runs, dealt to the workgroups in the
     // order of their XCDs -- what the production kernel did before the dynamic scheduling
     const long nruns = S * ((nA + run_len - 1) / run_len);
     const long nB = (nruns + PPW - 1) / PPW;
     if (Bstat >= nB) break;
     rid = xcd_remap_w(Bstat, nB) * PPW + wv;
     Bstat += gridDim.x;
     if (rid >= nruns) continue;
   } else
#endif
   if (pend > 0) {
     rid = next_rid++;
     --pend;
   } else {
     int code;
#ifdef LETKF_SCHED_RFL   // A/B knob: with the wave number made scalar here hipcc's allocation of the whole kernel changes (scratch 388 -> 612 B/lane)
     const int slot0 = first_draw ? (int)(blockIdx.x >> 3) * PPW + __builtin_amdgcn_readfirstlane(wv) : -1;
#else
     const int slot0 = first_draw ? (int)(blockIdx.x >> 3) * PPW + wv : -1;
#endif
     if constexpr (NW == 1) {
       code = sched_next(A.plan, A.sched, (int)(blockIdx.x & 7), slot0);
     } else {
       int* slot_ = reinterpret_cast<int*>(red + 6);
       __syncthreads();
       if (threadIdx.x < 64) {                  // (wave 0 draws for the workgroup)
         const int g = sched_next(A.plan, A.sched, (int)(blockIdx.x & 7), slot0);
         if (threadIdx.x == 0) *slot_ = g;
       }
       __syncthreads();
       code = *slot_;
     }
     first_draw = false;
     PROF_MARK(9)                              // (PROF twin: the time spent drawing)
     if (code < 0) break;
     PROF_UNIT
     if (code & 4) {
       ir0 = (code & 3) * run_len >> 2;
       ir1 = ((code & 3) + 1) * run_len >> 2;
     }
     rid = (long)(code >> 3) * A.plan.ub;
     const long left = A.plan.nruns - rid;
     pend = (int)(left < A.plan.ub ? left : A.plan.ub) - 1;
     next_rid = rid + 1;
   }
   const long rchunk = rid / S, rb = rid - rchunk * S;
   const long ra0 = rchunk * run_len;
   bool have_u = false;
   [[maybe_unused]] int pre_n = -1;             // mode 3: the NEXT point's list was assembled with this one's (its length; second half of the slot)
   for (int ir = ir0; ir < ir1; ++ir) {
    if (ra0 + ir >= nA) break;
    long pt = (ra0 + ir) * S + rb;
    if constexpr (FUSED == 3) {
      if (A.pt_stride) pt = (ra0 + ir) * A.pt_stride + rb + A.pt0;   // mode 3: columns pt0 .. pt0 + S of a wider domain
    }
    // Everything built from the lane number is the same for every point of the run, so hipcc hoists it out of this
    // loop -- dozens of LDS addresses -- cannot keep it in registers across the eigensolve, and reloads it from scratch
    // one dword at a time, each reload a round trip in front of its use.  Laundering the lane numbers keeps the address
    // arithmetic (one or two integer instructions) where it is used.  Measured (A/B on one box): in the apply phase
    // -0.8 % of the C2 time, in the Gram / tile transposition -2.2 % and k = 100 197 k -> 280 k solves/s; everywhere, as
    // here, another +13 % at k = 100 (316 k) but -1 % on C2 -- so the blanket version is for two-wave points only.
    if constexpr (NW == 2) asm volatile("" : "+v"(lane), "+v"(wlane));
    long o0 = 0;
    int n = 0;
    double beta = 1.0;
    const bool das = A.mode != 1;              // the das_letkf loop body (lists given: 0, search fused in: 2, column survivors: 3)
    if (A.mode == 0) {
      o0 = A.obs_off[pt];
      n = (int)(A.obs_off[pt + 1] - o0);
      if (A.beta) beta = A.beta[pt];
    } else if (FUSED == 3 && A.mode == 3) {
      if (A.beta) beta = A.beta[pt];
      n = 0;
      if constexpr (NW == 1 && FUSED == 3) {
        // ---- column-survivor mode: the horizontal half of obs_local was done once per COLUMN (letkf_survivors_kernel: the
        // rows inside the horizontal cut-off, in the reference's list order, with nd_h and their vertical coordinate); this
        // point adds its vertical half (search_dev::column_vertical_cal -- the expressions of the column search, so the weights
        // equal the lists' to the last bit) and leaves the accepted rows as a local list in this WAVE's slot of a small
        // workspace (written, read back by the Gram phase below and overwritten by the next point: it lives in L2).  The
        // lists of letkf_obs_search_columns_dev -- 20 B per (point, observation): 1 TB written and 1 TB read per analysis at
        // BASELINE configs[3] -- never exist, nor does the count pass over the levels.  (First version: the accepted rows went
        // straight into the Gram's staging buffer, flush and matrix-core steps inside this loop -- the four buffers and the
        // evaluated chunks stayed live across them, 800 B/lane of scratch, slower than the lists.)
        // Two levels per pass: the survivors are read once for this point AND the next one of the run (one level up the same
        // column): the stream is what bounds the pre-pass (configs[3]: 1.1 MB per point and level), the arithmetic doubles per
        // entry and halves per point.  The next point finds its list in the other half of the wave's slot (2 sl_cap entries).
        if (pre_n >= 0) {
          o0 = ((long)blockIdx.x * PPW + wv) * (2 * A.sl_cap) + A.sl_cap;
          n = LETKF_CHECK(pre_n <= A.sl_cap, 7, pre_n, A.sl_cap) ? pre_n : 0;
          pre_n = -1;
        } else if (beta != 0.0) {
          using namespace search_dev;
          const letkf_search_tables& t = A.stab;
          o0 = ((long)blockIdx.x * PPW + wv) * (2 * A.sl_cap);
          const long o1 = o0 + A.sl_cap;
          const bool two = ir + 1 < ir1 && ra0 + ir + 1 < nA;            // (wave-uniform) a next point in this run
          const long ptn = two ? pt + (A.pt_stride ? A.pt_stride : S) : pt;
          const double v_z = A.prz[pt], v_p = log(A.prlev[pt]), l_rain = log(t.rain_base);
          const double v_z1 = A.prz[ptn], v_p1 = log(A.prlev[ptn]);
          const unsigned long long lt_mask = (wlane == 0) ? 0ull : (~0ull >> (64 - wlane));
          long s_lo = A.sv_off[rb], s_hi = A.sv_off[rb + 1];
#ifdef LETKF_CHECKED
          {   // the run's column is one of the launch's, its survivors fit the slot the host sized, the slot is one of the grid's
            bool ok = LETKF_CHECK(rb >= 0 && rb < S, 1, rb, S);
            ok = ok && LETKF_CHECK(s_hi >= s_lo && ((s_hi - s_lo) & 63) == 0, 2, s_hi - s_lo, 64);
            ok = ok && LETKF_CHECK(s_hi - s_lo <= A.sl_cap, 3, s_hi - s_lo, A.sl_cap);
            ok = ok && LETKF_CHECK((long)blockIdx.x < (long)A.wave_grid, 4, blockIdx.x, A.wave_grid);
            if (!ok) s_hi = s_lo;
          }
#endif
          int ntot = 0, ntot1 = 0;
          if (s_hi > s_lo) {
            // Four chunks of 64 entries in flight, in four buffers with STATIC names: the survivors stream from HBM (no wave
            // reads a column's list while it is still in a cache: 320 KB per column at configs[3], 2048 columns in flight).
            // Found in the ISA of the first versions: one chunk ahead = a full memory round trip per chunk; four ahead
            // through a rotation of register copies (a0 = a1; ...) copies the destination of the load issued last --
            // s_waitcnt vmcnt(0) in every iteration; and the 64-bit `entry + lane` offsets were hoisted out of the point
            // loop, spilled, and their scratch_load -- which counts in vmcnt with the prefetches -- waited for everything in
            // flight.  Hence: four evaluations written out, each refilling its own buffer right behind itself, 32-bit
            // offsets from a laundered lane number, every load unconditional from a clamped address.
            const int ns_col = (int)(s_hi - s_lo);                         // (a multiple of 64: the survivor kernel pads)
            const double* sbase = A.surv + 4 * s_lo;
            int wl = wlane;
            asm volatile("" : "+v"(wl));
            auto ld = [&](const int e0, double2& x, double2& y) {
              int e = e0 + wl;
              e = e < ns_col ? e : ns_col - 1;
              x = *reinterpret_cast<const double2*>(&sbase[4 * e]);
              y = *reinterpret_cast<const double2*>(&sbase[4 * e + 2]);
            };
            double2 a0, b0, a1, b1, a2, b2, a3, b3;
            ld(0, a0, b0);
            ld(64, a1, b1);
            ld(128, a2, b2);
            ld(192, a3, b3);
            // the ctype's three numbers through the scalar cache (a chunk is of ONE type: the survivor kernel pads every type's
            // entries to whole chunks with rows outside every cut-off)
            int ic_s = -1, vm_s = 0;
            double vloc_s = 0.0, varloc_s = 0.0;
            auto eval = [&](const double2& ca, const double2& cb, const bool live) {
              const long rw = __double_as_longlong(ca.x);
              const int ic0 = __builtin_amdgcn_readfirstlane((int)(rw >> 32));
              if (ic0 != ic_s) {
                ic_s = ic0;
                vm_s = t.vmode[ic0];
                vloc_s = t.vert_loc[ic0];
                varloc_s = t.varloc[ic0];
              }
              {
                const ColVert vo = column_vertical_cal(vm_s, vloc_s, varloc_s, ca.y, cb.x, cb.y, v_z, v_p, l_rain);
                const bool acc_ = live && vo.rloc != 0.0;                    // :1460
                const unsigned long long mk = __ballot(acc_);
                if (acc_ && LETKF_CHECK(ntot + __popcll(mk) <= A.sl_cap, 5, ntot + __popcll(mk), A.sl_cap)) {
                  const long j = o0 + ntot + __popcll(mk & lt_mask);
                  A.sl_idx[j] = (int)(rw & 0xffffffffL);
                  A.sl_rd[j] = vo.rdiag;
                  A.sl_rl[j] = vo.rloc;
                }
                ntot += __popcll(mk);
              }
              if (two) {                                                     // (wave-uniform)
                const ColVert vo = column_vertical_cal(vm_s, vloc_s, varloc_s, ca.y, cb.x, cb.y, v_z1, v_p1, l_rain);
                const bool acc_ = live && vo.rloc != 0.0;
                const unsigned long long mk = __ballot(acc_);
                if (acc_ && LETKF_CHECK(ntot1 + __popcll(mk) <= A.sl_cap, 6, ntot1 + __popcll(mk), A.sl_cap)) {
                  const long j = o1 + ntot1 + __popcll(mk & lt_mask);
                  A.sl_idx[j] = (int)(rw & 0xffffffffL);
                  A.sl_rd[j] = vo.rdiag;
                  A.sl_rl[j] = vo.rloc;
                }
                ntot1 += __popcll(mk);
              }
            };
            for (int g0 = 0; g0 < ns_col; g0 += 256) {
              eval(a0, b0, true);
              ld(g0 + 256, a0, b0);
              eval(a1, b1, g0 + 64 < ns_col);
              ld(g0 + 320, a1, b1);
              eval(a2, b2, g0 + 128 < ns_col);
              ld(g0 + 384, a2, b2);
              eval(a3, b3, g0 + 192 < ns_col);
              ld(g0 + 448, a3, b3);
            }
          }
          if (two) pre_n = ntot1;
          n = ntot;
        }
      }
    } else if (FUSED == 2 && A.mode == 2) {
      n = -1;                                  // known after the kernel's own walk over the sorting mesh
      if (A.beta) beta = A.beta[pt];
    } else {
      n = A.nobsl[pt];
    }
    if (A.skip_trivial && (n == 0 || beta == 0.0)) {   // done by the streaming pass (letkf_trivial.hip)
      if (beta != 0.0) have_u = false;         // (as below: a point without observations leaves no eigenvectors behind,
      continue;                                //  a beta = 0 point does not touch the run's)
    }
    // per-lane member offset, laundered so that LICM does not park 2*NV hoisted 64-bit offsets in VGPRs
    long moff = (long)lane * A.sm;
    asm volatile("" : "+v"(moff));
    const double* g0 = A.gues ? A.gues + pt * A.sp : nullptr;
    double* a0 = A.anal ? A.anal + pt * A.sp : nullptr;

    if (das && beta == 0.0) {                  // letkf_tools.f90:333-359
      for (int v = 0; v < nv; ++v) {
        if (lane < k && ((A.var_mask >> v) & 1u)) a0[moff + v * A.sv] = g0[k * A.sm + v * A.sv] + g0[moff + v * A.sv];
      }
      const bool mine = lane < nv && ((A.var_mask >> lane) & 1u);
      if (A.det_run && mine) a0[(k + 1) * A.sm + lane * A.sv] = g0[(k + 1) * A.sm + lane * A.sv];
      if (A.rtps_out && mine) A.rtps_out[pt + A.infl_sv * (long)lane] = 1.0;
      if (lane == 0) {
        if (A.status) A.status[pt] = 0;
        if (A.nsweep) A.nsweep[pt] = 0;
        if (A.nobs_out) A.nobs_out[pt] = 0;
      }
      continue;
    }

    bool qskip = false;
    if (das && A.q_update_top > 0.0) qskip = g0[k * A.sm + A.iv_p * A.sv] < A.q_update_top;
    // first variable of this variable-localisation class that is actually updated: its inflation slot drives the solve
    int v0 = 0;
    while (v0 < nv && (!((A.var_mask >> v0) & 1u) || (qskip && v0 >= A.iv_q_first && v0 <= A.iv_q_last))) ++v0;
    double* infl_p = das ? ((v0 < nv) ? &A.infl[pt + A.infl_sv * (long)v0] : nullptr) : &A.infl[pt];
    const double infl_old = infl_p ? *infl_p : 1.0;

    PROF_MARK(0)
    // ------------------------------------------------------------ Gram on the FP64 matrix cores
    // A_aug = Ya^T Ya with Ya = sqrt(w) * [y_1 .. y_k | dep | dep_det]  (n x (k+2)), v_mfma_f64_16x16x4:
    // 4 obs per step.  Lane l supplies, for member block I, Ya[obs0 + (l>>4)][16 I + (l&15)] -- the SAME register
    // is the A operand of tile (I,*) and the B operand of tile (*,I), so the rows are loaded straight from the obs
    // table into MFMA operand layout (128-B coalesced segments), no LDS staging, no broadcast reads; only the
    // tiles I <= J are accumulated.  Column k of A_aug is r = Ys^T sqrt(w) dep, column k+1 the deterministic one,
    // entry (k,k) = sum w dep^2 (parm(1) of the adaptive inflation, common_letkf.f90:233-237).
    double g[KR];
    double racc = 0.0, rdacc = 0.0, p1 = 0.0, p3 = 0.0;
    int sweeps = 0, jconv = 1;
    double lam = km1 / infl_old;               // n == 0: T = sqrt(rho) I, Pa = rho/(k-1) I (common_letkf.f90:89-107)
    bool colvalid = lane < k;                  // does this lane hold an eigen-column?

    bool solved = false;
    if (n != 0) {
      constexpr int NBLK = (KR + 2 + 15) / 16;                 // member blocks incl. the 2 augmented columns
      constexpr int KMIN = (NW == 1) ? wave_kmin(KR, NW) : 1;  // launch_wave_kernel: this instantiation serves wave_kmin <= k <= KR
      // STRIP (r4): where the last block is narrow at compile time -- KR = 50: members 48, 49 and the two augmented columns, 4 of 16;
      // KR = 20: members 16 .. 19 + 2 -- its row / column of tiles does not go to the matrix cores (four 16 x 16 tiles of which
      // 3/4 are padding: 4 of the 10 instructions of a step at k = 50) but to RS x NBLK broadcast-FMAs per step: lane (q, c)
      // accumulates accS[a][I] += Ya[obs_q][16 S + a] * Ya[obs_q][16 I + c] over its own observations (summed over q at the end).
#ifndef LETKF_GRAM_STRIP
#define LETKF_GRAM_STRIP 1
#endif
      constexpr bool STRIP = LETKF_GRAM_STRIP && NW == 1 && (KR == 50 || KR == 20);
      constexpr int NBF = STRIP ? NBLK - 1 : NBLK;             // blocks whose tiles are on the matrix cores
      // The two departure columns do not ride in the narrow block either: sqrt(w) dep is the same number in every lane of an
      // observation's row, so r = Ya^T (sqrt(w) dep) is a plain FMA per block (accD, accDD; entry (k, k) = sum w dep^2 likewise:
      // accP) -- no per-step selects that put the departures into the block's lanes, two broadcast rows fewer.
      constexpr int RS = STRIP ? KR - 16 * (NBLK - 1) : 1;      // member columns of the narrow block (KR = 50: 2, KR = 20: 4)
      static_assert(!STRIP || (RS >= 1 && RS <= 6 && KMIN > 16 * (NBLK - 1)), "the narrow block must be the last one for every k of the instantiation");
      constexpr int NTILE = NBF * (NBF + 1) / 2;
      v4d acc[NTILE];
      [[maybe_unused]] double accS[RS][NBLK], accD[NBLK], accDD[NBLK], accP = 0.0;
#pragma unroll
      for (int I = 0; I < NBLK; ++I) accD[I] = accDD[I] = 0.0;
#pragma unroll
      for (int t = 0; t < NTILE; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int a = 0; a < RS; ++a)
#pragma unroll
        for (int I = 0; I < NBLK; ++I) accS[a][I] = 0.0;
      int q = wlane >> 4, c16 = wlane & 15;
      asm volatile("" : "+v"(q), "+v"(c16));   // (not loop-invariant for hipcc: see the apply phase)

      // Two phases per batch of kSC observations, so that the per-observation scalars are computed ONCE (lane = obs,
      // coalesced reads of the CSR slice, all gathers of a batch in flight together) instead of 16 times over in
      // every 4-obs MFMA step -- the first version spent ~250 VALU instructions and a dozen s_waitcnt per 4 obs in
      // its load pipeline, as much SIMD time as a Jacobi sweep per point:
      //   stage: lane i -> obs i: row base, sqrt(w), sqrt(w) dep, sqrt(w) dep_det into LDS (4 doubles per obs)
      //   MFMA : per 4-obs step 2 ds_read_b128 + NBLK row loads (three steps in flight) + NBLK multiplies
      constexpr int kSC = 256;                                 // obs per batch (4 kSC doubles <= wave_base_doubles of the slice)
      double* stg = slice;
      const bool mode0 = A.mode != 1;                          // rows come from the obs table (member-fastest)
      const double* ybase = mode0 ? A.ensval : A.hdxb;
      bool rowok[NBLK];
      long mo[NBLK];
#pragma unroll
      for (int I = 0; I < NBLK; ++I) {
        const int m = 16 * I + c16;
        rowok[I] = m < k;
        const long mm = rowok[I] ? m : 0;
        mo[I] = mode0 ? mm : mm * (long)A.nobs;
      }
      const bool is_d = c16 == (k & 15), is_dd = c16 == ((k + 1) & 15);   // lanes of the two augmented columns
      const int blk_d = k >> 4, blk_dd = (k + 1) >> 4;
      struct Step {
        double f[NBLK];
        double sw, dsw, ddsw;
      };

      // MFMA steps of 4 obs over the first nsp (multiple of 4) staged entries; with two waves each takes every other step
      auto run_steps = [&](const int nsp) {
        const int nch = nsp >> 2;
        auto fetch = [&](const int c, Step& t) {
          const bool ok = c < nch;
          const int i = 4 * (ok ? c : 0) + q;
          const double2 a2 = *reinterpret_cast<const double2*>(&stg[4 * i]);
          const double2 b2 = *reinterpret_cast<const double2*>(&stg[4 * i + 2]);
          const long rb = __double_as_longlong(a2.x);
          t.sw = ok ? a2.y : 0.0;
          t.dsw = ok ? b2.x : 0.0;
          t.ddsw = ok ? b2.y : 0.0;
#pragma unroll
          for (int I = 0; I < NBLK; ++I) t.f[I] = ybase[rb + mo[I]];
        };
        auto mma = [&](const Step& t) {
          double y[NBLK];
#pragma unroll
          for (int I = 0; I < NBLK; ++I) {
            double v = t.f[I] * t.sw;
            // (blocks that lie below the smallest k this instantiation is dispatched for are all members at compile
            // time: hipcc turns the wave-uniform test into 9 v_cndmask per block and step otherwise)
            if (16 * (I + 1) > KMIN && 16 * (I + 1) > k) {     // wave-uniform: block reaches past the members
              v = rowok[I] ? v : 0.0;
              if constexpr (!STRIP) {
                if (I == blk_d && is_d) v = t.dsw;
                if (I == blk_dd && is_dd) v = t.ddsw;
              }
            }
            y[I] = v;
          }
          if constexpr (STRIP) {
#pragma unroll
            for (int I = 0; I < NBLK; ++I) {
              accD[I] = fma(y[I], t.dsw, accD[I]);
              accDD[I] = fma(y[I], t.ddsw, accDD[I]);
            }
            accP = fma(t.dsw, t.dsw, accP);
          }
          int tt = 0;
#pragma unroll
          for (int I = 0; I < NBF; ++I)
#pragma unroll
            for (int J = I; J < NBF; ++J) {
              acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(y[I], y[J], acc[tt], 0, 0, 0);
              ++tt;
            }
          if constexpr (STRIP) {
            auto strip_row = [&](auto a_) {
              constexpr int a = decltype(a_)::value;
              if constexpr (a < RS) {
#pragma unroll
                for (int I = 0; I < NBLK; ++I) {
                  if (a == 0 && I == 0) fmac_row_bcast<a, true>(accS[a][I], y[NBLK - 1], y[I]);
                  else fmac_row_bcast<a, false>(accS[a][I], y[NBLK - 1], y[I]);
                }
              }
            };
            strip_row(std::integral_constant<int, 0>{});
            strip_row(std::integral_constant<int, 1>{});
            strip_row(std::integral_constant<int, 2>{});
            strip_row(std::integral_constant<int, 3>{});
            strip_row(std::integral_constant<int, 4>{});
            strip_row(std::integral_constant<int, 5>{});
          }
        };
        // PD steps in flight.  Three things keep the pipeline the way it is written (each found in the ISA):
        //  * no conditions around the fetches (a step past the end has sw = dsw = ddsw = 0 and adds nothing): with
        //    branches hipcc can no longer count the loads in flight and waits for ALL of them -- s_waitcnt vmcnt(0) --
        //    at the top of every iteration, one exposed L2 latency per iteration;
        //  * the empty asm statements keep each fetch where it is written: left alone, hipcc rotates the loop so that
        //    all loads sit at the top of the iteration that consumes them;
        //  * the loaded rows are routed through one just before their use: otherwise the multiplies of the LATER steps
        //    are hoisted to the top of the iteration and wait for the youngest loads there.
        if constexpr (NW == 1) {
          constexpr int PD = kGramDepth;
          Step ts[PD];
#pragma unroll
          for (int u = 0; u < PD; ++u) {
            fetch(wvp + u * NW, ts[u]);
            asm volatile("" ::: "memory");       // (same order as in the loop: the waits are counted statically)
          }
          auto pin = [&](Step& t) {
#pragma unroll
            for (int I = 0; I < NBLK; ++I) asm volatile("" : "+v"(t.f[I])::"memory");
          };
          for (int c = wvp; c < nch; c += PD * NW) {
#pragma unroll
            for (int u = 0; u < PD; ++u) {
              pin(ts[u]);
              if (u == 0 || c + u * NW < nch) mma(ts[u]);   // (wave-uniform; only the matrix instructions are skipped)
              fetch(c + (PD + u) * NW, ts[u]);
            }
          }
        } else {
          // two-wave points (up to 7 member blocks, part of the register file in AGPRs) keep the plain loop; their
          // register allocation is fragile -- the compile-time full blocks (KMIN) alone cost k = 100 a quarter of its
          // speed (162 k -> 116 k solves/s, A/B on one box), so none of the one-wave changes is applied to them
          Step t0, t1, t2;
          fetch(wvp, t0);
          fetch(wvp + NW, t1);
          fetch(wvp + 2 * NW, t2);
          for (int c = wvp; c < nch; c += 3 * NW) {
            mma(t0);
            fetch(c + 3 * NW, t0);
            if (c + NW < nch) {
              mma(t1);
              fetch(c + 4 * NW, t1);
            }
            if (c + 2 * NW < nch) {
              mma(t2);
              fetch(c + 5 * NW, t2);
            }
          }
        }
      };

      if (FUSED == 2 && A.mode == 2) {
        // ---- obs_local fused in (no-limit mode, scale/letkf/letkf_tools.f90:1438-1476): walk the rectangle of
        // sorting-mesh cells of every observation type exactly like letkf_search_kernel, evaluate obs_local_cal per
        // lane for 64 candidate rows at a time, and append the accepted ones (ballot + prefix popcount: the
        // reference's list order) straight to the staging buffer -- the local list never exists in memory.
        if constexpr (NW == 1 && FUSED == 2) {
          using namespace search_dev;
          const letkf_search_tables& t = A.stab;
          const double ri = A.pri[pt], rj = A.prj[pt], rlev = A.prlev[pt], rz = A.prz[pt];
          const unsigned long long lt_mask = (wlane == 0) ? 0ull : (~0ull >> (64 - wlane));
          int cnt = 0, ntot = 0, nconv = 0;                   // staged entries / accepted so far / entries at the front
                                                              // that are already in final form (wave-uniform)
          // stage buffer, phase A: (row, rdiag, rloc) as the candidates are accepted; phase B (convert): lane = entry,
          // all dep / dep_det gathers of the batch in flight together -> (row base, sqrt(w), sqrt(w) dep, sqrt(w) dep_det)
          auto convert = [&]() {
            for (int i = nconv + wlane; i < cnt; i += 64) {
              const double2 a2 = *reinterpret_cast<const double2*>(&stg[4 * i]);
              const double rloc_ = stg[4 * i + 2];
              const long row = __double_as_longlong(a2.x);
              const long rb = row * A.kld;
              const double d = A.dep[row];
              const double dd = A.det_run ? A.ensval[rb + k] : 0.0;
              const double sw = fast_rsqrt(a2.y);
              p3 += rloc_;
              *reinterpret_cast<double2*>(&stg[4 * i]) = double2{__longlong_as_double(rb), sw};
              *reinterpret_cast<double2*>(&stg[4 * i + 2]) = double2{d * sw, dd * sw};
            }
          };
          struct Meta {
            double lev, dat, ori, orj, err;
          };
          psync<NW>();
          for (int m = 0; m < t.group_start[t.ngroup]; ++m) {
            const int ic = t.group_member[m];
            const int vm = t.vmode[ic];
            const double dzi = t.hori_loc[ic] * kDistZeroFac / t.dx;        // obs_local_range :1775-1778
            const double dzj = t.hori_loc[ic] * kDistZeroFac / t.dy;
            int imin, imax, jmin, jmax;
            ij_obsgrd_ext(t, ic, ri - dzi, rj - dzj, imin, jmin);
            ij_obsgrd_ext(t, ic, ri + dzi, rj + dzj, imax, jmax);
            imin = max(imin, 1);
            jmin = max(jmin, 1);
            imax = min(imax, t.ngrdext_i[ic]);
            jmax = min(jmax, t.ngrdext_j[ic]);
            if (imin > imax || jmin > jmax) continue;
            const long acb = t.ac_off[ic];
            const int ld = t.ngrdext_i[ic] + 1;
            // metadata of the candidate rows [base, base + 64): loaded one chunk AHEAD of its evaluation (the first
            // version loaded and evaluated chunk by chunk and waited three dependent memory round trips per chunk --
            // 165 ms per C2 analysis against 59 ms for the stand-alone search kernel)
            auto load_meta = [&](const int base, const int hi) -> Meta {
              Meta q_{1.0, 1.0, 0.0, 0.0, 1.0};
              const int row = base + wlane;
              if (row < hi) {
                if (vm != 2 && vm != 3) q_.lev = t.ob_lev[row];
                if (vm == 2) q_.dat = t.ob_dat[row];
                q_.ori = t.ob_ri[row];
                q_.orj = t.ob_rj[row];
                q_.err = t.ob_err[row];
              }
              return q_;
            };
            for (int j0 = jmin; j0 <= jmax; j0 += 64) {                      // 64 mesh rows at a time: lane = row
              const int nr = min(64, jmax - j0 + 1);
              int lo_l = 0, hi_l = 0;
              if (wlane < nr) {
                lo_l = t.ac_ext[acb + (imin - 1) + (long)ld * (j0 + wlane - 1)];
                hi_l = t.ac_ext[acb + imax + (long)ld * (j0 + wlane - 1)];
              }
              int r = 0;
              int base = __shfl(lo_l, 0, 64), hi = __shfl(hi_l, 0, 64);
              auto skip_empty = [&]() {
                while (r < nr && base >= hi) {
                  ++r;
                  if (r < nr) {
                    base = __shfl(lo_l, r, 64);
                    hi = __shfl(hi_l, r, 64);
                  }
                }
              };
              skip_empty();
              Meta cur{1.0, 1.0, 0.0, 0.0, 1.0};
              if (r < nr) cur = load_meta(base, hi);
              while (r < nr) {
                const int cb = base, chi = hi;
                base += 64;
                skip_empty();
                Meta nxt{1.0, 1.0, 0.0, 0.0, 1.0};
                if (r < nr) nxt = load_meta(base, hi);
                const int row = cb + wlane;
                CalOut c{0.0, -1.0, -1.0};
                if (row < chi) c = local_cal_v(t, ic, ri, rj, rlev, rz, cur.lev, cur.dat, cur.ori, cur.orj, cur.err);
                const bool acc_ = c.rloc != 0.0;                             // :1460
                const unsigned long long mk = __ballot(acc_);
                if (acc_) {
                  const int i = cnt + __popcll(mk & lt_mask);
                  *reinterpret_cast<double2*>(&stg[4 * i]) = double2{__longlong_as_double((long)row), c.rdiag};
                  stg[4 * i + 2] = c.rloc;
                }
                const int na = __popcll(mk);
                cnt += na;
                ntot += na;
                if (cnt > kSC - 64) {                                        // room for one more chunk is gone
                  const int nuse = cnt & ~3;
                  psync<NW>();
                  convert();
                  psync<NW>();
                  run_steps(nuse);
                  psync<NW>();
                  // the 0..3 left-over entries (already converted) move to the front
                  double2 l0{0.0, 0.0}, l1{0.0, 0.0};
                  if (wlane < cnt - nuse) {
                    l0 = *reinterpret_cast<const double2*>(&stg[4 * (nuse + wlane)]);
                    l1 = *reinterpret_cast<const double2*>(&stg[4 * (nuse + wlane) + 2]);
                  }
                  psync<NW>();
                  if (wlane < cnt - nuse) {
                    *reinterpret_cast<double2*>(&stg[4 * wlane]) = l0;
                    *reinterpret_cast<double2*>(&stg[4 * wlane + 2]) = l1;
                  }
                  cnt -= nuse;
                  nconv = cnt;
                }
                cur = nxt;
              }
            }
          }
          psync<NW>();
          convert();
          const int nsp = (cnt + 3) & ~3;
          if (wlane < nsp - cnt) {                                           // pad the last step with weight-0 rows
            *reinterpret_cast<double2*>(&stg[4 * (cnt + wlane)]) = double2{0.0, 0.0};
            *reinterpret_cast<double2*>(&stg[4 * (cnt + wlane) + 2]) = double2{0.0, 0.0};
          }
          psync<NW>();
          if (nsp > 0) run_steps(nsp);                                       // (an empty buffer has no valid row to prefetch)
          n = ntot;
        }
      } else
      for (int s0 = 0; s0 < n; s0 += kSC) {
        const int ns = min(kSC, n - s0);
        const int nsp = (ns + 3) & ~3;
        psync<NW>();                                           // the previous batch has been consumed
        // ---- stage
        constexpr int NPASS = kSC / NL;
        int iob[NPASS];
        double rdv[NPASS], rlv[NPASS], dv[NPASS], ddv[NPASS];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
          const int i = ps * NL + lane;
          iob[ps] = 0;
          rdv[ps] = 1.0;
          rlv[ps] = 0.0;
          dv[ps] = 0.0;
          ddv[ps] = 0.0;
          if (i < ns) {
            if (mode0) {
              const long e = o0 + s0 + i;
              iob[ps] = A.obs_idx[e];
              rlv[ps] = A.rloc_l[e];
              rdv[ps] = A.rdiag_l[e];
            } else {
              const long e = pt * (long)A.nobs + s0 + i;
              rlv[ps] = A.rloc[e];
              rdv[ps] = A.rdiag[e];
              dv[ps] = A.depv[e];
              if (A.depd) ddv[ps] = A.depd[e];
            }
          }
        }
        if (mode0) {
#pragma unroll
          for (int ps = 0; ps < NPASS; ++ps) {
            const int i = ps * NL + lane;
            if (i < ns) {
              dv[ps] = A.dep[iob[ps]];
              if (A.det_run) ddv[ps] = A.ensval[(long)iob[ps] * A.kld + k];
            }
          }
        }
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
          const int i = ps * NL + lane;
          if (i < nsp) {
            // sqrt(w) with w = 1/rdiag (or rloc/rdiag): one rsqrt + Newton instead of an IEEE division and sqrt
            double sw = 0.0;
            long rb = 0;
            if (i < ns) {
              sw = fast_rsqrt(rdv[ps]);
              if (!mode0 && !A.rdiag_wloc) sw *= sqrt(rlv[ps]);
              rb = mode0 ? (long)iob[ps] * A.kld : pt * (long)A.nobs * (long)k + (s0 + i);
              p3 += rlv[ps];
            }
            *reinterpret_cast<double2*>(&stg[4 * i]) = double2{__longlong_as_double(rb), sw};
            *reinterpret_cast<double2*>(&stg[4 * i + 2]) = double2{dv[ps] * sw, ddv[ps] * sw};
          }
        }
        psync<NW>();
        run_steps(nsp);
      }
      PROF_MARK(1)
      if (n > 0) {
      solved = true;
      // accumulator tiles -> "lane j owns column j": 16 rows at a time through LDS.  C/D layout of the f64 MFMA:
      // lane l holds rows (l>>4) + 4*reg, column l&15 of its 16x16 tile.
      constexpr int LDA = 18;
      double* abuf = slice;                                    // [64 NW][LDA], spans the tile + bmat regions
      if constexpr (STRIP) {                                   // the narrow block's sums over the four observation residues q
#pragma unroll
        for (int a = 0; a < RS; ++a)
#pragma unroll
          for (int I = 0; I < NBLK; ++I) {
            accS[a][I] += wshfl_xor(accS[a][I], 16);
            accS[a][I] += wshfl_xor(accS[a][I], 32);
          }
#pragma unroll
        for (int I = 0; I < NBLK; ++I) {
          accD[I] += wshfl_xor(accD[I], 16);
          accD[I] += wshfl_xor(accD[I], 32);
          accDD[I] += wshfl_xor(accDD[I], 16);
          accDD[I] += wshfl_xor(accDD[I], 32);
        }
        accP += wshfl_xor(accP, 16);
        accP += wshfl_xor(accP, 32);
        // lane j = 16 q + c owns column j: its r_j, r_det_j are block q's sums (every lane (., c) holds the sums of all blocks)
        racc = rdacc = 0.0;
#pragma unroll
        for (int I = 0; I < NBLK; ++I)
          if (q == I) {
            racc = accD[I];
            rdacc = accDD[I];
          }
        p1 = accP;                                             // A_aug[k][k] = sum w dep^2
      }
#pragma unroll
      for (int I = 0; I < NBLK; ++I) {
        psync<NW>();
        // wave 0 stores its partial tiles, wave 1 (two-wave points) adds its own on top
#pragma unroll
        for (int pass = 0; pass < NW; ++pass) {
          if (pass == 1) __syncthreads();
          if (wvp == pass) {
            if constexpr (STRIP) {
              // rows of block I < S: the columns 16 S + a are accS[a][I] (lane c = the row); rows of block S (a < RS): every
              // column 16 J + c is accS[a][J] -- element (row, col) at abuf[col * LDA + row-in-block], as the tiles below.
              // (Every q row writes -- the same values, folded above: NOT `if (q == 0)`.  At the join of that branch hipcc put a
              // register spill in front of the exec restore in letkf_wave_kernel<20, 11, false> -- the defect of DESIGN.md section 8,
              // caught by tools/isa_exec_audit.py at build time when wave_sum changed the allocation.)
              {
                if (I < NBLK - 1) {
#pragma unroll
                  for (int a = 0; a < 16; ++a) abuf[(16 * (NBLK - 1) + a) * LDA + c16] = a < RS ? accS[a][I] : 0.0;   // (the padding columns: zeros, as the tiles left them)
                } else {
#pragma unroll
                  for (int J = 0; J < NBLK; ++J)
#pragma unroll
                    for (int a = 0; a < RS; ++a) abuf[(16 * J + c16) * LDA + a] = accS[a][J];
                }
              }
            }
#pragma unroll
            for (int J = 0; J < NBF; ++J) {
              if (STRIP && I == NBLK - 1) continue;            // (the narrow block's rows: written above)
              const int ti = I <= J ? I : J, tj = I <= J ? J : I;
              const int t = ti * NBF - ti * (ti - 1) / 2 + (tj - ti);
#pragma unroll
              for (int reg = 0; reg < 4; ++reg) {
                const int a = q + 4 * reg, b = c16;            // tile-local (row, col) of this element
                // rows of block I, columns of block J: element (I:a, J:b) directly, or the mirror of tile (J, I)
                double* dst = (I <= J) ? &abuf[(16 * J + b) * LDA + a] : &abuf[(16 * J + a) * LDA + b];
                if (pass == 0) *dst = acc[t][reg];
                else *dst += acc[t][reg];
              }
            }
          }
        }
        psync<NW>();
        if constexpr (NW == 1) {
          // Every lane reads (the lanes past the last block a copy of its last row; their g is zeroed below, racc / rdacc keep
          // their value through a select): NOT `if (lane < 16 * NBLK)`.  Behind the join of that branch hipcc (ROCm 7.2) put the
          // copies of a live-range split IN FRONT of the instruction that re-enables the lanes which skipped it -- found in
          // letkf_wave_kernel<16, 11, true> and <20, 0, true>: the run scheduler's `pend` saved for lanes 0..31 only and restored
          // for all 64, lanes 32..63 walked on into points that were not theirs with a stale slice (memory fault).  tools/
          // isa_exec_audit.py looks for that pattern in every unit's ISA; the Makefile runs it on every build.
          const int lrow = lane < 16 * NBLK ? lane : 16 * NBLK - 1;
          const bool mine = lane < 16 * NBLK;
#pragma unroll
          for (int e = 0; e < 16; e += 2) {
            if (16 * I + e < KR) {
              const double2 v2 = *reinterpret_cast<const double2*>(&abuf[lrow * LDA + e]);
              g[16 * I + e] = v2.x;
              g[16 * I + e + 1] = v2.y;
            }
          }
          if constexpr (!STRIP) {
            if ((k >> 4) == I) {
              const double t = abuf[lrow * LDA + (k & 15)];
              racc = mine ? t : racc;
            }
            if (((k + 1) >> 4) == I) {
              const double t = abuf[lrow * LDA + ((k + 1) & 15)];
              rdacc = mine ? t : rdacc;
            }
          }
        } else if (lane < 16 * NBLK) {
#pragma unroll
          for (int e = 0; e < 16; e += 2) {
            if (16 * I + e < KR) {
              const double2 v2 = *reinterpret_cast<const double2*>(&abuf[lane * LDA + e]);
              g[16 * I + e] = v2.x;
              g[16 * I + e + 1] = v2.y;
            }
          }
          if constexpr (!STRIP) {
            if ((k >> 4) == I) racc = abuf[lane * LDA + (k & 15)];
            if (((k + 1) >> 4) == I) rdacc = abuf[lane * LDA + ((k + 1) & 15)];
          }
        }
        if constexpr (!STRIP) {
          if ((k >> 4) == I) p1 = abuf[k * LDA + (k & 15)];    // A_aug[k][k] = sum w dep^2
        }
      }
      psync<NW>();
      // rows >= k of a column (the augmented rows) and whole columns >= k play no part in the eigenproblem
#pragma unroll
      for (int r = 0; r < KR; ++r)
        if (r >= k || lane >= k || lane >= 16 * NBLK) g[r] = 0.0;
      // diagonal: trace for the adaptive inflation, then the shift (common_letkf.f90:140-143)
      const double shift = km1 / infl_old;
      double diag = 0.0;
#pragma unroll
      for (int r = 0; r < KR; ++r) {
        if (r == lane && lane < k) {
          diag = g[r];
          g[r] += shift;
        }
      }
      double parm1 = 0.0, parm2 = 0.0, parm3 = 0.0;
      if (A.infl_adaptive) {
        parm1 = p1;
        parm3 = preduce<NW, 0>(p3, red, rslot);
        parm2 = preduce<NW, 0>(lane < k ? diag : 0.0, red, rslot) / km1;
      }

      PROF_MARK(2)
      // ------------------------------------------------------------ eigen-decomposition in registers
      if constexpr (WARM) {
        if (have_u && !(A.warm_dbg & 1)) {
          if constexpr (NW == 1 && KR <= 50) warm_start_product_mfma<KR>(g, uws - lane, slice);
          else warm_start_product<KR, NW>(g, uws, k, slice);
        }
      }
      PROF_MARK(3)
      sweeps = jacobi_split<KR, NW, wave_jacobi_rc(KR, NW), LETKF_EARLY_NW(KR, NW), LETKF_INPLACE_NW(KR, NW)>(g, k, A.max_sweep, slice, nullptr, &jconv);   // exchange buffer: 64*10 + 64*4 doubles of the tile+bmat region
      // per-lane values that were spilled around the eigensolve come back HERE, in one batch: reloaded lazily, each
      // scratch load sits behind the 50 workspace stores below and its s_waitcnt vmcnt(0) waits for all of them
      asm volatile("" : "+v"(racc), "+v"(rdacc), "+v"(moff));
      PROF_MARK(4)

      double ss = 0.0;
#pragma unroll
      for (int r = 0; r < KR; ++r) ss = fma(g[r], g[r], ss);
      lam = sqrt(ss);
      colvalid = ss > 0.0;     // after the rotate-and-swap sweeps the columns sit in permuted lanes; with an odd k the
                               // inert zero column that pads the line to even length can be anywhere among them
      const double il = colvalid ? 1.0 / lam : 0.0;
#pragma unroll
      for (int r = 0; r < KR; ++r) g[r] *= il;

      // adaptive inflation (common_letkf.f90:233-254), old rho everywhere above
      if (A.infl_adaptive) {
        const double parm4 = (parm1 - parm3) / parm2 - infl_old;
        const double tq = (infl_old * parm2 + parm3) / parm2;
        const double sigma_o = 2.0 / parm3 * (tq * tq);
        const double gain = 0.04 * 0.04 / (sigma_o + 0.04 * 0.04);
        p1 = infl_old + gain * parm4;            // reuse p1 as infl_new
      }
      }
    }
    // apply phase on the matrix cores (see below); these instantiations also give points without observations a
    // closed-form path that never touches g
    constexpr bool MAPPLY = !KKOUT && NW == 1 && KR <= 50 && NV > 0 && NB <= 14;
    constexpr int kBmRows = 4 * ((KR + 3) / 4), kBmOff = 32 * KR;   // MAPPLY: padded B [kBmRows][16] in the wave's LDS slice
    constexpr int kObOff = (32 * KR >= 1024) ? 0 : kBmOff + 16 * kBmRows + 128;   // MAPPLY: output transposition buffer [64][16], clear of B
    if constexpr (!MAPPLY) {
      if (!solved) {
#pragma unroll
        for (int r = 0; r < KR; ++r) g[r] = (r == lane && lane < k) ? 1.0 : 0.0;
      }
    }
    const double infl_new = (A.infl_adaptive && n > 0) ? p1 : infl_old;
    double xv[NV > 0 ? NV : 1];                 // MAPPLY: x'_v of member `lane`
    double xm_l = 0.0, xd_l = 0.0;             // MAPPLY: lane v < NV holds x-bar_v and the deterministic member of variable v
    if constexpr (MAPPLY) {
      // the state loads of the apply phase, issued here so that their latency (8-byte accesses npts*8 B apart) runs
      // under the normalisation, the workspace store and the status reductions
      const double* gp = g0 + moff;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        xv[v] = (lane < k) ? *gp : 0.0;
        gp += A.sv;
      }
      if (lane < NV) {
        xm_l = g0[k * A.sm + lane * A.sv];
        xd_l = A.det_run ? g0[(k + 1) * A.sm + lane * A.sv] : 0.0;
      }
      asm volatile("" ::: "memory");
    }
    if constexpr (WARM) {
      // leave the eigenvectors behind for the next point of the run.  Here, while g is still entirely in registers:
      // further down part of it is spilled, and a store loop that alternates scratch reloads with global stores
      // pays one store-acknowledge latency per element (s_waitcnt vmcnt counts both) -- measured 41 us per point.
      if (uws && solved && !(A.warm_dbg & 2)) {
        // (the pointer is laundered every 8 rows: otherwise all KR row addresses are hoisted out of the point loop
        // as 64-bit values, spilled, and reloaded one by one in front of each store -- same serialisation)
        unsigned long long pa = reinterpret_cast<unsigned long long>(uws);
#pragma unroll
        for (int r = 0; r < KR; ++r) {
          if ((r & 7) == 0) asm volatile("" : "+v"(pa));
          ((gdouble*)pa)[(size_t)(r & 7) * NL] = g[r];   // global_store, not flat_store: see warm_start_product_mfma
          if ((r & 7) == 7) pa += 8 * NL * sizeof(double);
        }
      }
    }

    // ------------------------------------------------------------ status (common_mtx.f90:66-78)
    int st = 0;
    {
      const double lmx = preduce<NW, 1>(colvalid ? lam : 0.0, red, rslot);
      const double lmn = preduce<NW, 2>(colvalid ? lam : 1e300, red, rslot);
      if (!jconv && A.max_sweep >= 60) st = 1;   // (a solve that converges in the last permitted sweep is converged)
      else if (!(lmx > 0.0)) st = 2;
      else if (lmn < lmx * 1.4901161193847656e-08) st = 3;
    }
    // (a point without observations hands nothing on: its V = I would be a cold start anyway)
    if constexpr (WARM) have_u = uws != nullptr && st == 0 && solved;
    const double sc1 = colvalid ? sqrt(km1 / lam) : 0.0;      // T spectrum
    const double sc2 = colvalid ? 1.0 / lam : 0.0;            // Pa spectrum

    PROF_MARK(5)
    // ------------------------------------------------------------ apply phase
    // U = V^T B, C = D U, Out = V C.  One-wave points with KR <= 50 run both products on the FP64 matrix cores
    // (MAPPLY); the others (two-wave points, the T / Pa instantiations) keep the LDS-broadcast version below.
    const int mrow_l = lane < KR ? lane : KR - 1;
    double cf[NV > 0 ? NV : 1];
    double out[NB];
    if constexpr (MAPPLY) {
      // The broadcast version is LDS-latency-bound like the old warm-start product was (16 % of the wave time on C2,
      // PROF build).  Here V is parked in LDS 32 columns at a time ([col][row], like A in warm_start_product_mfma --
      // a whole V plus B does not fit the 20 KB slice), and per half h:
      //   U tile il (rows j = 32h + 16 il + i): A operand V[4s+q][j] (ds_read_b64), B operand B[4s+q][c] (c = b < 14)
      //   C = D U on the accumulators: register `reg` of lane (q, c) is U[32h + 16 il + 4 reg + q][c] -- which is
      //     exactly the B-operand layout of contraction step s = 8h + 4 il + reg of Out = V C (j = 4s + q): no exchange
      //   Out tile I (rows m = 4i + I): A operand V[4c + I][4s + q] = 4 consecutive doubles of LDS column 4s+q.
      // var_a of the RTPS factor is summed from the same accumulators, var_g from the B operands.
      if (!solved) {
        // No observations: V = I and every eigenvalue is (k-1)/rho (common_letkf.f90:89-107), so U = B, w-bar = 0 and
        // T x' = sqrt(rho) x' in closed form.  (Besides saving these points the matrix work, this keeps g out of the
        // merge of the two paths: as a 50-register phi it cost every SOLVED point ~15 serialised scratch-to-scratch
        // copies, found in the ISA.)
        out[0] = 0.0;
        out[1] = 0.0;
        wave_lds_sync();
        if (lane < kBmRows) {
#pragma unroll
          for (int v = 0; v < NV; ++v) slice[kBmOff + lane * 16 + 2 + v] = xv[v];
        }
        wave_lds_sync();
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          out[2 + v] = sc1 * xv[v];
          double cfv = 1.0;
          if (A.relax_alpha != 0.0) {
            cfv = 1.0 - A.relax_alpha;
          } else if (A.relax_alpha_spread != 0.0) {
            const double parm = A.relax_to_inflated_prior ? A.infl[pt + A.infl_sv * (long)v] : 1.0;   // :387-391
            const double var_g = wave_sum(xv[v] * xv[v]);
            const double var_a = var_g * uniform(sc2);
            if (var_g > 0.0 && var_a > 0.0)
              cfv = A.relax_alpha_spread * sqrt(var_g * parm / (var_a * km1)) - A.relax_alpha_spread + 1.0;
          }
          cf[v] = uniform(cfv);
          if (A.rtps_out && lane == 0 && ((A.var_mask >> v) & 1u)) {
            const bool skipv = qskip && v >= A.iv_q_first && v <= A.iv_q_last;
            A.rtps_out[pt + A.infl_sv * (long)v] = (A.relax_alpha == 0.0 && A.relax_alpha_spread != 0.0 && !skipv) ? cfv : 1.0;
          }
        }
      } else {
      // No predicates anywhere in the two products (hipcc wraps every predicated LDS read in its own exec-mask
      // branch: 29 branches, ~100 v_readlane of spilled masks in the first version): B is stored 16 columns wide and
      // 4 KS rows deep with zeros in the padding, so a contraction row >= KR or a column >= NB multiplies zero;
      // eigen-columns j >= k have spectra (sc1, sc2) = 0, so whatever finite numbers U holds in their rows never
      // reach C; rows m >= KR of Out are computed from stale LDS contents and dropped.
      constexpr int KS = (KR + 3) / 4;
      constexpr int BR = 4 * KS;               // rows of the padded B
      double* vh = slice;                      // [32][KR]
      double* bm = slice + 32 * KR;            // [BR][16]
      double* scl = bm + BR * 16;              // [64][2]: (1/lam, sqrt((k-1)/lam)) per eigen-column
      // (q and c are laundered: they are the same for every point of the wave's run, so hipcc hoists the ~40 LDS
      // addresses built from them out of the point loop, keeps them in scratch, and reloads one -- a scratch round
      // trip -- in front of every MFMA step; found in the ISA)
      int q = wlane >> 4, c = wlane & 15;
      asm volatile("" : "+v"(q), "+v"(c));
      wave_lds_sync();
      {
        if (lane < BR) {
          double brow[16];
          brow[0] = (lane < k) ? racc : 0.0;
          brow[1] = (lane < k) ? rdacc : 0.0;
#pragma unroll
          for (int b = 2; b < 16; ++b) brow[b] = (b - 2 < NV) ? xv[b - 2 < NV ? b - 2 : 0] : 0.0;
          double* row = bm + lane * 16;
#pragma unroll
          for (int b = 0; b < 16; b += 2) *reinterpret_cast<double2*>(&row[b]) = double2{brow[b], brow[b + 1]};
        }
        *reinterpret_cast<double2*>(&scl[2 * lane]) = double2{sc2, sc1};
      }
      v4d accO[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) accO[t] = v4d{0.0, 0.0, 0.0, 0.0};
      double va = 0.0, vg = 0.0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (32 * h < KR) {
          wave_lds_sync();
          if ((lane >> 5) == h && lane < KR) {
            double* mine = vh + (lane - 32 * h) * KR;
#pragma unroll
            for (int r = 0; r < KR; r += 2) *reinterpret_cast<double2*>(&mine[r]) = double2{g[r], g[r + 1]};
          }
          wave_lds_sync();
          v4d accU[2];
          accU[0] = v4d{0.0, 0.0, 0.0, 0.0};
          accU[1] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s_ = 0; s_ < KS; ++s_) {
            const double bq = bm[(4 * s_ + q) * 16 + c];
            if (h == 0) vg = fma(bq, bq, vg);
#pragma unroll
            for (int il = 0; il < 2; ++il) {
              if (32 * h + 16 * il < KR) {
                const double a = vh[(16 * il + c) * KR + 4 * s_ + q];
                accU[il] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bq, accU[il], 0, 0, 0);
              }
            }
          }
#pragma unroll
          for (int il = 0; il < 2; ++il) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
              const int s_ = 4 * (2 * h + il) + reg;       // contraction step of Out: eigen-columns j = 4 s_ + q
              if (4 * s_ < KR) {
                const int j = 4 * s_ + q;
                const double2 sc = *reinterpret_cast<const double2*>(&scl[2 * j]);
                const double u = accU[il][reg];
                va = fma(u * u, sc.x, va);
                const double cv = u * (c < 2 ? sc.x : sc.y);
                const double* a = vh + (j - 32 * h) * KR + 4 * c;
                const double2 lo = *reinterpret_cast<const double2*>(a);
                const double2 hi = *reinterpret_cast<const double2*>(a + 2);
                accO[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(lo.x, cv, accO[0], 0, 0, 0);
                accO[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(lo.y, cv, accO[1], 0, 0, 0);
                accO[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(hi.x, cv, accO[2], 0, 0, 0);
                accO[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(hi.y, cv, accO[3], 0, 0, 0);
              }
            }
          }
        }
      }
      PROF_MARK(6)
      // RTPS factor per variable (letkf_tools.f90:1982-1999) in the lanes of column c = 2 + v:
      // var_a = x'^T Pa x' = sum_j U_jv^2 / lam_j, var_g = sum_m x'_v[m]^2 (both still split over the 4 q groups)
      va += wshfl_xor(va, 16);
      va += wshfl_xor(va, 32);
      vg += wshfl_xor(vg, 16);
      vg += wshfl_xor(vg, 32);
      {
        const int v = c - 2;
        const bool isv = c >= 2 && c < NB;
        double cfv = 1.0;
        if (A.relax_alpha != 0.0) {
          cfv = 1.0 - A.relax_alpha;
        } else if (A.relax_alpha_spread != 0.0) {
          const double parm = (A.relax_to_inflated_prior && isv) ? A.infl[pt + A.infl_sv * (long)v] : 1.0;   // :387-391
          if (vg > 0.0 && va > 0.0) cfv = A.relax_alpha_spread * sqrt(vg * parm / (va * km1)) - A.relax_alpha_spread + 1.0;
        }
        if (A.rtps_out && q == 0 && isv && ((A.var_mask >> v) & 1u)) {   // work3da (letkf_tools.f90:460-462); skipped variables keep 1
          const bool skipv = qskip && v >= A.iv_q_first && v <= A.iv_q_last;
          A.rtps_out[pt + A.infl_sv * (long)v] = (A.relax_alpha == 0.0 && A.relax_alpha_spread != 0.0 && !skipv) ? cfv : 1.0;
        }
#pragma unroll
        for (int vv = 0; vv < NV; ++vv) cf[vv] = readlane_d(cfv, 2 + vv);
      }
      // Out tiles -> lane m holds row m: register `reg` of tile I, lane (q, c) is Out[16 reg + 4 q + I][c]
      wave_lds_sync();
      double* ob = slice + kObOff;             // [64][16] ([32][16] for KR <= 20: rows >= KR are never read), on top of the V half (behind B and the spectra when the half is smaller)
      constexpr int OBR = wave_ob_rows(KR);
#pragma unroll
      for (int I = 0; I < 4; ++I)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
          if (16 * reg < OBR) ob[(16 * reg + 4 * q + I) * 16 + c] = accO[I][reg];
      wave_lds_sync();
      {
        const double* row = ob + (OBR == 64 || lane < OBR ? lane : OBR - 1) * 16;
#pragma unroll
        for (int b = 0; b < NB; b += 2) {
          const double2 o2 = *reinterpret_cast<const double2*>(&row[b]);
          out[b] = o2.x;
          if (b + 1 < NB) out[b + 1] = o2.y;
        }
      }
      wave_lds_sync();
      }
    } else {
      // ------------------------------------------------------------ B = [r, r_det, x'_v] as bmat[m][NBP]; U = V^T B
      psync<NW>();
      if (lane < KR) {
        bmat[lane * NBP + 0] = (lane < k) ? racc : 0.0;
        bmat[lane * NBP + 1] = (lane < k) ? rdacc : 0.0;
        if (NBP > NB) bmat[lane * NBP + NB] = 0.0;
      }
      if (NV > 0) {
        const double* gp = g0 + moff;
  #pragma unroll
        for (int v = 0; v < NV; ++v) {
          const double x = (lane < k) ? *gp : 0.0;
          gp += A.sv;
          if (lane < KR) bmat[lane * NBP + 2 + v] = x;
        }
        if (lane < NV) {
          xmean[lane] = g0[k * A.sm + lane * A.sv];
          xdet[lane] = A.det_run ? g0[(k + 1) * A.sm + lane * A.sv] : 0.0;
        }
      }
      psync<NW>();
      double crow[NB];
  #pragma unroll
      for (int b = 0; b < NB; ++b) crow[b] = 0.0;
  #pragma unroll
      for (int r = 0; r < KR; ++r) {
  #pragma unroll
        for (int b = 0; b < NBP; b += 2) {
          const double2 b2 = *reinterpret_cast<const double2*>(&bmat[r * NBP + b]);   // broadcast
          crow[b] = fma(g[r], b2.x, crow[b]);
          if (b + 1 < NB) crow[b + 1] = fma(g[r], b2.y, crow[b + 1]);
        }
        if ((r & 1) == 1) pin_acc<NB>(crow);
      }
      PROF_MARK(6)
      // RTPS factor per variable (letkf_tools.f90:1982-1999), kept in SGPRs: var_a = x'^T Pa x' = sum_j U_jv^2 / lam_j
      if (NV > 0) {
  #pragma unroll
        for (int v = 0; v < NV; ++v) {
          double cfv = 1.0;
          if (A.relax_alpha != 0.0) {
            cfv = 1.0 - A.relax_alpha;
          } else if (A.relax_alpha_spread != 0.0) {
            const double parm = A.relax_to_inflated_prior ? A.infl[pt + A.infl_sv * (long)v] : 1.0;   // :387-391
            const double x = (lane < k) ? bmat[mrow_l * NBP + 2 + v] : 0.0;
            const double var_g = preduce<NW, 0>(x * x, red, rslot);
            const double var_a = preduce<NW, 0>(crow[2 + v] * crow[2 + v] * sc2, red, rslot);
            if (var_g > 0.0 && var_a > 0.0)
              cfv = A.relax_alpha_spread * sqrt(var_g * parm / (var_a * km1)) - A.relax_alpha_spread + 1.0;
          }
          cf[v] = uniform(cfv);
          if (A.rtps_out && lane == 0 && ((A.var_mask >> v) & 1u)) {   // work3da (letkf_tools.f90:460-462); skipped variables keep 1
            const bool skipv = qskip && v >= A.iv_q_first && v <= A.iv_q_last;
            A.rtps_out[pt + A.infl_sv * (long)v] = (A.relax_alpha == 0.0 && A.relax_alpha_spread != 0.0 && !skipv) ? cfv : 1.0;
          }
        }
      }
      // C = D U : w-bar spectrum 1/lam, T spectrum sqrt((k-1)/lam)
      crow[0] *= sc2;
      crow[1] *= sc2;
  #pragma unroll
      for (int v = 0; v < NV; ++v) crow[2 + v] *= sc1;

      rows_times_c<KR, NB, NW>(g, crow, out, k, vbuf, cbuf);   // lane m: out[0] = w-bar_m, out[1] = w-bar_det_m, out[2+v] = (T x'_v)_m

    }
    PROF_MARK(7)
    // ------------------------------------------------------------ analysis members (letkf_tools.f90:472-513)
    if (NV > 0 && das) {
      double* ap = a0 + moff;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const bool skip = qskip && v >= A.iv_q_first && v <= A.iv_q_last;
        double x, xm, xdt;
        if constexpr (MAPPLY) {
          // x' comes back from the LDS copy of B (it survives the output transposition): keeping the 22 registers of xv
          // alive through the matrix phase made hipcc spill LDS addresses there -- a scratch round trip in front of
          // every MFMA step of the U product (PROF build ISA)
          const double xl = slice[kBmOff + (lane < kBmRows ? lane : kBmRows - 1) * 16 + 2 + v];
          x = (lane < k) ? xl : 0.0;
          xm = readlane_d(xm_l, v);
          xdt = readlane_d(xd_l, v);
        } else {
          x = (lane < k) ? bmat[mrow_l * NBP + 2 + v] : 0.0;
          xm = xmean[v];
          xdt = xdet[v];
        }
        const double sdot = preduce<NW, 0>(x * out[0], red, rslot);
        const double sdotd = A.det_run ? preduce<NW, 0>(x * out[1], red, rslot) : 0.0;
        double val;
        if (skip) {
          val = xm + x;
        } else {
          double cdv = 0.0;
          if (A.relax_alpha != 0.0) {              // RTPP diagonal term alpha*sqrt(parm), parm read before the update
            const double parm = A.relax_to_inflated_prior ? A.infl[pt + A.infl_sv * (long)v] : 1.0;
            cdv = A.relax_alpha * sqrt(parm);
          }
          const double pert = cf[v] * out[2 + v] + cdv * x;
          val = xm + beta * (pert + sdot) + (1.0 - beta) * x;
          if (A.q_sprd_max > 0.0 && v == A.iv_q_first) {      // :500-513
            const double q_mean = preduce<NW, 0>(lane < k ? val : 0.0, red, rslot) / (double)k;
            const double dq = (lane < k) ? val - q_mean : 0.0;
            const double q_sprd = sqrt(preduce<NW, 0>(dq * dq, red, rslot) / km1) / q_mean;
            if (q_sprd > A.q_sprd_max) val = q_mean + dq * A.q_sprd_max / q_sprd;
          }
        }
        const bool inclass = (A.var_mask >> v) & 1u;
        if (lane < k && inclass) *ap = val;
        ap += A.sv;
        if (A.det_run && lane == 0 && inclass)
          a0[(k + 1) * A.sm + v * A.sv] = skip ? xdt : xdt + sdotd * beta;     // :489-497
      }
      if (A.infl_adaptive) {                       // :396-398 (also without obs: the class copies its first slot), after every parm read above
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const bool skip = qskip && v >= A.iv_q_first && v <= A.iv_q_last;
          if (!skip && lane == 0 && ((A.var_mask >> v) & 1u)) A.infl[pt + A.infl_sv * (long)v] = infl_new;
        }
      }
    } else if (A.infl_adaptive && n > 0 && lane == 0) {
      A.infl[pt] = infl_new;
    }

    // ------------------------------------------------------------ optional outputs
    if (A.transm_out && lane < k) A.transm_out[(size_t)pt * k + lane] = out[0];
    if (A.transmd_out && lane < k) A.transmd_out[(size_t)pt * k + lane] = out[1];
    if (KKOUT && (A.trans_out || A.pa_out)) {
      // T = V diag(sc1) V^T and Pa = V diag(sc2) V^T: same row-gather with C[j][:] = sc * v_j
      #pragma unroll 1
      for (int which = 0; which < 2; ++which) {
        double* dst = which == 0 ? A.trans_out : A.pa_out;
        if (!dst) continue;
        dst += (size_t)pt * k * k;
        const double sc = which == 0 ? sc1 : sc2;
        double kk[KR];
#pragma unroll
        for (int r = 0; r < KR; ++r) kk[r] = 0.0;
        double* ckk = bmat;                       // [kChunk][KR], B vectors are dead by now
        const int ncol = (k + 1) & ~1;
        for (int j0 = 0; j0 < ncol; j0 += kChunk) {
          psync<NW>();
          if (lane >= j0 && lane < j0 + kChunk) {
            const int jj = lane - j0;
#pragma unroll
            for (int r = 0; r < KR; ++r) {
              vbuf[r * kVld + jj] = g[r];
              ckk[jj * KR + r] = sc * g[r];
            }
          }
          psync<NW>();
          const int mrow = lane < KR ? lane : KR - 1;
          const int nj = min(kChunk, ncol - j0);
          for (int jj = 0; jj < nj; ++jj) {
            const double vv = vbuf[mrow * kVld + jj];
#pragma unroll
            for (int r = 0; r < KR; r += 2) {
              const double2 c2 = *reinterpret_cast<const double2*>(&ckk[jj * KR + r]);
              kk[r] = fma(vv, c2.x, kk[r]);
              kk[r + 1] = fma(vv, c2.y, kk[r + 1]);
            }
          }
        }
        psync<NW>();
        // lane m holds row m; the matrices are symmetric, so write it as column m (coalescing is irrelevant here)
        if (lane < k) {
          const double add = (which == 0 && A.add_wbar_to_trans) ? 1.0 : 0.0;
          // trans(i,j) += w-bar(i) (common_letkf.f90:221-225): row i gets w-bar_i in every column
#pragma unroll
          for (int r = 0; r < KR; ++r)
            if (r < k) dst[(size_t)r * k + lane] = kk[r] + add * out[0];
        }
      }
    }
    PROF_MARK(8)
    if (lane == 0) {
      if (A.status) A.status[pt] = st;
      if (A.nsweep) A.nsweep[pt] = sweeps;
      if (A.nobs_out) A.nobs_out[pt] = n;
    }
   }
  }
  PROF_FLUSH
}

// ------------------------------------------------------------------ host launcher
template <int KR, int NV, bool KKOUT, int NW, int FUSED = 0>
static hipError_t launch_wave(const PointArgs& a, int num_cu, hipStream_t st) {
#ifdef LETKF_LDS_PAD   // A/B knob (make VARIANT=...): extra LDS per workgroup -- what does a workgroup less per CU cost?
  const size_t lds = (size_t)(NW == 1 ? 4 : 1) * wave_slice_doubles(KR, NV, NW) * sizeof(double) + LETKF_LDS_PAD;
#else
  const size_t lds = (size_t)(NW == 1 ? 4 : 1) * wave_slice_doubles(KR, NV, NW) * sizeof(double);
#endif
  if (a.k < wave_kmin(KR, NW) || a.k > KR) return hipErrorInvalidValue;   // the Gram assumes its full member blocks
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&letkf_wave_kernel<KR, NV, KKOUT, NW, FUSED>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  // Dynamic scheduling: a grid of exactly the workgroups that are resident together (every wave owns its first unit by
  // its position, the rest is drawn; a workgroup that had to wait for a slot would sit on its first unit until the others
  // have drawn everything else).  The PROF twin's static dealing keeps round 1's oversubscribed grid.
  int grid = a.wave_grid;
  if (a.sched) {
    static int occ = 0;                        // (per instantiation)
    if (occ == 0) {
      int nb = 0;
      hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, letkf_wave_kernel<KR, NV, KKOUT, NW, FUSED>,
                                                                  NW == 1 ? 256 : 128, lds);
      occ = (e == hipSuccess && nb > 0) ? nb : (NW == 1 ? 2 : 1);
      (void)hipGetLastError();
    }
    const long res = (long)occ * num_cu;
    const long S = a.warm_stride > 1 ? a.warm_stride : 1, rl = a.run_len > 1 ? a.run_len : 1;
    const long nruns = S * ((a.npts / S + rl - 1) / rl);
    constexpr int PPW = NW == 1 ? 4 : 1;
    if (NW == 1 && nruns < 8 * res * PPW) {
      // A small batch (fewer than 8 runs per resident wave) on one-wave points: every wave gets exactly one run, by its
      // position, and the hardware starts the next workgroup when one is done.  Measured with the PROF twin on C2-mini
      // (2.25 runs per wave): of two waves on a SIMD the older one is served first and draws most of the runs, the
      // younger one is left with its last run when everything is drawn, alone on its SIMD -- 4.9 ms against 4.4 ms.
      const long need = 8 * (((nruns + 7) / 8 + PPW - 1) / PPW);
      if (grid > need) grid = (int)need;
    } else if (grid > res) {
      grid = (int)res;
    }
  }
  PointArgs b = a;
  if (a.sched) {
    sched_make_plan(b.plan, a.npts, a.warm_stride, a.run_len, grid, NW == 1 ? 4 : 1, NW == 1 ? 128 * wave_occupancy(KR, NW) : 64);   // wave-slots in flight per XCD (32 CUs)
    // the counters start at zero -- unless every unit is given out by position (a small batch): then nothing is drawn,
    // and whatever non-negative counts an earlier launch left behind read as "nothing left"
    bool draws = false;
    for (int x = 0; x < 8; ++x) draws = draws || b.plan.whole[x] + 4 * b.plan.f[x] > b.plan.nstat[x];
    if (draws) {
      hipError_t e = hipMemsetAsync(a.sched, 0, 512, st);
      if (e != hipSuccess) return e;
    }
  }
  hipLaunchKernelGGL((letkf_wave_kernel<KR, NV, KKOUT, NW, FUSED>), dim3(grid), dim3(NW == 1 ? 256 : 128), lds, st, b);
  return hipGetLastError();
}

#ifndef LETKF_WAVE_UNIT3   // (letkf_trio.hip includes this file for its device helpers and the run scheduler only)
#ifndef LETKF_WAVE_UNIT2
bool wave_kernel_supports(int k, int nv, int mode) {
  // one wave per point up to k = 62 (k + 2 augmented Gram columns in 64 lanes); two waves for 63..100 (at k >= 65 the
  // half-columns no longer fit the 256 VALU-addressable VGPRs three times over and part of them lives in AGPRs)
  if (k > 100) return false;
  if (mode == 2 || mode == 3) return nv == 11 && k <= 62;   // the fused search / the column-survivor mode are written for one wave per point
  if (mode == 0) return nv == 11;
  return nv == 0;
}

int wave_kernel_kr(int k);
static int wave_kr(int k) { return wave_kernel_kr(k); }
int wave_kernel_kr(int k) {
  return k <= 16 ? 16 : k <= 20 ? 20 : k <= 32 ? 32 : k <= 48 ? 48 : k <= 50 ? 50 : k <= 62 ? 64 : k <= 64 ? 64 : k <= 80 ? 80 : 100;
}

// Launch shape of the wave kernel: run length of the warm-started runs, grid, and the bytes of warm-start workspace
// (one [KR][64] slot per resident-or-not wave of the grid).  run_req: 0 = library default, 1 = every point cold,
// n > 1 = runs of n points.
void wave_launch_shape(int k, int mode, long npts, int num_cu, int run_req, long stride, int* run_len, int* grid,
                       size_t* ws_bytes) {
  const bool one_wave = k <= 62;
  const long ppw = one_wave ? 4 : 1;           // points in flight per workgroup
  int R = 1;
  if (mode != 1) {
    if (run_req > 0) R = run_req;
    else {
      // runs of 16 (first point of a run is a cold start), shortened until there are >= 4 workgroups per CU
      long r = npts / (ppw * num_cu * 4);
      R = (int)(r < 1 ? 1 : r > 16 ? 16 : r);
    }
    if (R > 4096) R = 4096;
  }
  if (stride < 1) stride = 1;
  const long nA = npts / stride;
  // strided runs (up a column): the whole column in one run by default -- one cold start per column, and the wave stays
  // with one set of observation rows (C2, 60 levels: 401 ms against 426 ms for runs of 30 and 457 ms for runs of 16
  // along ij)
  if (mode != 1 && run_req <= 0 && stride > 1) R = (int)(nA < 128 ? nA : 128);
  if (R > nA) R = (int)(nA > 0 ? nA : 1);      // (a run does not leave its column)
  const long nwg = (stride * ((nA + R - 1) / R) + ppw - 1) / ppw;
  long g = (long)num_cu * 16;   // 8x oversubscribed: the static block stride balances better (measured 573 ms at 2x, 541 at 16x)
#ifdef LETKF_WAVE_PROF
  if (const char* e = std::getenv("LETKF_AMD_WAVE_GRID")) g = (long)num_cu * std::atoi(e);   // PROF twin only
#endif
  *grid = (int)(nwg < g ? (nwg > 0 ? nwg : 1) : g);
  *run_len = R;
  // one [KR][lanes of a point] slot per wave-group of the grid
  *ws_bytes = (R > 1) ? (size_t)*grid * ppw * (size_t)wave_kr(k) * (one_wave ? 64 : 128) * sizeof(double) : 0;
}
#endif   // LETKF_WAVE_UNIT2

// The two-wave instantiations (63 <= k <= 100) are compiled as a unit of their own (letkf_wave2.hip = this file with
// LETKF_WAVE_UNIT2): they take hipcc's max-memory-clause scheduling strategy (k = 100 +5 %, measured A/B; the one-wave
// kernels lose 1 % with it), and the two units compile side by side.
hipError_t launch_wave_kernel_two(const PointArgs& a, int num_cu, hipStream_t st);

#define LETKF_WAVE_CASE(KR, NW)                                                                                 \
  if (k <= (NW == 1 ? (KR < 62 ? KR : 62) : KR)) {                                                              \
    if constexpr (NW == 1) {                                                                                    \
      if (a.mode == 2) return launch_wave<KR, 11, false, NW, 2>(a, num_cu, st);                                 \
      if (a.mode == 3) return launch_wave<KR, 11, false, NW, 3>(a, num_cu, st);                                 \
    }                                                                                                           \
    if (a.mode != 1)                                                                                            \
      return kkout ? launch_wave<KR, 11, true, NW>(a, num_cu, st) : launch_wave<KR, 11, false, NW>(a, num_cu, st); \
    return launch_wave<KR, 0, true, NW>(a, num_cu, st);                                                         \
  }
#ifndef LETKF_WAVE_UNIT2
hipError_t launch_wave_kernel(const PointArgs& a, int num_cu, hipStream_t st) {
  const int k = a.k;
  const bool kkout = a.trans_out || a.pa_out;
  LETKF_WAVE_CASE(16, 1)
  LETKF_WAVE_CASE(20, 1)   // MEMBER = 20: the reference's test configuration (BASELINE configs[0]); in the 32-row instantiation a third of the rows were padding
  LETKF_WAVE_CASE(32, 1)
  LETKF_WAVE_CASE(48, 1)
  LETKF_WAVE_CASE(50, 1)
  LETKF_WAVE_CASE(64, 1)
  return launch_wave_kernel_two(a, num_cu, st);
}
#else
hipError_t launch_wave_kernel_two(const PointArgs& a, int num_cu, hipStream_t st) {
  const int k = a.k;
  const bool kkout = a.trans_out || a.pa_out;
  LETKF_WAVE_CASE(64, 2)
  LETKF_WAVE_CASE(80, 2)
  LETKF_WAVE_CASE(100, 2)
  return hipErrorInvalidValue;
}
#endif
#undef LETKF_WAVE_CASE
#endif   // LETKF_WAVE_UNIT3

}  // namespace letkf
