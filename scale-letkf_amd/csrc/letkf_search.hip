// letkf_search.hip -- obs_local on the device.
//
// One wavefront per grid point.  For every merged observation-type group the wave walks the rectangle of sorting-mesh
// cells that covers the localisation cut-off (obs_local_range, scale/letkf/letkf_tools.f90:1765; ij_obsgrd_ext,
// scale/letkf/letkf_obs.f90:1209), streams the candidate table rows cell row by cell row (obs_choose_ext, :1262)
// 64 at a time, evaluates obs_local_cal (letkf_tools.f90:1793-1906) per lane, and appends the accepted ones to the
// point's CSR list with ballot + prefix-popcount, i.e. in exactly the reference's order (ctype, mesh row j, table row).
// With an observation-number limit (MAX_NOBS_PER_GRID, :1479-1729) the N best keys (distance / weight / error) are
// found by an MSB-first 8-bit radix select over the 64-bit key patterns (8 histogram passes in LDS) instead of the
// reference's recursive quick-select (common/common_sort.f90:341): same selected set up to ties, no recursion, no
// per-point scratch.  HBM-bound integer/byte work: no MFMA here by design.

#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/letkf_amd.h"
#include "letkf_device.h"
#include "letkf_divby_dev.h"
#include "letkf_search_dev.h"

namespace letkf {

namespace {

using namespace search_dev;

__device__ __forceinline__ unsigned long long key_bits(int criterion, const CalOut& c) {
  // monotone map key -> uint64 (all keys are positive doubles): smaller pattern == better candidate
  if (criterion == 1) return (unsigned long long)__double_as_longlong(c.ndist);
  if (criterion == 2) return 0x7FFFFFFFFFFFFFFFull - (unsigned long long)__double_as_longlong(c.rloc);   // largest weight first
  return (unsigned long long)__double_as_longlong(c.rdiag);
}

enum Pass { kCount = 0, kEmitAll = 1, kHist = 2, kEmitSelect = 3 };

}  // namespace

struct SelState {
  unsigned long long prefix;   // radix-select: the high bits fixed so far ...
  int shift;                   // ... occupy bits [shift + 8, 64) while bits [shift, shift + 8) are being histogrammed
  unsigned long long thresh;   // final threshold key
  int tie_budget;              // how many keys == thresh still belong to the best nmax
  int emitted;                 // entries written for this group so far
};

// One sweep over a group's candidates; what happens to an accepted candidate depends on PASS.
template <int PASS>
__device__ __forceinline__ int sweep_group(const SearchArgs& A, const int gs, const int ge, const double ri,
                                           const double rj, const double rlev, const double rz, const long out,
                                           SelState& st, unsigned int* hist) {
  const letkf_search_tables& t = A.t;
  const int lane = threadIdx.x & 63;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  int cnt = 0;
  for (int m = gs; m < ge; ++m) {
    const int ic = t.group_member[m];
    const double dzi = t.hori_loc[ic] * kDistZeroFac / t.dx;        // obs_local_range :1775-1778
    const double dzj = t.hori_loc[ic] * kDistZeroFac / t.dy;
    int imin, imax, jmin, jmax;
    ij_obsgrd_ext(t, ic, ri - dzi, rj - dzj, imin, jmin);
    ij_obsgrd_ext(t, ic, ri + dzi, rj + dzj, imax, jmax);
    // the reference requires the extended mesh to cover the rectangle (DEBUG check :1780); clamp defensively
    imin = max(imin, 1);
    jmin = max(jmin, 1);
    imax = min(imax, t.ngrdext_i[ic]);
    jmax = min(jmax, t.ngrdext_j[ic]);
    if (imin > imax || jmin > jmax) continue;
    const long acb = t.ac_off[ic];
    const int ld = t.ngrdext_i[ic] + 1;
    for (int j = jmin; j <= jmax; ++j) {
      const int lo = t.ac_ext[acb + (imin - 1) + (long)ld * (j - 1)];
      const int hi = t.ac_ext[acb + imax + (long)ld * (j - 1)];
      for (int base = lo; base < hi; base += 64) {
        const int row = base + lane;
        CalOut c{0.0, -1.0, -1.0};
        if (row < hi) c = local_cal(t, ic, ri, rj, rlev, rz, row);
        bool acc = c.rloc != 0.0;                        // :1460
        unsigned long long key = 0;
        if (PASS >= kHist && acc) key = key_bits(t.criterion, c);
        if (PASS == kHist) {
          if (acc && (st.shift == 56 || (key >> (st.shift + 8)) == st.prefix))
            atomicAdd(&hist[(unsigned int)(key >> st.shift) & 0xFFu], 1u);
          continue;
        }
        bool tie = false;
        if (PASS == kEmitSelect) {
          tie = acc && key == st.thresh;
          acc = acc && key < st.thresh;
        }
        const unsigned long long mk = __ballot(acc);
        const int pos = __popcll(mk & lt_mask);
        if (PASS != kCount && acc) {
          const long o = out + st.emitted + pos;
          A.obs_idx[o] = row;
          A.rdiag_l[o] = c.rdiag;
          A.rloc_l[o] = c.rloc;
        }
        const int na = __popcll(mk);
        cnt += na;
        if (PASS != kCount) st.emitted += na;
        if (PASS == kEmitSelect) {                       // ties at the threshold: first come, first served
          const unsigned long long tk = __ballot(tie);
          const int tpos = __popcll(tk & lt_mask);
          if (tie && tpos < st.tie_budget) {
            const long o = out + st.emitted + tpos;
            A.obs_idx[o] = row;
            A.rdiag_l[o] = c.rdiag;
            A.rloc_l[o] = c.rloc;
          }
          const int nt = min(__popcll(tk), st.tie_budget);
          st.tie_budget -= nt;
          st.emitted += nt;
          cnt += nt;
        }
      }
    }
  }
  return cnt;
}

// ---------------------------------------------------------------------------------------------
// Limited mode, fast path: ONE sweep evaluates obs_local_cal for the group's candidates and keeps the accepted ones
// (key, row, rdiag, rloc: 32 B) in LDS; the radix select and the emission then run on that cache.  The first version
// re-swept all candidates for each of the 8 histogram rounds and once more to emit (10 evaluations of obs_local_cal
// per candidate: 634 ms for C2 with two ctypes limited to 100, more than the solve).  A group whose accepted
// candidates do not fit the cache (kCacheCap) falls back to that multi-sweep path.
// ---------------------------------------------------------------------------------------------
// inclusive prefix sum over the 64 lanes on the vector ALU (DPP row shifts + row broadcasts: the sequence LLVM's
// atomic optimiser emits for gfx9), instead of six ds_bpermute round trips
__device__ __forceinline__ int wave_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2 and 3
  return v;
}

// In which of the 256 histogram bins does the `want`-th smallest key (1-based) fall, and how many keys lie in the bins
// before it?  Lane l owns bins 4l..4l+3; a wave prefix sum over the lane totals finds the lane, the lane its bin.
// (The first version walked the 256 bins one dependent LDS read at a time: 8 rounds x 256 reads per group dominated
// the limited-mode search.)
__device__ __forceinline__ void find_bin(const unsigned int* hist, const int want, int& bsel, int& before) {
  const int lane = threadIdx.x & 63;
  const uint4 h = *reinterpret_cast<const uint4*>(&hist[4 * lane]);
  const int s = (int)(h.x + h.y + h.z + h.w);
  const int incl = wave_incl_scan(s);            // DPP, not six ds_bpermute round trips
  const int excl = incl - s;
  const bool here = excl < want && incl >= want;
  int b = 4 * lane, cum = excl;
  if (cum + (int)h.x < want) {
    cum += (int)h.x;
    ++b;
    if (cum + (int)h.y < want) {
      cum += (int)h.y;
      ++b;
      if (cum + (int)h.z < want) {
        cum += (int)h.z;
        ++b;
      }
    }
  }
  const unsigned long long mk = __ballot(here);
  if (mk == 0ull) {            // want exceeds the population (cannot happen for want <= count); keep the old default
    bsel = 255;
    before = __builtin_amdgcn_readlane(incl, 63) - __builtin_amdgcn_readfirstlane((int)hist[255]);
    return;
  }
  const int src = __ffsll((long long)mk) - 1;
  bsel = __builtin_amdgcn_readlane(b, src);
  before = __builtin_amdgcn_readlane(cum, src);
}

constexpr int kCacheCap = 512;

// returns the number of cached (= accepted) candidates, or -1 on overflow
__device__ __forceinline__ int cache_group(const SearchArgs& A, const int gs, const int ge, const double ri,
                                           const double rj, const double rlev, const double rz, double* cache) {
  const letkf_search_tables& t = A.t;
  const int lane = threadIdx.x & 63;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  int cnt = 0;
  for (int m = gs; m < ge; ++m) {
    const int ic = t.group_member[m];
    const double dzi = t.hori_loc[ic] * kDistZeroFac / t.dx;
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
    for (int j = jmin; j <= jmax; ++j) {
      const int lo = t.ac_ext[acb + (imin - 1) + (long)ld * (j - 1)];
      const int hi = t.ac_ext[acb + imax + (long)ld * (j - 1)];
      for (int base = lo; base < hi; base += 64) {
        const int row = base + lane;
        CalOut c{0.0, -1.0, -1.0};
        if (row < hi) c = local_cal(t, ic, ri, rj, rlev, rz, row);
        const bool acc = c.rloc != 0.0;
        const unsigned long long mk = __ballot(acc);
        const int na = __popcll(mk);
        if (cnt + na > kCacheCap) return -1;              // wave-uniform
        if (acc) {
          const int e = cnt + __popcll(mk & lt_mask);
          *reinterpret_cast<double2*>(&cache[4 * e]) =
              double2{__longlong_as_double((long long)key_bits(t.criterion, c)), __longlong_as_double((long long)row)};
          *reinterpret_cast<double2*>(&cache[4 * e + 2]) = double2{c.rdiag, c.rloc};
        }
        cnt += na;
      }
    }
  }
  return cnt;
}

// emits the nmax best of the nc cached candidates (all of them if nc <= nmax) in candidate order; returns the number emitted
__device__ __forceinline__ int select_from_cache(const SearchArgs& A, const int nc, const int nmax, const long out,
                                                 const double* cache, unsigned int* hist) {
  const int lane = threadIdx.x & 63;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  unsigned long long thresh = ~0ull;
  int tie_budget = 0;
  if (nc > nmax) {
    unsigned long long prefix = 0ull;
    int want = nmax;
    for (int round = 0; round < 8; ++round) {
      for (int b = lane; b < 256; b += 64) hist[b] = 0;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int shift = 56 - 8 * round;
      for (int e = lane; e < nc; e += 64) {
        const unsigned long long key = (unsigned long long)__double_as_longlong(cache[4 * e]);
        if (round == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&hist[(unsigned int)(key >> shift) & 0xFFu], 1u);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      int cum = 0, bsel = 255;
      find_bin(hist, want, bsel, cum);
      want -= cum;
      prefix = (round == 0) ? (unsigned long long)bsel : ((prefix << 8) | (unsigned long long)bsel);
    }
    thresh = prefix;
    tie_budget = want;
  }
  int emitted = 0;
  for (int e0 = 0; e0 < nc; e0 += 64) {
    const int e = e0 + lane;
    bool acc = false, tie = false;
    double2 a2{0.0, 0.0}, b2{0.0, 0.0};
    if (e < nc) {
      a2 = *reinterpret_cast<const double2*>(&cache[4 * e]);
      b2 = *reinterpret_cast<const double2*>(&cache[4 * e + 2]);
      const unsigned long long key = (unsigned long long)__double_as_longlong(a2.x);
      acc = nc <= nmax || key < thresh;
      tie = nc > nmax && key == thresh;
    }
    const unsigned long long mk = __ballot(acc);
    if (acc) {
      const long o = out + emitted + __popcll(mk & lt_mask);
      A.obs_idx[o] = (int)__double_as_longlong(a2.y);
      A.rdiag_l[o] = b2.x;
      A.rloc_l[o] = b2.y;
    }
    emitted += __popcll(mk);
    const unsigned long long tk = __ballot(tie);                  // ties at the threshold: first come, first served
    const int tpos = __popcll(tk & lt_mask);
    if (tie && tpos < tie_budget) {
      const long o = out + emitted + tpos;
      A.obs_idx[o] = (int)__double_as_longlong(a2.y);
      A.rdiag_l[o] = b2.x;
      A.rloc_l[o] = b2.y;
    }
    const int nt = min(__popcll(tk), tie_budget);
    tie_budget -= nt;
    emitted += nt;
  }
  return emitted;
}

__global__ void __launch_bounds__(256) letkf_search_kernel(const SearchArgs A) {
  __shared__ unsigned int hist_all[4][256];
  extern __shared__ __attribute__((aligned(16))) double cache_all[];       // [4][kCacheCap][4]
  const letkf_search_tables& t = A.t;
  const int lane = threadIdx.x & 63;
  // (scalar: one column / point per WAVE -- derived from threadIdx alone hipcc takes everything that hangs on it, the
  // mesh walk, the survivor counts, the select's loop, for divergent and wraps it in exec-mask loops)
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned int* hist = hist_all[wv];
  double* cache = cache_all + (size_t)wv * 4 * kCacheCap;

  for (long pt = (long)blockIdx.x * 4 + wv; pt < A.npts; pt += (long)gridDim.x * 4) {
    const double ri = A.ri[pt], rj = A.rj[pt], rlev = A.rlev[pt], rz = A.rz[pt];
    long out = A.fill ? A.obs_off[pt] : 0;
    int total = 0;

    for (int ig = 0; ig < t.ngroup; ++ig) {
      const int gs = t.group_start[ig], ge = t.group_start[ig + 1];
      const int nmax = t.max_nobs[t.group_member[gs]];        // the master's limit rules the group (:1434)
      SelState st{0ull, 56, 0ull, 0, 0};
      int ngrp = 0;
      if (nmax <= 0) {                                         // no limit: :1438-1476
        ngrp = A.fill ? sweep_group<kEmitAll>(A, gs, ge, ri, rj, rlev, rz, out, st, hist)
                      : sweep_group<kCount>(A, gs, ge, ri, rj, rlev, rz, out, st, hist);
      } else if (A.fill && (ngrp = cache_group(A, gs, ge, ri, rj, rlev, rz, cache)) >= 0) {
        ngrp = select_from_cache(A, ngrp, nmax, out, cache, hist);   // one sweep; select and emit from LDS
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      } else {
        const int cnt = sweep_group<kCount>(A, gs, ge, ri, rj, rlev, rz, out, st, hist);
        if (cnt <= nmax) {
          ngrp = cnt;
          if (A.fill) sweep_group<kEmitAll>(A, gs, ge, ri, rj, rlev, rz, out, st, hist);
        } else if (!A.fill) {
          ngrp = nmax;                                         // :1616 / :1703
        } else {
          // MSB-first radix select of the nmax-th smallest key
          int want = nmax;                                     // rank (1-based) inside the current prefix class
          for (int round = 0; round < 8; ++round) {
            for (int b = lane; b < 256; b += 64) hist[b] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            st.shift = 56 - 8 * round;
            sweep_group<kHist>(A, gs, ge, ri, rj, rlev, rz, out, st, hist);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            int cum = 0, bsel = 255;
            find_bin(hist, want, bsel, cum);
            want -= cum;
            st.prefix = (round == 0) ? (unsigned long long)bsel : ((st.prefix << 8) | (unsigned long long)bsel);
          }
          st.thresh = st.prefix;
          st.tie_budget = want;
          st.emitted = 0;
          ngrp = sweep_group<kEmitSelect>(A, gs, ge, ri, rj, rlev, rz, out, st, hist);
        }
      }
      total += ngrp;
      if (A.fill) out += ngrp;
    }
    if (!A.fill && lane == 0) A.counts[pt] = total;
  }
}

// ---------------------------------------------------------------------------------------------
// Column-cooperative search (no-limit mode): one wavefront per horizontal point ij does ALL its levels.
// The horizontal part of obs_local_cal (two subtractions, a square root, a division, the cut-off test) depends only
// on (ij, observation), so it is evaluated once per column and observation instead of once per level (60 times at
// C2); the survivors wait in LDS with everything the vertical part needs (row, nd_h, the vertical coordinate already
// through its log where the type calls for one, the error) and every level then runs only
//   nd_v = |v_obs - v_point| / vert_loc,  nd = nd_h^2 + nd_v^2,  rloc = varloc exp(-nd/2),  rdiag = err^2 / rloc
// over them.  Same expressions in the same order as local_cal_v (compiled without fusion), same candidate order, so
// the lists are identical to the per-point kernel's, entry for entry.
// ---------------------------------------------------------------------------------------------
constexpr int kSurv = 512;                  // survivors buffered per wave (32 B each)

struct ColArgs {
  letkf_search_tables t;
  long nij1;
  int nlev;
  const double* rig;
  const double* rjg;
  const double* rlev;    // [nij1 * nlev], point p = ij + nij1 * lev
  const double* rz;
  int fill;
  int* counts;
  const long* obs_off;
  int* obs_idx;
  double* rdiag_l;
  double* rloc_l;
  int* nobs_ctype;       // [npts][nctype] accepted rows per combined type (nobsl_t of obs_local), or null
};

// FILL = false is the counting pass of the two-phase CSR build: no weights are needed there, only whether a row is accepted.
template <bool FILL>
__global__ void __launch_bounds__(256) letkf_search_columns_kernel(const ColArgs A) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) double smem_col[];
  const letkf_search_tables& t = A.t;
  const int lane = threadIdx.x & 63;
  // (scalar: one column / point per WAVE -- derived from threadIdx alone hipcc takes everything that hangs on it, the
  // mesh walk, the survivor counts, the select's loop, for divergent and wraps it in exec-mask loops)
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nlev = A.nlev;
  const int cstride = 4 * kSurv + 2 * ((nlev + 1) & ~1);         // doubles per wave: survivors + 2 level counter arrays
  double* sb = smem_col + (size_t)wv * cstride;                  // [kSurv][4]: row bits, nd_h, v_obs, err
  int* cntl = reinterpret_cast<int*>(sb + 4 * kSurv);            // [nlev] entries emitted so far per level
  int* cprev = cntl + 2 * ((nlev + 1) & ~1);                     // [nlev] the same at the start of the current ctype
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

  for (long col = (long)blockIdx.x * 4 + wv; col < A.nij1; col += (long)gridDim.x * 4) {
    const double ri = A.rig[col], rj = A.rjg[col];
    for (int l = lane; l < nlev; l += 64) cntl[l] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    for (int m = 0; m < t.group_start[t.ngroup]; ++m) {
      const int ic = t.group_member[m];
      const double varloc = t.varloc[ic];
      if (varloc < kTiny) continue;                               // local_cal :1843
      const int vm = t.vmode[ic];
      const bool wsafe = varloc > 1e-290;
      const double vloc = t.vert_loc[ic], hloc = t.hori_loc[ic];
      const double dzi = hloc * kDistZeroFac / t.dx, dzj = hloc * kDistZeroFac / t.dy;
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
      int ns = 0;                                                 // survivors in the buffer
      if (A.nobs_ctype)
        for (int l = lane; l < nlev; l += 64) cprev[l] = cntl[l];

      // every level against the buffered survivors, then the buffer is empty again
      auto vertical = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int lev = 0; lev < nlev; ++lev) {
          const long p = col + A.nij1 * (long)lev;
          double vref = 0.0;
          if (vloc != 0.0) {
            if (vm == 1) vref = A.rz[p];
            else vref = log(A.rlev[p]);
          }
          const double vconst = (vm == 3 && vloc != 0.0) ? fabs(log(t.rain_base) - vref) / vloc : 0.0;
          int emitted = cntl[lev];
          const long out = FILL ? A.obs_off[p] : 0;
          for (int s0 = 0; s0 < ns; s0 += 64) {
            const int si = s0 + lane;
            bool acc = false;
            double rloc = 0.0, rdiag = 0.0;
            int row = 0;
            if (si < ns) {
              const double2 a2 = *reinterpret_cast<const double2*>(&sb[4 * si]);
              const double2 b2 = *reinterpret_cast<const double2*>(&sb[4 * si + 2]);
              row = (int)__double_as_longlong(a2.x);
              const double nd_h = a2.y;
              double nd_v;
              if (vloc == 0.0) nd_v = 0.0;                        // :1851-1865
              else if (vm == 3) nd_v = vconst;
              else nd_v = fabs(b2.x - vref) / vloc;
              if (!(nd_v > kDistZeroFac)) {                       // :1869
                const double nd = nd_h * nd_h + nd_v * nd_v;      // :1888
                if (!(nd > kDistZeroFacSq)) {                     // :1891
                  if (FILL || !wsafe) {
                    rloc = varloc * exp(-0.5 * nd);               // :1899
                    rdiag = b2.y * b2.y / rloc;                   // :1903
                    acc = rloc != 0.0;                            // letkf_tools.f90:1460
                  } else {
                    // counting pass: inside the cut-off nd <= 13.3, exp(-nd / 2) >= 1.2e-3, so the weight cannot vanish
                    // unless the variable localisation is itself at the bottom of the exponent range (wsafe): the
                    // exponential and the division -- half of this loop's instructions -- are the fill pass's alone
                    acc = true;
                  }
                }
              }
            }
            const unsigned long long mk = __ballot(acc);
            if (FILL && acc) {
              const long o = out + emitted + __popcll(mk & lt_mask);
              A.obs_idx[o] = row;
              A.rdiag_l[o] = rdiag;
              A.rloc_l[o] = rloc;
            }
            emitted += __popcll(mk);
          }
          if (lane == 0) cntl[lev] = emitted;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        ns = 0;
      };

      for (int j = jmin; j <= jmax; ++j) {
        const int lo = t.ac_ext[acb + (imin - 1) + (long)ld * (j - 1)];
        const int hi = t.ac_ext[acb + imax + (long)ld * (j - 1)];
        for (int base = lo; base < hi; base += 64) {
          const int row = base + lane;
          bool ok = false;
          double nd_h = 0.0, vobs = 0.0, err = 0.0;
          if (row < hi) {
            const double rdx = (ri - t.ob_ri[row]) * t.dx;        // :1876-1878
            const double rdy = (rj - t.ob_rj[row]) * t.dy;
            nd_h = sqrt(rdx * rdx + rdy * rdy) / hloc;
            ok = !(nd_h > kDistZeroFac);                          // :1881
            if (ok) {
              if (vloc != 0.0) {
                if (vm == 1) vobs = t.ob_lev[row];
                else if (vm == 2) vobs = log(t.ob_dat[row]);
                else if (vm != 3) vobs = log(t.ob_lev[row]);
              }
              err = t.ob_err[row];
            }
          }
          const unsigned long long mk = __ballot(ok);
          if (ok) {
            const int si = ns + __popcll(mk & lt_mask);
            *reinterpret_cast<double2*>(&sb[4 * si]) = double2{__longlong_as_double((long)row), nd_h};
            *reinterpret_cast<double2*>(&sb[4 * si + 2]) = double2{vobs, err};
          }
          ns += __popcll(mk);
          if (ns > kSurv - 64) vertical();
        }
      }
      if (ns > 0) vertical();
      if (A.nobs_ctype)                                           // nobsl_t (letkf_tools.f90:1473-1475)
        for (int l = lane; l < nlev; l += 64)
          A.nobs_ctype[(col + A.nij1 * (long)l) * t.nctype + ic] = cntl[l] - cprev[l];
    }
    if (!FILL)
      for (int l = lane; l < nlev; l += 64) A.counts[col + A.nij1 * (long)l] = cntl[l];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}


// ---------------------------------------------------------------------------------------------
// Column-cooperative search WITH an observation-number limit (MAX_NOBS_PER_GRID > 0, letkf_tools.f90:1479-1729; the
// production setting of the reference's radar configurations).  Per merged group (:1434-1436: the master's limit
// rules the group) the horizontal survivors of ALL its members wait in LDS -- once per column, as above -- and every
// level then (1) runs the vertical part over them and leaves each candidate's selection key in LDS, (2) if more than
// nmax were accepted finds the nmax-th smallest key with the MSB-first radix select of the per-point kernel, on the
// LDS keys, (3) walks the survivors once more in candidate order and emits the selected ones, re-evaluating the two
// or three expressions of the vertical part for those (the same device function as in step 1: bit-identical).
// The selected SET equals the per-point kernel's (and the reference's, up to ties); groups without a limit emit in
// step 1.  A group whose survivors do not fit the buffer falls back to the per-point multi-sweep for that column.
// Also fills the NOBS_OUT inputs (letkf_tools.f90:440-447): nobsl_t per combined type and the cut-off measure
// cutd_t of :1604-1660 / :1716-1727 (criterion 1: hori_loc * sqrt(largest selected distance) once the limit is hit,
// else hori_loc * dist_zero_fac; criterion 2 / 3: the smallest selected weight / largest selected error, else 0).
// ---------------------------------------------------------------------------------------------
constexpr int kSurvL = 576;                 // survivors of one group buffered per wave (32 B each): 18 KB, 8 waves per CU
constexpr int kKeyS = kSurvL / 64;          // ... their selection keys: this many per lane, in registers
constexpr unsigned long long kNoKey = ~0ull;

struct VertOut {
  double nd, rloc, rdiag;
  bool acc;
};
__device__ __forceinline__ VertOut vertical_cal(const int vm, const double vloc, const double varloc, const double vconst,
                                                const double vref, const double nd_h, const double vobs, const double err) {
#pragma clang fp contract(off)
  VertOut o{0.0, 0.0, 0.0, false};
  double nd_v;
  if (vloc == 0.0) nd_v = 0.0;                        // :1851-1865
  else if (vm == 3) nd_v = vconst;
  else nd_v = fabs(vobs - vref) / vloc;
  if (nd_v > kDistZeroFac) return o;                  // :1869
  const double nd = nd_h * nd_h + nd_v * nd_v;        // :1888
  if (nd > kDistZeroFacSq) return o;                  // :1891
  o.nd = nd;
  o.rloc = varloc * exp(-0.5 * nd);                   // :1899
  o.rdiag = err * err / o.rloc;                       // :1903
  o.acc = o.rloc != 0.0;                              // letkf_tools.f90:1460
  return o;
}

// the cut-off tests alone (distance criterion: the selection key is nd itself, and rloc = varloc exp(-nd/2) cannot
// be zero inside the cut-off unless varloc is denormal-small, so the exponential and the error division are only
// needed for the observations that end up selected)
__device__ __forceinline__ VertOut vertical_nd(const int vm, const double vloc, const double varloc, const double vconst,
                                               const double vref, const double nd_h, const double vobs, const double err) {
#pragma clang fp contract(off)
  if (varloc < 1e-290) return vertical_cal(vm, vloc, varloc, vconst, vref, nd_h, vobs, err);
  VertOut o{0.0, 0.0, 0.0, false};
  double nd_v;
  if (vloc == 0.0) nd_v = 0.0;
  else if (vm == 3) nd_v = vconst;
  else nd_v = fabs(vobs - vref) / vloc;
  if (nd_v > kDistZeroFac) return o;
  const double nd = nd_h * nd_h + nd_v * nd_v;
  if (nd > kDistZeroFacSq) return o;
  o.nd = nd;
  o.acc = true;
  return o;
}

// wave-wide minimum / maximum of one unsigned word per lane on the vector ALU (the DPP sequence of wave_incl_scan with
// the operation's identity shifted in at the row ends; lane 63 ends up with the result)
__device__ __forceinline__ unsigned int wave_min_u32(unsigned int v) {
  v = min(v, (unsigned int)__builtin_amdgcn_update_dpp(-1, (int)v, 0x111, 0xF, 0xF, false));
  v = min(v, (unsigned int)__builtin_amdgcn_update_dpp(-1, (int)v, 0x112, 0xF, 0xF, false));
  v = min(v, (unsigned int)__builtin_amdgcn_update_dpp(-1, (int)v, 0x114, 0xF, 0xF, false));
  v = min(v, (unsigned int)__builtin_amdgcn_update_dpp(-1, (int)v, 0x118, 0xF, 0xF, false));
  v = min(v, (unsigned int)__builtin_amdgcn_update_dpp(-1, (int)v, 0x142, 0xA, 0xF, false));
  v = min(v, (unsigned int)__builtin_amdgcn_update_dpp(-1, (int)v, 0x143, 0xC, 0xF, false));
  return (unsigned int)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned int wave_max_u32(unsigned int v) {
  v = max(v, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false));
  v = max(v, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false));
  v = max(v, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false));
  v = max(v, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false));
  v = max(v, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));
  v = max(v, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false));
  return (unsigned int)__builtin_amdgcn_readlane((int)v, 63);
}
// a wave-uniform value, said so: into scalar registers (what hipcc loads through the tables' plain pointers it keeps in
// vector registers -- and spills to scratch inside the level loop)
__device__ __forceinline__ int uni(const int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ double uni(const double x) {
  const long long b = __double_as_longlong(x);
  const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)b);
  const unsigned int hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// number of set bits of a wave mask below this lane (v_mbcnt_lo / _hi: two instructions)
__device__ __forceinline__ int mbcnt(const unsigned long long m) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
}
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// want-th smallest (1-based, want <= nreal) of the wave's keys held in REGISTERS (slot u of lane l = candidate 64 u + l;
// kNoKey = none; every real key < 2^63), and how many of the keys equal to it belong to the selection.
//   rounds: the class [lo, hi) that holds the wanted key is cut into <= 256 equal key ranges, the class members are
//     histogrammed with LDS atomics (one per candidate) and find_bin's wave scan names the range holding the wanted key --
//     2-3 dependent wave steps per round; the first class spans the high words actually present (wave min / max), so a
//     round tells ~8 bits, and ~380 distance keys are down to a handful after one or two rounds;
//   finish: a class of <= 64 is laid out one key per lane and every lane counts the class keys below / not above its own
//     (broadcast LDS reads): the wanted key is the one whose two counts straddle `want`.
// (The first version searched bit by bit, counting with a ballot per slot and bit: ~20 steps of 9 dependent
// VALU -> SALU round trips, 28 k cycles per select at the 2 waves per SIMD this kernel runs at -- 57 % of the fill pass.)
template <int KS>
__device__ __forceinline__ void hist_thresh(const unsigned long long (&key_in)[KS], const int ns, const int nreal, int want,
                                            unsigned int* hist, unsigned long long& thresh, int& tie_budget,
                                            const double lin_scale = 0.0) {
  const int lane = threadIdx.x & 63;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  unsigned long long key[KS];
#pragma unroll
  for (int u = 0; u < KS; ++u) key[u] = key_in[u];
  int inclass = nreal;
  if (lin_scale > 0.0 && inclass > 64) {
    // first round by VALUE: the keys are doubles in [0, 256 / lin_scale) (squared distances up to the cut-off), and
    // floor(key * lin_scale) is a monotone map onto 256 ranges of equal width -- no look at the keys' spread is needed
    // and a class of ~380 is down to a few; whoever is not in the wanted range drops out of the wave's copy of the keys
    *reinterpret_cast<uint4*>(&hist[4 * lane]) = uint4{0u, 0u, 0u, 0u};
    wave_lds_sync();
    int dig[KS];
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      dig[u] = min((int)(__longlong_as_double((long long)key[u]) * lin_scale), 255);
      if (u * 64 < ns && key[u] != kNoKey) atomicAdd(&hist[dig[u]], 1u);
    }
    wave_lds_sync();
    int bsel = 0, before = 0;
    find_bin(hist, want, bsel, before);
    const int cb = (int)hist[bsel];
    wave_lds_sync();
    want -= before;
    inclass = __builtin_amdgcn_readfirstlane(cb);
#pragma unroll
    for (int u = 0; u < KS; ++u)
      if (dig[u] != bsel) key[u] = kNoKey;
  }
  unsigned long long lo = 0ull, hi = 1ull << 63;                 // (every real key)
  if (inclass > 64) {
    unsigned int mn = ~0u, mx = 0u;
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      const unsigned int h = (unsigned int)(key[u] >> 32);
      mn = min(mn, h);                                           // (kNoKey's high word is the identity of the minimum)
      mx = max(mx, key[u] != kNoKey ? h : 0u);
    }
    lo = (unsigned long long)wave_min_u32(mn) << 32;
    hi = ((unsigned long long)wave_max_u32(mx) + 1ull) << 32;
  }
  while (inclass > 64) {
    const unsigned long long span1 = hi - lo - 1ull;
    if (span1 == 0ull) {                                         // one key value, more than 64 times
      thresh = lo;
      tie_budget = want;
      return;
    }
    const int w = 64 - __clzll((long long)span1);
    const int sh = w > 8 ? w - 8 : 0;
    *reinterpret_cast<uint4*>(&hist[4 * lane]) = uint4{0u, 0u, 0u, 0u};
    wave_lds_sync();
#pragma unroll
    for (int u = 0; u < KS; ++u)
      if (u * 64 < ns && key[u] >= lo && key[u] < hi) atomicAdd(&hist[(unsigned int)((key[u] - lo) >> sh)], 1u);
    wave_lds_sync();
    int bsel = 0, before = 0;
    find_bin(hist, want, bsel, before);
    const int cb = (int)hist[bsel];
    wave_lds_sync();
    want -= before;
    inclass = __builtin_amdgcn_readfirstlane(cb);
    lo += (unsigned long long)bsel << sh;
    const unsigned long long top = lo + (1ull << sh);
    hi = top < hi ? top : hi;
  }
  // finish: the class, one key per lane
  unsigned long long* cbuf = reinterpret_cast<unsigned long long*>(hist);
  int base = 0;
#pragma unroll
  for (int u = 0; u < KS; ++u)
    if (u * 64 < ns) {
      const bool in = key[u] >= lo && key[u] < hi;
      const unsigned long long m = __ballot(in);
      if (in) cbuf[base + __popcll(m & lt_mask)] = key[u];
      base += __popcll(m);
    }
  wave_lds_sync();
  const unsigned long long mine = lane < inclass ? cbuf[lane] : kNoKey;
  int r_lt = 0, r_le = 0;
  for (int j = 0; j < inclass; ++j) {
    const unsigned long long kj = cbuf[j];                      // (one address: a broadcast read)
    r_lt += kj < mine;
    r_le += kj <= mine;
  }
  const unsigned long long hit = __ballot(lane < inclass && r_lt < want && r_le >= want);
  const int src = __ffsll((long long)hit) - 1;                  // (hit != 0: want <= inclass)
  thresh = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(mine >> 32), src) << 32) |
           (unsigned int)__builtin_amdgcn_readlane((int)mine, src);
  tie_budget = want - __builtin_amdgcn_readlane(r_lt, src);
  wave_lds_sync();
}

#ifdef LETKF_WAVE_PROF
__device__ unsigned long long g_lim_prof[8];
#define LP_T() __builtin_readcyclecounter()
#define LP_ADD(i, v) lpa[i] += (unsigned long long)(v)
#define LP_DECL() unsigned long long lpa[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define LP_FLUSH() do { if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&g_lim_prof[i], lpa[i]); } while (0)
#else
#define LP_DECL() do {} while (0)
#define LP_FLUSH() do {} while (0)
#define LP_T() 0ull
#define LP_ADD(i, v) do {} while (0)
#endif
struct ColLimArgs {
  ColArgs c;
  double* cutd_ctype;    // [npts][nctype] or null
};

__global__ void __launch_bounds__(256, 2) letkf_search_columns_limited_kernel(const ColLimArgs L) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) double smem_lim[];
  __shared__ __attribute__((aligned(16))) unsigned int hist_all[4][kSurvL / 2];   // 256 bins / the emission list
  const ColArgs& A = L.c;
  const letkf_search_tables& t = A.t;
  const int lane = threadIdx.x & 63;
  // (scalar: one column / point per WAVE -- derived from threadIdx alone hipcc takes everything that hangs on it, the
  // mesh walk, the survivor counts, the select's loop, for divergent and wraps it in exec-mask loops)
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nlev = A.nlev;
  const int nl2 = (nlev + 1) & ~1;
  const int cstride = 4 * kSurvL + nl2;                           // doubles per wave
  double* sb = smem_lim + (size_t)wv * cstride;                   // [kSurvL][4]: nd_h, v_obs, row, err
  int* cntl = reinterpret_cast<int*>(sb + 4 * kSurvL);            // [nlev] entries emitted so far per level
  unsigned int* hist = hist_all[wv];
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  SearchArgs PA;                                                  // for the per-point fall-back
  PA.t = t;
  PA.fill = A.fill;
  PA.obs_idx = A.obs_idx;
  PA.rdiag_l = A.rdiag_l;
  PA.rloc_l = A.rloc_l;

  LP_DECL();
  for (long col = (long)blockIdx.x * 4 + wv; col < A.nij1; col += (long)gridDim.x * 4) {
    const double ri = A.rig[col], rj = A.rjg[col];
    for (int l = lane; l < nlev; l += 64) cntl[l] = 0;
    if (A.nobs_ctype || L.cutd_ctype)                             // defaults: letkf_tools.f90:1380-1391, :1427-1432
      for (int e = lane; e < nlev * t.nctype; e += 64) {
        const int l = e / t.nctype, ic = e - l * t.nctype;
        const long p = col + A.nij1 * (long)l;
        if (A.nobs_ctype) A.nobs_ctype[p * t.nctype + ic] = 0;
        if (L.cutd_ctype) L.cutd_ctype[p * t.nctype + ic] = 0.0;
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    for (int ig = 0; ig < t.ngroup; ++ig) {
      const int gs = t.group_start[ig], ge = t.group_start[ig + 1];
      const int icm = t.group_member[gs];                         // master
      const int nmax = t.max_nobs[icm];
      // ---- horizontal part, all members of the group.  Every member starts a fresh 64-entry slot of the buffer, so
      // that one slot holds ONE member's survivors and its vertical parameters are wave-uniform in the level loop.
      [[maybe_unused]] const unsigned long long lp0 = LP_T();
      int ns = 0;
      bool overflow = ge - gs > 4;                                // (not expected: the reference merges two types)
      int mbeg[4] = {INT_MAX, INT_MAX, INT_MAX, INT_MAX}, mend[4] = {0, 0, 0, 0};
      for (int m = gs; m < ge && !overflow; ++m) {
        const int ic = t.group_member[m];
        ns = (ns + 63) & ~63;
        const int ns0 = ns;
        const bool skip = t.varloc[ic] < kTiny;                   // local_cal :1843
        const int vm = t.vmode[ic];
        const double vloc = t.vert_loc[ic], hloc = t.hori_loc[ic];
        const double dzi = hloc * kDistZeroFac / t.dx, dzj = hloc * kDistZeroFac / t.dy;
        int imin, imax, jmin, jmax;
        ij_obsgrd_ext(t, ic, ri - dzi, rj - dzj, imin, jmin);
        ij_obsgrd_ext(t, ic, ri + dzi, rj + dzj, imax, jmax);
        imin = max(imin, 1);
        jmin = max(jmin, 1);
        imax = min(imax, t.ngrdext_i[ic]);
        jmax = min(jmax, t.ngrdext_j[ic]);
        if (skip || imin > imax) jmax = jmin - 1;                 // (nothing to walk)
        const long acb = t.ac_off[ic];
        const int ld = t.ngrdext_i[ic] + 1;
        for (int j = jmin; j <= jmax && !overflow; ++j) {
          const int lo = t.ac_ext[acb + (imin - 1) + (long)ld * (j - 1)];
          const int hi = t.ac_ext[acb + imax + (long)ld * (j - 1)];
          for (int base = lo; base < hi; base += 64) {
            const int row = base + lane;
            bool ok = false;
            double nd_h = 0.0, vobs = 0.0, err = 0.0;
            if (row < hi) {
              const double rdx = (ri - t.ob_ri[row]) * t.dx;      // :1876-1878
              const double rdy = (rj - t.ob_rj[row]) * t.dy;
              nd_h = sqrt(rdx * rdx + rdy * rdy) / hloc;
              ok = !(nd_h > kDistZeroFac);                        // :1881
              if (ok) {
                if (vloc != 0.0) {
                  if (vm == 1) vobs = t.ob_lev[row];
                  else if (vm == 2) vobs = log(t.ob_dat[row]);
                  else if (vm != 3) vobs = log(t.ob_lev[row]);
                }
                err = t.ob_err[row];
              }
            }
            const unsigned long long mk = __ballot(ok);
            const int na = __popcll(mk);
            if (ns + na > kSurvL) {                               // (wave-uniform)
              overflow = true;
              break;
            }
            if (ok) {
              const int si = ns + __popcll(mk & lt_mask);
              *reinterpret_cast<double2*>(&sb[4 * si]) = double2{nd_h, vobs};
              *reinterpret_cast<double2*>(&sb[4 * si + 2]) = double2{__longlong_as_double((long)row), err};
            }
            ns += na;
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (m - gs == q) {
            mbeg[q] = ns0;
            mend[q] = ns;
          }
      }
      wave_lds_sync();
      ns = uni(ns);
      overflow = uni((int)overflow) != 0;
      LP_ADD(0, LP_T() - lp0);
      LP_ADD(6, ns);
      const double cut_default = (t.criterion == 1) ? t.hori_loc[icm] * kDistZeroFac : 0.0;   // :1384-1389
      // the members' vertical parameters, once per group (wave-uniform)
      int vm_m[4], ic_m[4];
      double vloc_m[4], varloc_m[4];
      // fast path (distance criterion, at most two members, ordinary vertical scales): the keys are the distances
      // themselves (vertical_nd), the survivors' level-independent parts stay in registers over the level loop and the
      // division by the member's vertical scale goes through its reciprocal (letkf_divby_dev.h)
      bool fast = nmax > 0 && t.criterion == 1 && ge - gs <= 2 && !overflow;
#pragma unroll
      for (int mo = 0; mo < 4; ++mo) {
        const int ic = t.group_member[min(gs + mo, ge - 1)];
        ic_m[mo] = uni(ic);
        vm_m[mo] = uni(t.vmode[ic]);
        vloc_m[mo] = uni(t.vert_loc[ic]);
        varloc_m[mo] = uni(t.varloc[ic]);
        if (varloc_m[mo] < 1e-290) fast = false;                  // (rloc may underflow to 0 inside the cut-off)
        if (vloc_m[mo] != 0.0 && !divby::in_range(vloc_m[mo])) fast = false;
      }
      const bool count_only = !A.fill && !L.cutd_ctype && nmax > 0;
      const int mbeg1 = uni(mbeg[1]);                             // (INT_MAX: one member)
      double yv_m[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) yv_m[q] = uni(divby::reciprocal(vloc_m[q]));
      double ndh2_r[kKeyS], vobs_r[kKeyS];
      if (fast) {
#pragma unroll
        for (int u = 0; u < kKeyS; ++u) {
          ndh2_r[u] = __longlong_as_double(0x7FF0000000000000ll);  // +inf: padding lanes fail the cut-off test
          vobs_r[u] = 0.0;
          if (u * 64 >= ns) continue;
          const int si = u * 64 + lane;
          const int me = u * 64 >= mbeg1 ? mend[1] : mend[0];
          if (si < me) {
            const double2 a2 = *reinterpret_cast<const double2*>(&sb[4 * si]);
            ndh2_r[u] = a2.x * a2.x;
            vobs_r[u] = a2.y;
          }
        }
      }

      for (int lev = 0; lev < nlev; ++lev) {
        const long p = col + A.nij1 * (long)lev;
        const long out = (A.fill ? A.obs_off[p] : 0) + cntl[lev];
        int nsel = 0;
        double cutd = cut_default;
        int cm[4] = {0, 0, 0, 0};                                 // no-limit group: accepted rows per member (nobsl_t)
        if (overflow) {
          // ---- per-point multi-sweep for this (level, group): letkf_search_kernel's slow path
          const double rlev = A.rlev[p], rz = A.rz[p];
          SelState st{0ull, 56, 0ull, 0, 0};
          if (nmax <= 0) {
            nsel = A.fill ? sweep_group<kEmitAll>(PA, gs, ge, ri, rj, rlev, rz, out, st, hist)
                          : sweep_group<kCount>(PA, gs, ge, ri, rj, rlev, rz, out, st, hist);
          } else {
            const int cnt = sweep_group<kCount>(PA, gs, ge, ri, rj, rlev, rz, out, st, hist);
            if (cnt <= nmax) {
              nsel = cnt;
              if (A.fill) sweep_group<kEmitAll>(PA, gs, ge, ri, rj, rlev, rz, out, st, hist);
            } else {
              nsel = nmax;
              if (A.fill || L.cutd_ctype) {
                int want = nmax;
                for (int round = 0; round < 8; ++round) {
                  for (int b = lane; b < 256; b += 64) hist[b] = 0;
                  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                  __builtin_amdgcn_wave_barrier();
                  st.shift = 56 - 8 * round;
                  sweep_group<kHist>(PA, gs, ge, ri, rj, rlev, rz, out, st, hist);
                  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                  __builtin_amdgcn_wave_barrier();
                  int cum = 0, bsel = 255;
                  find_bin(hist, want, bsel, cum);
                  want -= cum;
                  st.prefix = (round == 0) ? (unsigned long long)bsel : ((st.prefix << 8) | (unsigned long long)bsel);
                }
                st.thresh = st.prefix;
                st.tie_budget = want;
                st.emitted = 0;
                if (A.fill) sweep_group<kEmitSelect>(PA, gs, ge, ri, rj, rlev, rz, out, st, hist);
                const double kv = __longlong_as_double((long long)(t.criterion == 2 ? 0x7FFFFFFFFFFFFFFFull - st.thresh : st.thresh));
                cutd = (t.criterion == 1) ? t.hori_loc[icm] * sqrt(kv) : kv;
              }
            }
          }
        } else {
          const double vz = uni(A.rz[p]), vlnp = uni(log(A.rlev[p]));
          const double lnrain = uni(log(t.rain_base));
          double vref_m[4], vconst_m[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            vref_m[q] = (vloc_m[q] != 0.0) ? (vm_m[q] == 1 ? vz : vlnp) : 0.0;
            vconst_m[q] = 0.0;
            if (vm_m[q] == 3 && vloc_m[q] != 0.0) vconst_m[q] = uni(fabs(lnrain - vref_m[q]) / vloc_m[q]);
          }
          // ---- (1) vertical part: keys (limited group) or straight emission (no limit)
          int acc_n = 0;
          [[maybe_unused]] const unsigned long long lp1 = LP_T();
          unsigned long long keyr[kKeyS];
          if (fast) {
            // three slots at a time in straight-line code (their dependent chains interleave), wave-uniform skips between
            static_assert(kKeyS % 3 == 0, "slots are walked in threes");
#pragma unroll
            for (int u = 0; u < kKeyS; ++u) keyr[u] = kNoKey;
#pragma unroll
            for (int g3 = 0; g3 < kKeyS; g3 += 3) {
              if (g3 * 64 >= ns) continue;                       // (wave-uniform)
              if (count_only && acc_n >= nmax) continue;         // (the count is min(accepted, limit): enough seen)
              unsigned long long am[3];
#pragma unroll
              for (int v = 0; v < 3; ++v) {
                const int u = g3 + v;
                const bool second = u * 64 >= mbeg1;             // the slot's member (wave-uniform)
                const int vm = second ? vm_m[1] : vm_m[0];
                const double vloc = second ? vloc_m[1] : vloc_m[0], yv = second ? yv_m[1] : yv_m[0];
                const double vref = second ? vref_m[1] : vref_m[0], vconst = second ? vconst_m[1] : vconst_m[0];
                const double qv = divby::quotient(fabs(vobs_r[u] - vref), vloc, yv);
                const double nd_v = (vloc == 0.0) ? 0.0 : (vm == 3 ? vconst : qv);      // :1851-1865
                const double nd = ndh2_r[u] + nd_v * nd_v;                              // :1888
                const bool acc = !(nd_v > kDistZeroFac) && !(nd > kDistZeroFacSq);      // :1869, :1891
                keyr[u] = acc ? (unsigned long long)__double_as_longlong(nd) : kNoKey;
                am[v] = __ballot(acc);
              }
              acc_n += __popcll(am[0]) + __popcll(am[1]) + __popcll(am[2]);
            }
          } else
#pragma unroll
          for (int u = 0; u < kKeyS; ++u) {
            keyr[u] = kNoKey;
            if (u * 64 >= ns) continue;                          // (wave-uniform)
            if (count_only && acc_n >= nmax) continue;           // (the count is min(accepted, limit): enough seen)
            // the slot's member (wave-uniform)
            int vm = vm_m[0], mo = 0, me = mend[0];
            double vloc = vloc_m[0], varloc = varloc_m[0], vref = vref_m[0], vconst = vconst_m[0];
#pragma unroll
            for (int q = 1; q < 4; ++q)
              if (u * 64 >= mbeg[q]) {
                mo = q;
                me = mend[q];
                vm = vm_m[q];
                vloc = vloc_m[q];
                varloc = varloc_m[q];
                vref = vref_m[q];
                vconst = vconst_m[q];
              }
            const int si = u * 64 + lane;
            VertOut vo{0.0, 0.0, 0.0, false};
            int row = 0;
            if (si < me) {
              const double2 a2 = *reinterpret_cast<const double2*>(&sb[4 * si]);
              const double2 b2 = *reinterpret_cast<const double2*>(&sb[4 * si + 2]);
              row = (int)__double_as_longlong(b2.x);
              vo = vertical_cal(vm, vloc, varloc, vconst, vref, a2.x, a2.y, b2.y);
            }
            const unsigned long long mk = __ballot(vo.acc);
            if (nmax > 0) {
              unsigned long long key = kNoKey;
              if (vo.acc) {
                CalOut c{vo.rloc, vo.rdiag, vo.nd};
                key = key_bits(t.criterion, c);
              }
              keyr[u] = key;
            } else {
              if (A.fill && vo.acc) {
                const long o = out + acc_n + __popcll(mk & lt_mask);
                A.obs_idx[o] = row;
                A.rdiag_l[o] = vo.rdiag;
                A.rloc_l[o] = vo.rloc;
              }
#pragma unroll
              for (int q = 0; q < 4; ++q)
                if (mo == q) cm[q] += __popcll(mk);
            }
            acc_n += __popcll(mk);
          }
          nsel = acc_n;
          [[maybe_unused]] const unsigned long long lp2 = LP_T();
          LP_ADD(1, lp2 - lp1);
          if (nmax > 0) {
            nsel = min(acc_n, nmax);
            unsigned long long thresh = kNoKey;                  // every real key is below it
            int tie_budget = 0;
            if (acc_n >= nmax && (acc_n > nmax ? (A.fill || L.cutd_ctype) : L.cutd_ctype != nullptr)) {
              // (2) the nmax-th smallest key: selection threshold, and the cut-off measure once the limit is hit
              hist_thresh<kKeyS>(keyr, ns, acc_n, nmax, hist, thresh, tie_budget, fast ? 256.0 / 13.5 : 0.0);
              const double kv = __longlong_as_double((long long)(t.criterion == 2 ? 0x7FFFFFFFFFFFFFFFull - thresh : thresh));
              cutd = (t.criterion == 1) ? t.hori_loc[icm] * sqrt(kv) : kv;
              if (acc_n == nmax) {
                thresh = kNoKey;
                tie_budget = 0;
              }
            }
            [[maybe_unused]] const unsigned long long lp3 = LP_T();
            LP_ADD(2, lp3 - lp2);
            LP_ADD(5, acc_n > nmax);
            if (A.fill) {
              // (3) emission in candidate order.  The selected candidates' buffer slots are first compacted into a list
              // (LDS, 16 bit each) in output order; the weights -- an exponential and two divisions each -- are then
              // computed for full wavefronts of SELECTED rows, not for every slot with a selected lane in it (a limit
              // of 100 out of ~380 survivors: 2 rounds instead of 6), and the stores are contiguous.
              unsigned short* list = reinterpret_cast<unsigned short*>(hist);   // [kSurvL]
              // (all the ballots first -- they do not depend on one another --, then the running positions on the
              // scalar unit, then the writes: walked slot by slot with the tie budget carried along, the
              // compare -> ballot -> count -> position chain of each slot waited for the one before)
              const bool ties = thresh != kNoKey;
              tie_budget = uni(tie_budget);
              unsigned long long mk[kKeyS], tk[kKeyS];
#pragma unroll
              for (int u = 0; u < kKeyS; ++u) {
                mk[u] = __ballot(keyr[u] < thresh);              // (kNoKey, the largest pattern, is below nothing)
                tk[u] = ties ? __ballot(keyr[u] == thresh) : 0ull;
              }
              int nl = 0;
#pragma unroll
              for (int u = 0; u < kKeyS; ++u) {
                if (u * 64 >= ns) continue;                      // (wave-uniform)
                const bool take = keyr[u] < thresh, tie = ties && keyr[u] == thresh;
                const int nm = __popcll(mk[u]), tpos = mbcnt(tk[u]);
                // (the order inside one slot: the strictly better ones, then the ties)
                if (take || (tie && tpos < tie_budget))
                  list[nl + (take ? mbcnt(mk[u]) : nm + tpos)] = (unsigned short)(u * 64 + lane);
                const int nt = min((int)__popcll(tk[u]), tie_budget);
                tie_budget -= nt;
                nl += nm + nt;
              }
              {
                wave_lds_sync();
                [[maybe_unused]] const unsigned long long lp4 = LP_T();
                LP_ADD(4, lp4 - lp3);
                // two wavefronts of selected rows at a time, in straight-line code: every one of them passed the cut-off
                // tests, so obs_local_cal's remaining arithmetic runs without its early exits and the two chains
                // (reciprocal-free division, exponential, division) interleave
                for (int r = 0; r < nl; r += 128) {
                  double rl[2], rd[2];
                  int rw[2];
#pragma unroll
                  for (int h = 0; h < 2; ++h) {
                    const int e = r + 64 * h + lane;
                    const int si = e < nl ? list[e] : 0;
                    const double2 a2 = *reinterpret_cast<const double2*>(&sb[4 * si]);
                    const double2 b2 = *reinterpret_cast<const double2*>(&sb[4 * si + 2]);
                    int vm = vm_m[0];
                    double vloc = vloc_m[0], varloc = varloc_m[0], vref = vref_m[0], vconst = vconst_m[0];
#pragma unroll
                    for (int q = 1; q < 4; ++q)
                      if (si >= mbeg[q]) {
                        vm = vm_m[q];
                        vloc = vloc_m[q];
                        varloc = varloc_m[q];
                        vref = vref_m[q];
                        vconst = vconst_m[q];
                      }
                    const double qv = fabs(a2.y - vref) / vloc;
                    const double nd_v = (vloc == 0.0) ? 0.0 : (vm == 3 ? vconst : qv);      // :1851-1865
                    const double nd = a2.x * a2.x + nd_v * nd_v;                            // :1888
                    rl[h] = varloc * exp(-0.5 * nd);                                        // :1899
                    rd[h] = b2.y * b2.y / rl[h];                                            // :1903
                    rw[h] = (int)__double_as_longlong(b2.x);
                  }
#pragma unroll
                  for (int h = 0; h < 2; ++h) {
                    const int e = r + 64 * h + lane;
                    if (e < nl) {
                      const long o = out + e;
                      A.obs_idx[o] = rw[h];
                      A.rdiag_l[o] = rd[h];
                      A.rloc_l[o] = rl[h];
                    }
                  }
                }
                wave_lds_sync();
              }
            }
            if (acc_n < nmax) cutd = cut_default;
            LP_ADD(3, LP_T() - lp3);
          }
        }
        if (lane == 0) {
          cntl[lev] += nsel;
          if (A.nobs_ctype) {
            if (nmax > 0) {
              A.nobs_ctype[p * t.nctype + icm] = nsel;            // nobsl_t of the master (:1633, :1713)
            } else if (!overflow) {
#pragma unroll
              for (int mo = 0; mo < 4; ++mo)
                if (gs + mo < ge) A.nobs_ctype[p * t.nctype + t.group_member[gs + mo]] = cm[mo];
            } else {
              A.nobs_ctype[p * t.nctype + icm] = nsel;            // (fall-back path: the group's total on its master)
            }
          }
          if (L.cutd_ctype) L.cutd_ctype[p * t.nctype + icm] = cutd;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    if (!A.fill)
      for (int l = lane; l < nlev; l += 64) A.counts[col + A.nij1 * (long)l] = cntl[l];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  LP_FLUSH();
}

hipError_t launch_search_columns_limited(const letkf_search_tables& t, long nij1, int nlev, const double* rig,
                                         const double* rjg, const double* rlev, const double* rz, int fill, int* counts,
                                         const long* obs_off, int* obs_idx, double* rdiag_l, double* rloc_l,
                                         int* nobs_ctype, double* cutd_ctype, int num_cu, hipStream_t st) {
  ColLimArgs a{{t, nij1, nlev, rig, rjg, rlev, rz, fill, counts, obs_off, obs_idx, rdiag_l, rloc_l, nobs_ctype},
               cutd_ctype};
  const size_t lds = (size_t)4 * (4 * kSurvL + ((nlev + 1) & ~1)) * sizeof(double);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&letkf_search_columns_limited_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const long nwg = (nij1 + 3) / 4;
  const long g = (long)num_cu * 8;
  const int grid = (int)(nwg < g ? (nwg > 0 ? nwg : 1) : g);
#ifdef LETKF_WAVE_PROF
  unsigned long long z[8] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lim_prof), z, sizeof z);
#endif
  hipLaunchKernelGGL(letkf_search_columns_limited_kernel, dim3(grid), dim3(256), lds, st, a);
#ifdef LETKF_WAVE_PROF
  (void)hipStreamSynchronize(st);
  (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(g_lim_prof), sizeof z);
  fprintf(stderr, "LIM_PROF fill=%d grid=%d horiz=%llu vert=%llu select=%llu emit=%llu nsel=%llu sum_ns=%llu\n", fill, grid, z[0], z[1],
          z[2], z[3], z[5], z[6]);
  fprintf(stderr, "LIM_PROF   emit: list=%llu (first turn; the rest is weights + stores)\n", z[4]);
#endif
  return hipGetLastError();
}

// ------------------------------------------------------------------ MAX_NOBS_PER_GRID on DENSE observations: rings
// The column kernel above buffers a group's horizontal survivors in LDS (kSurvL = 576) and falls back to a per-point
// multi-sweep radix select when they do not fit -- ten walks over the sorting mesh per (point, group).  BASELINE configs[3]
// with the reference's usual MAX_NOBS_PER_GRID = 100 has ~5000 survivors per column and group: 55 s of obs_local per analysis
// for a loop body of 1.3 s.  For the distance criterion the selection key is nd = nd_h^2 + nd_v^2 >= nd_h^2, so the survivors of
// a (column, group) are kept in global memory ORDERED BY RINGS of nd_h^2 (kRings equal steps up to the cut-off): a level takes
// them tile by tile, nearest ring first, carries the best nmax found so far along, and stops as soon as the nmax-th best key
// is below the lower bound of the next ring -- at configs[3] after the two innermost rings of sixteen.  (The weight criterion, 2,
// orders like the distance where a group's types share one variable-localisation factor: served too.)  Exact: the selected
// SET is the reference's (up to ties of equal keys, as everywhere), nobsl_t and the cut-off measure likewise.
constexpr int kRings = 16;
constexpr double kRingScale = (double)kRings / 13.5;    // ring = min(kRings - 1, (int)(nd_h^2 * kRingScale)); 13.5 > dist_zero_fac^2

struct RingBuildArgs {
  letkf_search_tables t;
  long col0, ncol;
  const double *rig, *rjg;
  int* counts;           // FILL = 0: [ncol * ngroup] survivors per (column, group)
  const long* goff;      // FILL = 1: [ncol * ngroup + 1] entry offsets (absolute), group-fastest
  double* sv;            // entries: (row | ctype << 32 as bits, nd_h, v_obs, err)
  int* roff;             // FILL = 1: [ncol * ngroup][kRings + 1] ring starts relative to the (column, group)'s first entry
  int gen;               // general ring key (r4, see ring_offset): criterion 3, and criterion 2 with several factors in a group
  const double* kref;    // gen: [ngroup] the group's reference offset (a lower estimate of its entries' offsets)
};

// The general ring key (r4).  Selection by weight (criterion 2: largest rloc = varloc exp(-nd / 2)) and by error (criterion 3:
// smallest rdiag = err^2 / rloc) order like  nd + off  with an offset per ENTRY,
//     criterion 2: off = -2 ln varloc(ctype)          criterion 3: off = 2 ln(err^2 / varloc(ctype)),
// so the survivors are ringed by  nd_h^2 + off - kref(group)  (kref: the smallest offset the host could find for the group;
// an entry below it lands in ring 0, which has no lower bound anyway) and a ring's lower edge bounds the exact key of
// everything behind it: rdiag >= exp((edge + kref) / 2), rloc <= exp(-(edge + kref) / 2).  The SELECTION itself compares the
// reference's own numbers (rloc, rdiag as obs_local_cal computes them), never the ring key.
__device__ __forceinline__ double ring_offset(const int criterion, const double varloc, const double err) {
  return criterion == 2 ? -2.0 * log(varloc) : 2.0 * log(err * err / varloc);
}

template <bool FILL>
__global__ void __launch_bounds__(256) letkf_ring_survivors_kernel(const RingBuildArgs A) {
#pragma clang fp contract(off)
  __shared__ int rpos_all[4][kRings + 1];
  const letkf_search_tables& t = A.t;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int* rpos = rpos_all[wv];
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (long cb = (long)blockIdx.x * 4 + wv; cb < A.ncol; cb += (long)gridDim.x * 4) {
    const long col = A.col0 + cb;
    const double ri = A.rig[col], rj = A.rjg[col];
    for (int ig = 0; ig < t.ngroup; ++ig) {
      const int gs = t.group_start[ig], ge = t.group_start[ig + 1];
      const bool ringed = t.max_nobs[t.group_member[gs]] > 0;      // (an unlimited group stays in the list order: one ring)
      // walk(what): 0 count, 1 ring histogram, 2 placement
      auto walk = [&](const int what, const long out) -> long {
        long ns = 0;
        for (int m = gs; m < ge; ++m) {
          const int ic = t.group_member[m];
          if (t.varloc[ic] < kTiny) continue;                       // local_cal :1843
          const int vm = t.vmode[ic];
          const double vloc = t.vert_loc[ic], hloc = t.hori_loc[ic];
          const double dzi = hloc * kDistZeroFac / t.dx, dzj = hloc * kDistZeroFac / t.dy;
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
          for (int j = jmin; j <= jmax; ++j) {
            const int lo = t.ac_ext[acb + (imin - 1) + (long)ld * (j - 1)];
            const int hi = t.ac_ext[acb + imax + (long)ld * (j - 1)];
            for (int base = lo; base < hi; base += 64) {
              const int row = base + lane;
              bool ok = false;
              double nd_h = 0.0;
              if (row < hi) {
                const double rdx = (ri - t.ob_ri[row]) * t.dx;      // :1876-1878
                const double rdy = (rj - t.ob_rj[row]) * t.dy;
                nd_h = sqrt(rdx * rdx + rdy * rdy) / hloc;
                ok = !(nd_h > kDistZeroFac);                        // :1881
              }
              const unsigned long long mk = __ballot(ok);
              double hk = nd_h * nd_h;
              if (A.gen && what != 0 && ok) hk = fmax(hk + ring_offset(t.criterion, t.varloc[ic], t.ob_err[row]) - A.kref[ig], 0.0);
              if (what == 1 && mk) {
                // the histogram: one LDS atomic per survivor (the ORDER of the additions is irrelevant here)
                const int ring = ringed ? min(kRings - 1, (int)(hk * kRingScale)) : 0;
                if (ok) atomicAdd(&rpos[ring], 1);
              } else if (what == 2 && mk) {
                const int ring = ringed ? min(kRings - 1, (int)(hk * kRingScale)) : 0;
                // lane order inside a ring: deterministic positions (an LDS atomic per lane would scatter them run by run)
                for (int r = 0; r < kRings; ++r) {
                  const unsigned long long mr = __ballot(ok && ring == r);
                  if (!mr) continue;
                  const int b0 = rpos[r];
                  if (what == 2 && ok && ring == r) {
                    double vobs = 0.0;
                    if (vloc != 0.0) {
                      if (vm == 1) vobs = t.ob_lev[row];
                      else if (vm == 2) vobs = log(t.ob_dat[row]);
                      else if (vm != 3) vobs = log(t.ob_lev[row]);
                    }
                    const long o = 4 * (out + b0 + __popcll(mr & lt_mask));
                    *reinterpret_cast<double2*>(&A.sv[o]) = double2{__longlong_as_double((long)row | ((long)ic << 32)), nd_h};
                    *reinterpret_cast<double2*>(&A.sv[o + 2]) = double2{vobs, t.ob_err[row]};
                  }
                  wave_lds_sync();
                  if (lane == 0) rpos[r] = b0 + __popcll(mr);
                  wave_lds_sync();
                }
              }
              ns += __popcll(mk);
            }
          }
        }
        return ns;
      };
      if (!FILL) {
        const long ns = walk(0, 0);
        if (lane == 0) A.counts[cb * t.ngroup + ig] = (int)ns;
      } else {
        const long e0 = A.goff[cb * t.ngroup + ig];
        if (lane <= kRings) rpos[lane] = 0;
        wave_lds_sync();
        walk(1, 0);                                                 // rpos[r] = survivors in ring r
        wave_lds_sync();
        // (exclusive scan by lane 0 -- 16 additions)
        wave_lds_sync();
        if (lane == 0) {
          int acc = 0;
          for (int r = 0; r < kRings; ++r) {
            const int cr = rpos[r];
            rpos[r] = acc;
            A.roff[(cb * t.ngroup + ig) * (kRings + 1) + r] = acc;
            acc += cr;
          }
          A.roff[(cb * t.ngroup + ig) * (kRings + 1) + kRings] = acc;
        }
        wave_lds_sync();
        walk(2, e0);
        wave_lds_sync();
      }
    }
  }
}

struct RingSearchArgs {
  letkf_search_tables t;
  long col0, ncol, nij1;
  int nlev;
  const double *rlev, *rz;
  int fill;
  int* counts;
  const long* obs_off;
  int* obs_idx;
  double *rdiag_l, *rloc_l;
  int* nobs_ctype;
  double* cutd_ctype;
  const long* goff;      // [ncol * ngroup + 1] absolute entry offsets of this batch's columns
  const double* sv;      // (base shifted so that the absolute offsets address it)
  const int* roff;       // [ncol * ngroup][kRings + 1]
  const double* kref;    // GEN: [ngroup] (RingBuildArgs)
};
constexpr int kRingSlots = 6;    // slots of 64 keys per lane a tile takes (carried selection + new entries): with the 9 of the LDS kernel the tile spilled 1.2 KB per lane
constexpr int kRingSel = 128;    // largest MAX_NOBS_PER_GRID this kernel serves (the carried selection lives in LDS and re-enters every tile)

// GEN: the general ring key (ring_offset): the tile's keys are the reference's own selection numbers -- key_bits of rloc
// (criterion 2) / rdiag (criterion 3), computed for every entry of a visited ring --, the value carried beside a key is the
// observation error (2) / rloc (3), and the stop test compares the ring edge's bound in those units.
template <bool GEN>
__global__ void __launch_bounds__(256, 2) letkf_search_rings_kernel(const RingSearchArgs A) {
#pragma clang fp contract(off)
  __shared__ __attribute__((aligned(16))) unsigned int hist_all[4][kSurvL / 2];
  __shared__ unsigned long long bkey_all[4][kRingSel];
  __shared__ long brw_all[4][kRingSel];
  __shared__ double berr_all[4][kRingSel];
  __shared__ int s_vm[64], s_nct[4][64];
  __shared__ double s_vloc[64], s_varloc[64];
  const letkf_search_tables& t = A.t;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned int* hist = hist_all[wv];
  unsigned long long* bkey = bkey_all[wv];
  long* brw = brw_all[wv];
  double* berr = berr_all[wv];
  int* nct = s_nct[wv];
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int i = threadIdx.x; i < t.nctype; i += 256) {
    s_vm[i] = t.vmode[i];
    s_vloc[i] = t.vert_loc[i];
    s_varloc[i] = t.varloc[i];
  }
  __syncthreads();
  const double lnrain = log(t.rain_base);
  const int ng = t.ngroup;
  for (long cb = (long)blockIdx.x * 4 + wv; cb < A.ncol; cb += (long)gridDim.x * 4) {
    const long col = A.col0 + cb;
    for (int lev = 0; lev < A.nlev; ++lev) {
      const long p = col + A.nij1 * (long)lev;
      const double vz = A.rz[p], vlnp = log(A.rlev[p]);
      const long out0 = A.fill ? A.obs_off[p] : 0;
      int emitted = 0;
      if (A.nobs_ctype || A.cutd_ctype)                            // defaults: letkf_tools.f90:1380-1391, :1427-1432
        for (int ic = lane; ic < t.nctype; ic += 64) {
          if (A.nobs_ctype) A.nobs_ctype[p * t.nctype + ic] = 0;
          if (A.cutd_ctype) A.cutd_ctype[p * t.nctype + ic] = 0.0;
        }
      for (int ig = 0; ig < ng; ++ig) {
        const int gs = t.group_start[ig];
        const int icm = t.group_member[gs];
        const int nmax = t.max_nobs[icm];
        const long e0 = A.goff[cb * ng + ig], e1 = A.goff[cb * ng + ig + 1];
        const int n_g = (int)(e1 - e0);
        const int* ro = A.roff + (cb * ng + ig) * (kRings + 1);
        // one entry -> its selection key (the squared normalised distance as bits), or kNoKey
        auto entry_key = [&](const double2 a2, const double2 b2, double& nd_out) -> bool {
          const int ic = (int)(__double_as_longlong(a2.x) >> 32);
          const int vm = s_vm[ic];
          const double vloc = s_vloc[ic];
          const double vref = vm == 1 ? vz : vlnp;
          double vconst = 0.0;
          if (vm == 3 && vloc != 0.0) vconst = fabs(lnrain - vref) / vloc;
          const VertOut vo = vertical_nd(vm, vloc, s_varloc[ic], vconst, vloc != 0.0 ? vref : 0.0, a2.y, b2.x, b2.y);
          nd_out = vo.nd;
          return vo.acc;
        };
        int nsel = 0;
        double cutd = (t.criterion == 1) ? t.hori_loc[icm] * kDistZeroFac : 0.0;   // :1384-1389
        if (nmax <= 0) {
          // ---- no limit: every accepted row, in the list order
          if (A.nobs_ctype) {
            nct[lane] = 0;
            wave_lds_sync();
          }
          for (int pos = 0; pos < n_g; pos += 64) {
            const int e = pos + lane;
            bool acc = false;
            double nd = 0.0;
            double2 a2{0.0, 0.0}, b2{0.0, 0.0};
            if (e < n_g) {
              a2 = *reinterpret_cast<const double2*>(&A.sv[4 * (e0 + e)]);
              b2 = *reinterpret_cast<const double2*>(&A.sv[4 * (e0 + e) + 2]);
              acc = entry_key(a2, b2, nd);
            }
            const unsigned long long mk = __ballot(acc);
            if (acc) {
              const long rw = __double_as_longlong(a2.x);
              const int ic = (int)(rw >> 32);
              if (A.fill) {
                const long o = out0 + emitted + nsel + __popcll(mk & lt_mask);
                const double rloc = s_varloc[ic] * exp(-0.5 * nd);                       // :1899
                A.obs_idx[o] = (int)(rw & 0xffffffffL);
                A.rdiag_l[o] = b2.y * b2.y / rloc;                                        // :1903
                A.rloc_l[o] = rloc;
              }
              if (A.nobs_ctype) atomicAdd(&nct[ic], 1);
            }
            nsel += __popcll(mk);
          }
          if (A.nobs_ctype) {
            wave_lds_sync();
            for (int m = gs + lane; m < t.group_start[ig + 1]; m += 64) {
              const int ic = t.group_member[m];
              A.nobs_ctype[p * t.nctype + ic] = nct[ic];
            }
            wave_lds_sync();
          }
        } else {
          // ---- the nmax nearest: tiles in ring order, the selection so far carried in LDS and re-entered with every tile
          const bool count_only = !A.fill && !A.cutd_ctype;
          int nB = 0, tot = 0, pos = 0, rp = 0;
          unsigned long long tau = kNoKey;
          while (pos < n_g) {
            while (rp + 1 < kRings && ro[rp + 1] <= pos) ++rp;      // the ring the next entry lies in
            if (count_only && tot >= nmax) break;
            if (nB == nmax) {
              if constexpr (!GEN) {
                const double lb = (double)rp * (1.0 / kRingScale) * (1.0 - 1e-12);   // nd >= nd_h^2 >= lb for everything from here on
                if ((unsigned long long)__double_as_longlong(lb) >= tau) break;
              } else if (rp > 0) {                                   // (ring 0 also holds what lies below the reference offset)
                const double edge = (double)rp * (1.0 / kRingScale) + A.kref[ig];
                const unsigned long long lbk =
                    t.criterion == 2 ? 0x7FFFFFFFFFFFFFFFull - (unsigned long long)__double_as_longlong(exp(-0.5 * edge) * (1.0 + 1e-9))
                                     : (unsigned long long)__double_as_longlong(exp(0.5 * edge) * (1.0 - 1e-9));
                if (lbk >= tau) break;
              }
            }
            const int nbs = (nB + 63) >> 6;                          // slots the carried selection takes
            unsigned long long keyr[kRingSlots];
            long rwr[kRingSlots];
            double errr[kRingSlots];
            int nreal = 0, nnew = 0;
            // the tile's entries: every load issued before the first is used, unconditionally from a clamped address (written
            // as `if (e < n_g) load` per slot, hipcc waits vmcnt(0) behind each branch: a memory round trip per slot, 12 us per
            // tile -- found in the kernel stats: 65 us per (column, level, group) instead of ~3)
            double2 ta[kRingSlots], tb[kRingSlots];
#pragma unroll
            for (int u = 0; u < kRingSlots; ++u) {
              const int e = min(pos + (u - nbs) * 64 + lane, n_g - 1);
              const long ea = e0 + (u >= nbs ? e : 0);
              ta[u] = *reinterpret_cast<const double2*>(&A.sv[4 * ea]);
              tb[u] = *reinterpret_cast<const double2*>(&A.sv[4 * ea + 2]);
            }
#pragma unroll
            for (int u = 0; u < kRingSlots; ++u) {
              keyr[u] = kNoKey;
              rwr[u] = 0;
              errr[u] = 0.0;
              if (u < nbs) {
                const int j = u * 64 + lane;
                if (j < nB) {
                  keyr[u] = bkey[j];
                  rwr[u] = brw[j];
                  errr[u] = berr[j];
                }
              } else {
                const int e = pos + (u - nbs) * 64 + lane;
                double nd;
                if (e < n_g && entry_key(ta[u], tb[u], nd)) {
                  rwr[u] = __double_as_longlong(ta[u].x);
                  if constexpr (!GEN) {
                    keyr[u] = (unsigned long long)__double_as_longlong(nd);
                    errr[u] = tb[u].y;
                  } else {
                    const double rloc = s_varloc[(int)(rwr[u] >> 32)] * exp(-0.5 * nd);          // :1899
                    if (t.criterion == 2) {
                      keyr[u] = 0x7FFFFFFFFFFFFFFFull - (unsigned long long)__double_as_longlong(rloc);
                      errr[u] = tb[u].y;
                    } else {
                      keyr[u] = (unsigned long long)__double_as_longlong(tb[u].y * tb[u].y / rloc);   // :1903
                      errr[u] = rloc;
                    }
                  }
                }
              }
              const int c = __popcll(__ballot(keyr[u] != kNoKey));
              nreal += c;
              if (u >= nbs) nnew += c;
            }
            tot += nnew;
            pos += (kRingSlots - nbs) * 64;
            if (count_only) continue;
            wave_lds_sync();                                         // (everybody has read the carried selection)
            unsigned long long thresh = kNoKey;
            int tie_budget = 0;
            if (nreal > nmax) hist_thresh<kRingSlots>(keyr, kRingSlots * 64, nreal, nmax, hist, thresh, tie_budget, GEN ? 0.0 : 256.0 / 13.5);
            tie_budget = uni(tie_budget);
            const bool ties = thresh != kNoKey;
            int nl = 0;
#pragma unroll
            for (int u = 0; u < kRingSlots; ++u) {
              const bool take = keyr[u] < thresh, tie = ties && keyr[u] == thresh;   // (kNoKey is below nothing)
              const unsigned long long mk = __ballot(take), tk = __ballot(tie);
              const int nm = __popcll(mk), tpos = mbcnt(tk);
              if (take || (tie && tpos < tie_budget)) {
                const int j = nl + (take ? mbcnt(mk) : nm + tpos);
                bkey[j] = keyr[u];
                brw[j] = rwr[u];
                berr[j] = errr[u];
              }
              const int nt = min((int)__popcll(tk), tie_budget);
              tie_budget -= nt;
              nl += nm + nt;
            }
            nB = nl;
            if (nreal >= nmax) {
              // tau = the nmax-th best key so far (with nreal == nmax: the largest key of the selection)
              if (nreal > nmax) tau = thresh;
              else {
                unsigned int hi = 0u;
#pragma unroll
                for (int u = 0; u < kRingSlots; ++u)
                  if (keyr[u] != kNoKey) hi = max(hi, (unsigned int)(keyr[u] >> 32));
                tau = ((unsigned long long)wave_max_u32(hi) + 1ull) << 32;   // (an upper bound is enough for the stop test)
              }
            }
            wave_lds_sync();
          }
          nsel = count_only ? min(tot, nmax) : nB;
          if (!count_only && nB == nmax) {
            // the cut-off measure once the limit is reached: hori_loc * sqrt(nmax-th smallest squared distance) (:1384-1389)
            unsigned long long kmx = 0ull;
            for (int j = lane; j < nB; j += 64) kmx = bkey[j] > kmx ? bkey[j] : kmx;
            const unsigned int hmx = wave_max_u32((unsigned int)(kmx >> 32));
            unsigned int lmx = ((unsigned int)(kmx >> 32) == hmx) ? (unsigned int)kmx : 0u;
            lmx = wave_max_u32(lmx);
            const double kv = __longlong_as_double((long long)(((unsigned long long)hmx << 32) | lmx));
            // (criterion 2: the smallest selected weight -- one variable-localisation factor per group, the host checked)
            if constexpr (!GEN) cutd = (t.criterion == 1) ? t.hori_loc[icm] * sqrt(kv) : s_varloc[icm] * exp(-0.5 * kv);
            else   // the smallest selected weight / the largest selected error variance (:1716-1727): the largest key itself
              cutd = t.criterion == 2 ? __longlong_as_double((long long)(0x7FFFFFFFFFFFFFFFull - (((unsigned long long)hmx << 32) | lmx))) : kv;
          }
          if (A.fill) {
            for (int j = lane; j < nB; j += 64) {
              const long rw = brw[j];
              const int ic = (int)(rw >> 32);
              double rloc, rdiag;
              if constexpr (!GEN) {
                const double nd = __longlong_as_double((long long)bkey[j]);
                rloc = s_varloc[ic] * exp(-0.5 * nd);                                    // :1899
                rdiag = berr[j] * berr[j] / rloc;                                         // :1903
              } else if (t.criterion == 2) {
                rloc = __longlong_as_double((long long)(0x7FFFFFFFFFFFFFFFull - bkey[j]));
                rdiag = berr[j] * berr[j] / rloc;
              } else {
                rloc = berr[j];
                rdiag = __longlong_as_double((long long)bkey[j]);
              }
              const long o = out0 + emitted + j;
              A.obs_idx[o] = (int)(rw & 0xffffffffL);
              A.rdiag_l[o] = rdiag;
              A.rloc_l[o] = rloc;
            }
          }
          if (lane == 0) {
            if (A.nobs_ctype) A.nobs_ctype[p * t.nctype + icm] = nsel;                    // nobsl_t of the master (:1633, :1713)
            if (A.cutd_ctype) A.cutd_ctype[p * t.nctype + icm] = cutd;
          }
          wave_lds_sync();
        }
        emitted += nsel;
      }
      if (!A.fill && lane == 0) A.counts[p] = emitted;
    }
  }
}

// smallest observation error of every combined type (one workgroup per type; the rows of a type are one range of the table:
// from the first to the last prefix sum of its mesh) -- for the reference offsets of the general ring key (criterion 3)
__global__ void __launch_bounds__(256) letkf_ctype_min_err_kernel(const letkf_search_tables t, double* out) {
  __shared__ double red[256];
  const int ic = blockIdx.x;
  const long acb = t.ac_off[ic];
  const long nac = (long)(t.ngrdext_i[ic] + 1) * t.ngrdext_j[ic];
  const int lo = t.ac_ext[acb], hi = t.ac_ext[acb + nac - 1];
  double m = 1e300;
  for (int r = lo + threadIdx.x; r < hi; r += 256) m = fmin(m, t.ob_err[r]);
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s_ = 128; s_ > 0; s_ >>= 1) {
    if ((int)threadIdx.x < s_) red[threadIdx.x] = fmin(red[threadIdx.x], red[threadIdx.x + s_]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[ic] = red[0];
}
hipError_t launch_ctype_min_err(const letkf_search_tables& t, double* out, hipStream_t st) {
  hipLaunchKernelGGL(letkf_ctype_min_err_kernel, dim3(t.nctype), dim3(256), 0, st, t, out);
  return hipGetLastError();
}

hipError_t launch_ring_survivors(const letkf_search_tables& t, long col0, long ncol, const double* rig, const double* rjg, int fill,
                                 int* counts, const long* goff, double* sv, int* roff, const double* kref, int num_cu, hipStream_t st) {
  if (ncol <= 0) return hipSuccess;
  RingBuildArgs a{t, col0, ncol, rig, rjg, counts, goff, sv, roff, kref != nullptr ? 1 : 0, kref};
  const long nwg = (ncol + 3) / 4;
  const long g = (long)num_cu * 8;
  const int grid = (int)(nwg < g ? nwg : g);
  if (fill) hipLaunchKernelGGL(letkf_ring_survivors_kernel<true>, dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(letkf_ring_survivors_kernel<false>, dim3(grid), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_search_rings(const letkf_search_tables& t, long col0, long ncol, long nij1, int nlev, const double* rlev,
                               const double* rz, int fill, int* counts, const long* obs_off, int* obs_idx, double* rdiag_l,
                               double* rloc_l, int* nobs_ctype, double* cutd_ctype, const long* goff, const double* sv,
                               const int* roff, const double* kref, int num_cu, hipStream_t st) {
  if (ncol <= 0) return hipSuccess;
  RingSearchArgs a{t, col0, ncol, nij1, nlev, rlev, rz, fill, counts, obs_off, obs_idx, rdiag_l, rloc_l, nobs_ctype, cutd_ctype,
                   goff, sv, roff, kref};
  const long nwg = (ncol + 3) / 4;
  const long g = (long)num_cu * 4;
  const int grid = (int)(nwg < g ? nwg : g);
  if (kref) hipLaunchKernelGGL(letkf_search_rings_kernel<true>, dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(letkf_search_rings_kernel<false>, dim3(grid), dim3(256), 0, st, a);
  return hipGetLastError();
}

int search_rings_max_nobs() { return kRingSel; }
int search_rings_count() { return kRings; }
int search_rings_lds_survivors() { return kSurvL; }

// ------------------------------------------------------------------ the horizontal half of obs_local, once per column
// (the list-free route of letkf_das_columns_dev; the vertical half runs inside the loop body kernel, letkf_wave.hip mode 3).
// One wave per column walks the rectangle of sorting-mesh cells of every combined type exactly like the column search and
// keeps the rows inside the horizontal cut-off, in the reference's list order: entry = (row | ctype << 32 as bits, nd_h,
// v_obs, err) with v_obs the observation's vertical coordinate in its ctype's mode (lev, ln lev, ln dat); the entries of a
// type are padded to a multiple of 64.
// FILL = false: counts[col] = survivors of the column; FILL = true: writes them at sv + 4 * sv_off[col].
struct SurvArgs {
  letkf_search_tables t;
  long col0, ncol;       // columns col0 .. col0 + ncol of rig / rjg; counts / sv_off are indexed by the column's number in the call
  const double *rig, *rjg;
  int* counts;
  const long* sv_off;
  double* sv;
};

template <bool FILL>
__global__ void __launch_bounds__(256) letkf_survivors_kernel(const SurvArgs A) {
#pragma clang fp contract(off)
  const letkf_search_tables& t = A.t;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (long cb = (long)blockIdx.x * 4 + wv; cb < A.ncol; cb += (long)gridDim.x * 4) {
    const long col = A.col0 + cb;
    const double ri = A.rig[col], rj = A.rjg[col];
    long ns = 0;
    const long out = FILL ? A.sv_off[cb] : 0;
    for (int m = 0; m < t.group_start[t.ngroup]; ++m) {
      const int ic = t.group_member[m];
      if (t.varloc[ic] < kTiny) continue;                         // local_cal :1843
      const int vm = t.vmode[ic];
      const double vloc = t.vert_loc[ic], hloc = t.hori_loc[ic];
      const double dzi = hloc * kDistZeroFac / t.dx, dzj = hloc * kDistZeroFac / t.dy;
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
      for (int j = jmin; j <= jmax; ++j) {
        const int lo = t.ac_ext[acb + (imin - 1) + (long)ld * (j - 1)];
        const int hi = t.ac_ext[acb + imax + (long)ld * (j - 1)];
        for (int base = lo; base < hi; base += 64) {
          const int row = base + lane;
          bool ok = false;
          double nd_h = 0.0;
          if (row < hi) {
            const double rdx = (ri - t.ob_ri[row]) * t.dx;        // :1876-1878
            const double rdy = (rj - t.ob_rj[row]) * t.dy;
            nd_h = sqrt(rdx * rdx + rdy * rdy) / hloc;
            ok = !(nd_h > kDistZeroFac);                          // :1881
          }
          const unsigned long long mk = __ballot(ok);
          if (FILL && ok) {
            double vobs = 0.0;
            if (vloc != 0.0) {
              if (vm == 1) vobs = t.ob_lev[row];
              else if (vm == 2) vobs = log(t.ob_dat[row]);
              else if (vm != 3) vobs = log(t.ob_lev[row]);
            }
            const long o = 4 * (out + ns + __popcll(mk & lt_mask));
            const long rw = (long)row | ((long)ic << 32);
            *reinterpret_cast<double2*>(&A.sv[o]) = double2{__longlong_as_double(rw), nd_h};
            *reinterpret_cast<double2*>(&A.sv[o + 2]) = double2{vobs, t.ob_err[row]};
          }
          ns += __popcll(mk);
        }
      }
      // every type's segment is padded to whole chunks of 64 entries (nd_h = 1e30: outside every cut-off), so that a chunk
      // of the loop body kernel is of ONE type and reads the type's numbers through the scalar cache
      const long npad = (64 - (ns & 63)) & 63;
      if (FILL && lane < npad) {
        const long o = 4 * (out + ns + lane);
        *reinterpret_cast<double2*>(&A.sv[o]) = double2{__longlong_as_double((long)ic << 32), 1e30};
        *reinterpret_cast<double2*>(&A.sv[o + 2]) = double2{0.0, 1.0};
      }
      ns += npad;
    }
    if (!FILL && lane == 0) A.counts[cb] = (int)ns;
  }
}

hipError_t launch_survivors(const letkf_search_tables& t, long col0, long ncol, const double* rig, const double* rjg, int fill,
                            int* counts, const long* sv_off, double* sv, int num_cu, hipStream_t st) {
  if (ncol <= 0) return hipSuccess;
  SurvArgs a{t, col0, ncol, rig, rjg, counts, sv_off, sv};
  const long nwg = (ncol + 3) / 4;
  const long g = (long)num_cu * 8;
  const int grid = (int)(nwg < g ? nwg : g);
  if (fill) hipLaunchKernelGGL(letkf_survivors_kernel<true>, dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(letkf_survivors_kernel<false>, dim3(grid), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_search_columns(const letkf_search_tables& t, long nij1, int nlev, const double* rig,
                                 const double* rjg, const double* rlev, const double* rz, int fill, int* counts,
                                 const long* obs_off, int* obs_idx, double* rdiag_l, double* rloc_l, int* nobs_ctype,
                                 int num_cu, hipStream_t st) {
  ColArgs a{t, nij1, nlev, rig, rjg, rlev, rz, fill, counts, obs_off, obs_idx, rdiag_l, rloc_l, nobs_ctype};
  const size_t lds = (size_t)4 * (4 * kSurv + 2 * ((nlev + 1) & ~1)) * sizeof(double);
  auto kern = fill ? &letkf_search_columns_kernel<true> : &letkf_search_columns_kernel<false>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  const long nwg = (nij1 + 3) / 4;
  const long g = (long)num_cu * 8;
  const int grid = (int)(nwg < g ? (nwg > 0 ? nwg : 1) : g);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
  return hipGetLastError();
}

hipError_t launch_search(const SearchArgs& a, int num_cu, hipStream_t st) {
  const long nwg = (a.npts + 3) / 4;
  const long g = (long)num_cu * 8;
  const int grid = (int)(nwg < g ? (nwg > 0 ? nwg : 1) : g);
  // the candidate cache is only touched in limited mode: without a limit the kernel keeps its full occupancy
  const size_t lds = a.limited ? (size_t)4 * 4 * kCacheCap * sizeof(double) : 0;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&letkf_search_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(letkf_search_kernel, dim3(grid), dim3(256), lds, st, a);
  return hipGetLastError();
}

}  // namespace letkf
