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
#include <stdint.h>

#include "../../include/letkf_amd.h"
#include "letkf_device.h"
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

__global__ void __launch_bounds__(256) letkf_search_kernel(const SearchArgs A) {
  __shared__ unsigned int hist_all[4][256];
  const letkf_search_tables& t = A.t;
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  unsigned int* hist = hist_all[wv];

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
            for (int b = 0; b < 256; ++b) {                    // wave-uniform scan of the 256 bins
              const int hb = (int)hist[b];
              if (cum + hb >= want) {
                bsel = b;
                break;
              }
              cum += hb;
            }
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

hipError_t launch_search(const SearchArgs& a, int num_cu, hipStream_t st) {
  const long nwg = (a.npts + 3) / 4;
  const long g = (long)num_cu * 8;
  const int grid = (int)(nwg < g ? (nwg > 0 ? nwg : 1) : g);
  hipLaunchKernelGGL(letkf_search_kernel, dim3(grid), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace letkf
