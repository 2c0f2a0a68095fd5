// letkf_obsprep.hip -- set_letkf_obs on the device (scale/letkf/letkf_obs.f90):
//   departure + QC            :361-561   one pass over the local H(x) table, row = observation, member-fastest
//   bucket sort onto the mesh :762-822   (ctype, mesh j, mesh i, obs number) radix sort: the stable counting sort
//                                        of the reference, as 64-bit keys (rocPRIM device radix sort)
//   extended-subdomain tables :922-976   host plan from every rank's cell counts (small integer tables)
//   obsda_sort copy           :1036-1100 row gather through the plan's source-row map
// All integer / index results are bit-identical to the reference's loops; the departures use the same sequential
// member sum, so they are bit-identical too.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>

#include "letkf_device.h"

namespace letkf {

namespace {

// common_obs_scale.f90:48-72, :139-151; common/common.f90:38
constexpr int kIdRain = 19999, kIdRadarRef = 4001, kIdRadarRefZero = 4004, kIdRadarVr = 4002, kIdRadarPrh = 4003,
              kIdTclon = 99991, kIdTclat = 99992, kIdTcmip = 99993, kIdH08IR = 8800;
constexpr int kQcGood = 0, kQcGross = 5, kQcRefMem = 12, kQcObsBad = 50, kQcOtype = 90;
constexpr double kUndef = -9.99e33;

// One wave per 64 observations: the rows (kld doubles each, contiguous) are read and written back in 512-B
// coalesced segments through LDS; lane i then owns row i and walks it with the reference's sequential member sum.
constexpr int kDepRows = 64;
__global__ void __launch_bounds__(64) departure_kernel(const letkf_qc_params P, const long nobs, const int* __restrict__ elm,
                                                       const double* __restrict__ dat, const double* __restrict__ err,
                                                       double* __restrict__ ensval, const long kld,
                                                       double* __restrict__ val, int* __restrict__ qc) {
  extern __shared__ double rows[];                       // [64][ldr], ldr odd: conflict-free row walks
  const int lane = threadIdx.x;
  const long ldr = kld | 1;
  for (long n0 = (long)blockIdx.x * kDepRows; n0 < nobs; n0 += (long)gridDim.x * kDepRows) {
    const long nr = min((long)kDepRows, nobs - n0);
    const long tot = nr * kld;
    const double* src = ensval + n0 * kld;
    for (long t = lane; t < tot; t += 64) rows[(t / kld) * ldr + (t % kld)] = src[t];
    __syncthreads();
    if (lane < nr) {
      const long n = n0 + lane;
      double* e = rows + lane * ldr;
      int q = qc[n];
      if (q <= 0) {                                      // letkf_obs.f90:362
        const int K = P.member;
        const int el = elm[n];
        const double d = dat[n];
        bool live = true;
        if (el == kIdRadarRef || el == kIdRadarRefZero) {            // :372-411
          if (!P.use_radar_ref) {
            q = kQcOtype;
            live = false;
          } else if (d == kUndef) {
            q = kQcObsBad;
            live = false;
          } else {
            int mem_ref = 0;
            for (int i = 0; i < K; ++i) mem_ref += e[i] > P.radar_ref_thres_dbz + 1.0e-6 ? 1 : 0;
            const int need = d > P.radar_ref_thres_dbz + 1.0e-6 ? P.min_radar_ref_member_obsref : P.min_radar_ref_member;
            if (mem_ref < need) {
              q = kQcRefMem;
              live = false;
            }
          }
        }
        if (live && el == kIdRadarVr && !P.use_radar_vr) {           // :413-418
          q = kQcOtype;
          live = false;
        }
        int mem_cld = 0;
        if (live && P.h08 && el == kIdH08IR) {                       // -DH08, :432-469
          if (d == kUndef || P.h08_lev[n] < P.h08_limit_lev) {
            q = kQcObsBad;
            live = false;
          } else {
            for (int i = 0; i < K; ++i)
              if (e[i] < 0.0) {                                      // cloudy members carry negative values
                ++mem_cld;
                e[i] = e[i] * (-1.0);
              }
          }
        }
        if (live) {
          double v = e[0];                                             // :475-479
          for (int i = 1; i < K; ++i) v = v + e[i];
          v = v / (double)K;
          if (P.h08 && P.h08_val2) {                                   // -DH08, :480-487: CA (Okamoto et al. 2014)
            const double clr = P.h08_val2[n];
            P.h08_val2[n] = (fabs(v - clr) + fabs(d - clr)) * 0.5;
          }
          for (int i = 0; i < K; ++i) e[i] = e[i] - v;                 // :488-490
          v = d - v;                                                   // :491
          val[n] = v;
          if (P.det_run) e[K] = d - e[K];                              // :492-494
          double ge;                                                   // :504-561
          switch (el) {
            case kIdRain: ge = P.gross_error_rain; break;
            case kIdRadarRef:
            case kIdRadarRefZero: ge = P.gross_error_radar_ref; break;
            case kIdRadarVr: ge = P.gross_error_radar_vr; break;
            case kIdRadarPrh: ge = P.gross_error_radar_prh; break;
            case kIdTclon: ge = P.gross_error_tcx; break;
            case kIdTclat: ge = P.gross_error_tcy; break;
            case kIdTcmip: ge = P.gross_error_tcp; break;
            default: ge = P.gross_error;
          }
          if (P.h08 && el == kIdH08IR) {                               // -DH08, :520-541
            ge = mem_cld < P.h08_min_cld_member ? 1.0 : P.gross_error_h08;
            if (fabs(v) > ge * err[n]) q = kQcGross;
            if (d < P.h08_bt_min) q = kQcGross;
          } else if (fabs(v) > ge * err[n]) {
            q = kQcGross;
          }
        }
        qc[n] = q;
      }
    }
    __syncthreads();
    double* dst = ensval + n0 * kld;
    for (long t = lane; t < tot; t += 64) dst[t] = rows[(t / kld) * ldr + (t % kld)];
    __syncthreads();
  }
}

struct MeshDev {
  int nctype, nlon, nlat, ihalo, jhalo, rank_i, rank_j, fix_j;
  const int* ngrd_i;      // device copies
  const int* ngrd_j;
  const long* coff;       // [nctype + 1] cell offsets
};

// ij_obsgrd, letkf_obs.f90:1186-1203 with the clamps of :768-771; the reference scales rj by ngrd_i (:1200) -- restated
// as written unless the caller asks for ngrd_j (letkf_mesh.fix_ij_obsgrd: what ij_obsgrd_ext, :1223, looks up with)
__device__ __forceinline__ long mesh_cell(const MeshDev& m, int ic, double ri, double rj) {
  const double ril = ri - (double)(m.rank_i * m.nlon);
  const double rjl = rj - (double)(m.rank_j * m.nlat);
  const int gi = m.ngrd_i[ic], gj = m.ngrd_j[ic];
  int i = (int)ceil((ril - (double)m.ihalo - 0.5) * (double)gi / (double)m.nlon);
  int j = (int)ceil((rjl - (double)m.jhalo - 0.5) * (double)(m.fix_j ? gj : gi) / (double)m.nlat);
  i = i < 1 ? 1 : (i > gi ? gi : i);
  j = j < 1 ? 1 : (j > gj ? gj : j);
  return m.coff[ic] + (long)(j - 1) * gi + (i - 1);
}

__global__ void mesh_key_kernel(const MeshDev m, const long nobs, const int* __restrict__ ctype,
                                const double* __restrict__ ri, const double* __restrict__ rj,
                                const int* __restrict__ qc, unsigned long long* __restrict__ keys,
                                int* __restrict__ n_cell) {
  for (long n = (long)blockIdx.x * blockDim.x + threadIdx.x; n < nobs; n += (long)gridDim.x * blockDim.x) {
    unsigned long long key = ~0ull;                      // rejected observations sort to the end
    if (qc[n] == kQcGood) {
      const long c = mesh_cell(m, ctype[n], ri[n], rj[n]);
      key = ((unsigned long long)c << 32) | (unsigned long long)(unsigned int)n;
      atomicAdd(&n_cell[c], 1);
    }
    keys[n] = key;
  }
}

__global__ void key_low_kernel(const long n, const unsigned long long* __restrict__ keys, int* __restrict__ key) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x)
    key[t] = (int)(unsigned int)(keys[t] & 0xFFFFFFFFull);
}

// one wave per extended cell: src_row[dst0 + t] = src0 + t
__global__ void expand_plan_kernel(const long ncellx, const int* __restrict__ dst0, const int* __restrict__ src0,
                                   const int* __restrict__ len, int* __restrict__ src_row) {
  const int lane = threadIdx.x & 63;
  const long w0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nw = ((long)gridDim.x * blockDim.x) >> 6;
  for (long c = w0; c < ncellx; c += nw) {
    const int l = len[c], d = dst0[c], s = src0[c];
    for (int t = lane; t < l; t += 64) src_row[d + t] = s + t;
  }
}

__global__ void gather_rows_kernel(const long nrows, const int* __restrict__ src_row, const int ncols,
                                   const double* __restrict__ src, const long ld_src, double* __restrict__ dst,
                                   const long ld_dst) {
  // consecutive threads walk along a row: coalesced on both sides when ncols is a row of the obs table
  const long tot = nrows * (long)ncols;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < tot; t += (long)gridDim.x * blockDim.x) {
    const long r = t / ncols;
    const int c = (int)(t - r * ncols);
    dst[r * ld_dst + c] = src[(long)src_row[r] * ld_src + c];
  }
}

__global__ void gather_i32_kernel(const long nrows, const int* __restrict__ src_row, const int* __restrict__ src,
                                  int* __restrict__ dst) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < nrows; t += (long)gridDim.x * blockDim.x)
    dst[t] = src[src_row[t]];
}

inline int grid_for(long n, int block, int num_cu) {
  long g = (n + block - 1) / block;
  const long cap = (long)num_cu * 16;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

hipError_t launch_obs_departure(const letkf_qc_params& p, long nobs, const int* elm, const double* dat, const double* err,
                                double* ensval, long kld, double* val, int* qc, int num_cu, hipStream_t st) {
  if (nobs <= 0) return hipSuccess;
  const size_t lds = (size_t)kDepRows * (size_t)(kld | 1) * sizeof(double);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&departure_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  const int grid = grid_for((nobs + kDepRows - 1) / kDepRows, 1, num_cu);
  hipLaunchKernelGGL(departure_kernel, dim3(grid), dim3(64), lds, st, p, nobs, elm, dat, err, ensval, kld, val, qc);
  return hipGetLastError();
}

// Scratch layout (bytes) for the sort: [keys_in nobs*8][keys_out nobs*8][ngrd_i, ngrd_j: nctype*4 each][coff (nctype+1)*8]
// [rocprim temp].  Returns the bytes needed when scratch == nullptr.
hipError_t obs_mesh_sort(const letkf_mesh& m, long nobs, const int* ctype, const double* ri, const double* rj,
                         const int* qc, int* n_cell, int* key, long* nsorted, void* scratch, size_t* scratch_bytes,
                         int num_cu, hipStream_t st) {
  std::vector<long> coff(m.nctype + 1, 0);
  for (int ic = 0; ic < m.nctype; ++ic) coff[ic + 1] = coff[ic] + (long)m.ngrd_i[ic] * m.ngrd_j[ic];
  const long ncell = coff[m.nctype];
  int cell_bits = 1;
  while ((1L << cell_bits) < ncell + 1) ++cell_bits;
  size_t temp_bytes = 0;
  hipError_t e = rocprim::radix_sort_keys(nullptr, temp_bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                          (size_t)(nobs > 0 ? nobs : 1), 0, 32 + cell_bits, st);
  if (e != hipSuccess) return e;
  const size_t a_keys = ((size_t)(nobs > 0 ? nobs : 1) * 8 + 255) & ~(size_t)255;
  const size_t a_tab = ((size_t)m.nctype * 8 + (size_t)(m.nctype + 1) * 8 + 255) & ~(size_t)255;
  const size_t need = 2 * a_keys + a_tab + temp_bytes + 256;
  if (!scratch) {
    *scratch_bytes = need;
    return hipSuccess;
  }
  if (*scratch_bytes < need) return hipErrorInvalidValue;
  char* base = static_cast<char*>(scratch);
  unsigned long long* kin = reinterpret_cast<unsigned long long*>(base);
  unsigned long long* kout = reinterpret_cast<unsigned long long*>(base + a_keys);
  int* d_gi = reinterpret_cast<int*>(base + 2 * a_keys);
  int* d_gj = d_gi + m.nctype;
  long* d_coff = reinterpret_cast<long*>(base + 2 * a_keys + (size_t)m.nctype * 8);
  void* temp = base + 2 * a_keys + a_tab;
  if ((e = hipMemcpyAsync(d_gi, m.ngrd_i, sizeof(int) * m.nctype, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
  if ((e = hipMemcpyAsync(d_gj, m.ngrd_j, sizeof(int) * m.nctype, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
  if ((e = hipMemcpyAsync(d_coff, coff.data(), sizeof(long) * (m.nctype + 1), hipMemcpyHostToDevice, st)) != hipSuccess)
    return e;
  if ((e = hipMemsetAsync(n_cell, 0, sizeof(int) * (size_t)(ncell > 0 ? ncell : 1), st)) != hipSuccess) return e;
  *nsorted = 0;
  if (nobs <= 0) return hipStreamSynchronize(st);
  MeshDev md{m.nctype, m.nlon, m.nlat, m.ihalo, m.jhalo, m.rank_i, m.rank_j, m.fix_ij_obsgrd != 0, d_gi, d_gj, d_coff};
  hipLaunchKernelGGL(mesh_key_kernel, dim3(grid_for(nobs, 256, num_cu)), dim3(256), 0, st, md, nobs, ctype, ri, rj, qc,
                     kin, n_cell);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if ((e = rocprim::radix_sort_keys(temp, temp_bytes, kin, kout, (size_t)nobs, 0, 32 + cell_bits, st)) != hipSuccess)
    return e;
  hipLaunchKernelGGL(key_low_kernel, dim3(grid_for(nobs, 256, num_cu)), dim3(256), 0, st, nobs, kout, key);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  // number of accepted observations = total of the cell counts
  std::vector<int> h(ncell > 0 ? ncell : 1, 0);
  if ((e = hipMemcpyAsync(h.data(), n_cell, sizeof(int) * (size_t)ncell, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
  if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
  long tot = 0;
  for (long c = 0; c < ncell; ++c) tot += h[c];
  *nsorted = tot;
  return hipSuccess;
}

// Host plan (letkf_obs.f90:922-976, :1036-1100) + device expansion into src_row.  n_all: device [nprocs][ncell].
hipError_t obs_halo_plan(const letkf_halo_layout& l, const int* n_all, int* ac_ext, int* src_row, long cap,
                         long* nobstotal, int num_cu, hipStream_t st) {
  const int nc = l.nctype, np = l.nprocs;
  std::vector<long> coff(nc + 1, 0), xoff(nc + 1, 0), ecell(nc + 1, 0);
  for (int ic = 0; ic < nc; ++ic) {
    const long ei = l.ngrd_i[ic] + 2L * l.ngrdsch_i[ic], ej = l.ngrd_j[ic] + 2L * l.ngrdsch_j[ic];
    coff[ic + 1] = coff[ic] + (long)l.ngrd_i[ic] * l.ngrd_j[ic];
    xoff[ic + 1] = xoff[ic] + (ei + 1) * ej;
    ecell[ic + 1] = ecell[ic] + ei * ej;
  }
  const long ncell = coff[nc], ncellx = ecell[nc];
  std::vector<int> h((size_t)np * (ncell > 0 ? ncell : 1), 0);
  hipError_t e;
  if (ncell > 0) {
    if ((e = hipMemcpyAsync(h.data(), n_all, sizeof(int) * (size_t)np * ncell, hipMemcpyDeviceToHost, st)) != hipSuccess)
      return e;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
  }
  std::vector<long> start((size_t)np * (ncell > 0 ? ncell : 1), 0), dspr(np + 1, 0);
  for (int ip = 0; ip < np; ++ip) {                      // ac(:,:,ip) and dspr (:979-984)
    long acc = 0;
    for (long c = 0; c < ncell; ++c) {
      start[(size_t)ip * ncell + c] = acc;
      acc += h[(size_t)ip * ncell + c];
    }
    dspr[ip + 1] = dspr[ip] + acc;
  }
  const int myp_i = l.myrank % l.prc_num_x, myp_j = l.myrank / l.prc_num_x;
  std::vector<int> h_ac(xoff[nc] > 0 ? xoff[nc] : 1, 0), dst0(ncellx > 0 ? ncellx : 1, 0), src0(dst0.size(), 0),
      len(dst0.size(), 0);
  long acx = 0;
  for (int ic = 0; ic < nc; ++ic) {
    const int gi = l.ngrd_i[ic], gj = l.ngrd_j[ic], si = l.ngrdsch_i[ic], sj = l.ngrdsch_j[ic];
    const int ei = gi + 2 * si, ej = gj + 2 * sj;
    int* nx = len.data() + ecell[ic];
    int* sx = src0.data() + ecell[ic];
    const int imin1 = myp_i * gi + 1 - si, imax1 = (myp_i + 1) * gi + si;     // :925-928
    const int jmin1 = myp_j * gj + 1 - sj, jmax1 = (myp_j + 1) * gj + sj;
    for (int ip = 0; ip < np; ++ip) {
      const int ip_i = ip % l.prc_num_x, ip_j = ip / l.prc_num_x;
      const int imin2 = std::max(1, imin1 - ip_i * gi), imax2 = std::min(gi, imax1 - ip_i * gi);   // :932-936
      const int jmin2 = std::max(1, jmin1 - ip_j * gj), jmax2 = std::min(gj, jmax1 - ip_j * gj);
      if (imin2 > imax2 || jmin2 > jmax2) continue;
      const int ishift = (ip_i - myp_i) * gi + si, jshift = (ip_j - myp_j) * gj + sj;               // :938-940
      for (int j = jmin2; j <= jmax2; ++j)
        for (int i = imin2; i <= imax2; ++i) {
          const long c = coff[ic] + (long)(j - 1) * gi + (i - 1);
          const long x = (long)(j + jshift - 1) * ei + (i + ishift - 1);
          nx[x] = h[(size_t)ip * ncell + c];
          sx[x] = (int)(dspr[ip] + start[(size_t)ip * ncell + c]);
        }
    }
    int* ax = h_ac.data() + xoff[ic];                     // [ej][ei + 1], cumulative over ctypes (:946-956)
    int* d0 = dst0.data() + ecell[ic];
    for (int j = 0; j < ej; ++j) {
      ax[(long)j * (ei + 1)] = (int)acx;
      for (int i = 0; i < ei; ++i) {
        d0[(long)j * ei + i] = (int)acx;
        acx += nx[(long)j * ei + i];
        ax[(long)j * (ei + 1) + i + 1] = (int)acx;
      }
    }
  }
  *nobstotal = acx;
  if ((e = hipMemcpyAsync(ac_ext, h_ac.data(), sizeof(int) * (size_t)xoff[nc], hipMemcpyHostToDevice, st)) != hipSuccess)
    return e;
  if (acx > cap) return hipErrorInvalidValue;
  if (acx > 0) {
    // the three per-cell arrays ride in a temporary device buffer
    int* d_plan = nullptr;
    if ((e = hipMalloc(reinterpret_cast<void**>(&d_plan), sizeof(int) * 3 * (size_t)ncellx)) != hipSuccess) return e;
    (void)hipMemcpyAsync(d_plan, dst0.data(), sizeof(int) * (size_t)ncellx, hipMemcpyHostToDevice, st);
    (void)hipMemcpyAsync(d_plan + ncellx, src0.data(), sizeof(int) * (size_t)ncellx, hipMemcpyHostToDevice, st);
    (void)hipMemcpyAsync(d_plan + 2 * ncellx, len.data(), sizeof(int) * (size_t)ncellx, hipMemcpyHostToDevice, st);
    hipLaunchKernelGGL(expand_plan_kernel, dim3(grid_for(ncellx * 64, 256, num_cu)), dim3(256), 0, st, ncellx, d_plan,
                       d_plan + ncellx, d_plan + 2 * ncellx, src_row);
    e = hipGetLastError();
    hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(d_plan);
    if (e != hipSuccess) return e;
    if (e2 != hipSuccess) return e2;
  } else {
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
  }
  return hipSuccess;
}

hipError_t launch_gather_rows(long nrows, const int* src_row, int ncols, const double* src, long ld_src, double* dst,
                              long ld_dst, int num_cu, hipStream_t st) {
  if (nrows <= 0 || ncols <= 0) return hipSuccess;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(nrows * ncols, 256, num_cu)), dim3(256), 0, st, nrows, src_row,
                     ncols, src, ld_src, dst, ld_dst);
  return hipGetLastError();
}

hipError_t launch_gather_i32(long nrows, const int* src_row, const int* src, int* dst, int num_cu, hipStream_t st) {
  if (nrows <= 0) return hipSuccess;
  hipLaunchKernelGGL(gather_i32_kernel, dim3(grid_for(nrows, 256, num_cu)), dim3(256), 0, st, nrows, src_row, src, dst);
  return hipGetLastError();
}

}  // namespace letkf
