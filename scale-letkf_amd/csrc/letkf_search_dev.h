// letkf_search_dev.h -- device functions shared by the search kernel (letkf_search.hip) and the fused search of the
// wave kernel (letkf_wave.hip): obs_local_cal (scale/letkf/letkf_tools.f90:1793-1906), ij_obsgrd_ext
// (scale/letkf/letkf_obs.f90:1209-1227) and the single-precision cut-off literals of letkf_obs.f90:27-28.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/letkf_amd.h"

namespace letkf {
namespace search_dev {

constexpr double kDistZeroFac = (double)3.651483717f;          // letkf_obs.f90:27 (single-precision literal)
constexpr double kDistZeroFacSq = (double)13.33333333f;        // :28
constexpr double kTiny = 2.2250738585072014e-308;              // tiny(var_local)

struct CalOut {
  double rloc, rdiag, ndist;
};

// scale/letkf/letkf_tools.f90:1793-1906, on the observation's metadata already in registers
__device__ __forceinline__ CalOut local_cal_v(const letkf_search_tables& t, int ic, double ri, double rj, double rlev,
                                              double rz, double ob_lev, double ob_dat, double ob_ri, double ob_rj,
                                              double ob_err) {
  // no multiply-add fusion here: the function is inlined into two different kernels (the stand-alone search and the
  // fused search of the wave kernel), and their weights must agree to the last bit for the two paths to be identical
#pragma clang fp contract(off)
  CalOut o{0.0, -1.0, -1.0};
  double nrloc = t.varloc[ic];                                 // :1840
  if (nrloc < kTiny) return o;                                 // :1843
  const double vloc = t.vert_loc[ic];
  double nd_v;
  const int vm = t.vmode[ic];
  if (vloc == 0.0) nd_v = 0.0;                                 // :1851-1865
  else if (vm == 2) nd_v = fabs(log(ob_dat) - log(rlev)) / vloc;
  else if (vm == 3) nd_v = fabs(log(t.rain_base) - log(rlev)) / vloc;
  else if (vm == 1) nd_v = fabs(ob_lev - rz) / vloc;
  else nd_v = fabs(log(ob_lev) - log(rlev)) / vloc;
  if (nd_v > kDistZeroFac) return o;                           // :1869
  const double rdx = (ri - ob_ri) * t.dx;                      // :1876-1878
  const double rdy = (rj - ob_rj) * t.dy;
  const double nd_h = sqrt(rdx * rdx + rdy * rdy) / t.hori_loc[ic];
  if (nd_h > kDistZeroFac) return o;                           // :1881
  const double nd = nd_h * nd_h + nd_v * nd_v;                 // :1888
  if (nd > kDistZeroFacSq) return o;                           // :1891
  nrloc = nrloc * exp(-0.5 * nd);                              // :1899
  o.rloc = nrloc;
  o.rdiag = ob_err * ob_err / nrloc;                           // :1903
  o.ndist = nd;
  return o;
}

// The vertical half of obs_local_cal for an observation whose horizontal half is done (the column search: nd_h and the
// observation's vertical coordinate v_obs -- lev, ln lev or ln dat by the ctype's mode -- are the same for every level of a
// column).  v_z = the point's height, v_p = ln of its pressure, l_rain = ln VERT_LOCAL_RAIN_BASE.  rloc = 0: rejected.
// Shared by the column search (lists) and the wave kernel's column-survivor mode (no lists): same weights to the last bit.
struct ColVert {
  double rloc, rdiag;
};
__device__ __forceinline__ ColVert column_vertical_cal(const int vm, const double vloc, const double varloc, const double nd_h,
                                                const double v_obs, const double err, const double v_z, const double v_p,
                                                const double l_rain) {
#pragma clang fp contract(off)
  ColVert o{0.0, 0.0};
  double nd_v;
  if (vloc == 0.0) nd_v = 0.0;                                 // :1851-1865
  else if (vm == 3) nd_v = fabs(l_rain - v_p) / vloc;
  else nd_v = fabs(v_obs - (vm == 1 ? v_z : v_p)) / vloc;
  if (nd_v > kDistZeroFac) return o;                           // :1869
  const double nd = nd_h * nd_h + nd_v * nd_v;                 // :1888
  if (nd > kDistZeroFacSq) return o;                           // :1891
  o.rloc = varloc * exp(-0.5 * nd);                            // :1899
  o.rdiag = err * err / o.rloc;                                // :1903
  return o;
}

__device__ __forceinline__ CalOut local_cal(const letkf_search_tables& t, int ic, double ri, double rj, double rlev,
                                            double rz, int row) {
  const int vm = t.vmode[ic];
  return local_cal_v(t, ic, ri, rj, rlev, rz, (vm == 2 || vm == 3) ? 1.0 : t.ob_lev[row], vm == 2 ? t.ob_dat[row] : 1.0,
                     t.ob_ri[row], t.ob_rj[row], t.ob_err[row]);
}

__device__ __forceinline__ void ij_obsgrd_ext(const letkf_search_tables& t, int ic, double ri, double rj, int& ogi,
                                              int& ogj) {       // letkf_obs.f90:1221-1224
  ogi = (int)ceil((ri - t.i_org) * (double)t.ngrd_i[ic] / (double)t.nlon) + t.ngrdsch_i[ic];
  ogj = (int)ceil((rj - t.j_org) * (double)t.ngrd_j[ic] / (double)t.nlat) + t.ngrdsch_j[ic];
}


}  // namespace search_dev
}  // namespace letkf
