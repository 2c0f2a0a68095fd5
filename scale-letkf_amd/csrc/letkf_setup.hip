// letkf_setup.hip -- what das_letkf derives before its loop (C ABI section 7):
// the variable-localisation classes, the merge groups of the obs-number limit and radar_only on the host (tens of
// integers, once per analysis), relax_beta and the inflation-field initialisation as streaming passes on the device.
// scale/letkf/letkf_tools.f90:130-267, :1911-1948.

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>

#include "letkf_device.h"

namespace letkf {

// relax_beta, letkf_tools.f90:1911-1948; dist_zero_fac is the single-precision literal of letkf_obs.f90:27
__global__ void relax_beta_kernel(const letkf_beta_params P, long nij1, long npts, const double* __restrict__ rig,
                                  const double* __restrict__ rjg, const double* __restrict__ hgt,
                                  double* __restrict__ beta) {
  const double zcut = P.radar_zmax + P.vert_local_radar * (double)3.651483717f;
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < npts; p += (long)gridDim.x * blockDim.x) {
    const long ij = p % nij1;
    double b = 1.0;
    if (P.radar_only && hgt[p] > zcut) {
      b = 0.0;
    } else if (P.boundary_buffer_width > 0.0) {
      const double ri = rig[ij], rj = rjg[ij];
      const double di = fmin(ri - (double)P.ihalo, (double)(P.nlong + P.ihalo + 1) - ri) * P.dx;
      const double dj = fmin(rj - (double)P.jhalo, (double)(P.nlatg + P.jhalo + 1) - rj) * P.dy;
      const double dist_bdy = fmin(di, dj) / P.boundary_buffer_width;
      if (dist_bdy < 1.0) b = fmax(dist_bdy, 0.0);
    }
    beta[p] = b;
  }
}

// letkf_tools.f90:237-267
__global__ void infl_init_kernel(long n, double* __restrict__ w, double infl_mul, double infl_mul_min) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    double v = infl_mul > 0.0 ? infl_mul : w[e];
    if (infl_mul_min > 0.0) v = fmax(v, infl_mul_min);
    w[e] = v;
  }
}

static unsigned stream_grid(long n, int num_cu) {
  long g = (n + 255) / 256;
  const long cap = (long)num_cu * 32;
  return (unsigned)(g < 1 ? 1 : g > cap ? cap : g);
}

hipError_t launch_relax_beta(const letkf_beta_params& p, long nij1, int nlev, const double* rig, const double* rjg,
                             const double* hgt, double* beta, int num_cu, hipStream_t st) {
  const long npts = nij1 * nlev;
  hipLaunchKernelGGL(relax_beta_kernel, dim3(stream_grid(npts, num_cu)), dim3(256), 0, st, p, nij1, npts, rig, rjg, hgt,
                     beta);
  return hipGetLastError();
}

hipError_t launch_infl_init(long n, double* w, double infl_mul, double infl_mul_min, int num_cu, hipStream_t st) {
  hipLaunchKernelGGL(infl_init_kernel, dim3(stream_grid(n, num_cu)), dim3(256), 0, st, n, w, infl_mul, infl_mul_min);
  return hipGetLastError();
}

}  // namespace letkf

extern "C" {

// letkf_tools.f90:139-157
int letkf_var_local_classes(int32_t nvar, int32_t nlt, const double* var_local, int32_t* n2nc, int32_t* n2n,
                            int32_t* nclass) {
  if (nvar < 1 || nlt < 1 || !var_local || !n2nc || !n2n || !nclass) return LETKF_E_INVALID;
  int nc = 1;
  n2nc[0] = 0;
  n2n[0] = 0;
  for (int n = 1; n < nvar; ++n) {
    bool found = false;
    for (int i = 0; i < nc && !found; ++i) {
      // the reference compares against row var_local_n2nc(i) -- the class NUMBER used as a variable index (:143);
      // restated as written: class i is looked up through the class id of variable i
      const int rep = n2nc[i];
      double mx = 0.0;
      for (int t = 0; t < nlt; ++t) mx = fmax(mx, fabs(var_local[rep + (long)nvar * t] - var_local[n + (long)nvar * t]));
      if (mx < DBL_MIN) {                          // tiny(var_local)
        n2nc[n] = n2nc[i];
        n2n[n] = n2n[n2nc[n]];
        found = true;
      }
    }
    if (!found) {
      n2nc[n] = nc++;
      n2n[n] = n;
    }
  }
  *nclass = nc;
  return LETKF_OK;
}

// letkf_tools.f90:167-192
int letkf_ctype_merge_groups(int32_t nctype, const int32_t* elm_u_ctype, const int32_t* typ_ctype, int32_t nid_obs,
                             int32_t nobtype, const int32_t* ctype_merge, int32_t* group_start, int32_t* group_member,
                             int32_t* ngroup) {
  if (nctype < 0 || nid_obs < 1 || nobtype < 1 || !ngroup || !group_start) return LETKF_E_INVALID;
  if (nctype > 0 && (!elm_u_ctype || !typ_ctype || !ctype_merge || !group_member)) return LETKF_E_INVALID;
  for (int ic = 0; ic < nctype; ++ic)
    if (elm_u_ctype[ic] < 1 || elm_u_ctype[ic] > nid_obs || typ_ctype[ic] < 1 || typ_ctype[ic] > nobtype)
      return LETKF_E_INVALID;
  auto merge_of = [&](int ic) { return ctype_merge[(elm_u_ctype[ic] - 1) + (long)nid_obs * (typ_ctype[ic] - 1)]; };
  int ng = 0, pos = 0;
  // n_merge(ic) == 0 marks a ctype that an earlier master took (:176-181); kept in group_start's tail as scratch
  // would alias the output, so a small bitmap on the stack / heap it is
  bool* taken = new bool[nctype > 0 ? nctype : 1]();
  group_start[0] = 0;
  for (int ic = 0; ic < nctype; ++ic) {
    if (taken[ic]) continue;
    group_member[pos++] = ic;
    if (merge_of(ic) > 0)
      for (int ic2 = ic + 1; ic2 < nctype; ++ic2)
        if (merge_of(ic2) == merge_of(ic)) {
          group_member[pos++] = ic2;
          taken[ic2] = true;
        }
    group_start[++ng] = pos;
  }
  delete[] taken;
  *ngroup = ng;
  return LETKF_OK;
}

// letkf_tools.f90:197-203
int letkf_radar_only(int32_t nctype, const int32_t* typ_ctype, int32_t typ_radar) {
  for (int ic = 0; ic < nctype; ++ic)
    if (typ_ctype[ic] != typ_radar) return 0;
  return 1;
}

}  // extern "C"
