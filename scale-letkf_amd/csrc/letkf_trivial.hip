// letkf_trivial.hip -- the grid points of the das_letkf loop body (scale/letkf/letkf_tools.f90:313-527) that have nothing
// to solve, as ONE streaming pass with a thread per (point, variable):
//   * beta == 0 (letkf_tools.f90:333-359: above the radar top / outside the buffer zone): analysis = first guess;
//   * no local observation (common/common_letkf.f90:89-107): T = sqrt(rho) I, w-bar = 0 in closed form; what is left is
//     the relaxation (letkf_tools.f90:457-469, :1953-2002), the beta blend, the q clamp (:500-513) and the class copy of
//     the inflation slot (:396-398).
// The register kernels (letkf_wave.hip) give such a point a whole wavefront, lane = member: with the reference's
// point-fastest state (gues3d(nij1, nlev, nens, nv3d)) every one of its 11 x (k + 1) state values is an 8-byte access
// npts * 8 bytes from the next -- 70 KB of cache lines fetched and as many written per point for 4.5 KB of data, ~30 us
// of a wave's time, nothing to hide it behind.  On a domain whose observations sit in a radar disc most points are of
// this kind (bench workload C2-disc: 71 % of the points, a quarter of the kernel time).  Here consecutive lanes are
// consecutive POINTS (of one variable): they read consecutive addresses of every member plane, so the pass runs at HBM speed.
// PointArgs::skip_trivial tells the solve kernel that these points are done.
//
// The arithmetic follows the solve kernel's closed-form branch (letkf_wave.hip, `!solved`) statement by statement.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "letkf_device.h"

namespace letkf {

// One thread per (point, variable), points fastest: 11 x the threads of a thread-per-point pass and chains of 3 k instead
// of 33 k dependent accesses -- with few trivial points (C2-mini-disc: 3520 of them = 55 waves) the pass is bound by the
// latency of one thread's chain, not by bandwidth.  The member loops are unrolled by hand with their loads first: gues
// and anal are different buffers (trivial_pass_supports), which the compiler cannot know.
__global__ void __launch_bounds__(256) letkf_trivial_points_kernel(const PointArgs A) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nv = A.nv;
  if (idx >= A.npts * nv) return;
  const int v = (int)(idx / A.npts);
  const long pt = idx - (long)v * A.npts;
  const int n = (int)(A.obs_off[pt + 1] - A.obs_off[pt]);
  const double beta = A.beta ? A.beta[pt] : 1.0;
  if (n != 0 && beta != 0.0) return;
  const int k = A.k;
  const double km1 = (double)(k - 1);
  const bool inclass = (A.var_mask >> v) & 1u;
  const double* __restrict__ gv = A.gues + pt * A.sp + v * A.sv;
  double* __restrict__ av = A.anal + pt * A.sp + v * A.sv;
  const long sm = A.sm;
  constexpr int U = 10;                                // members in flight per thread

  if (beta == 0.0) {                                   // letkf_tools.f90:333-359
    if (inclass) {
      const double xm = gv[k * sm];
      for (int m0 = 0; m0 < k; m0 += U) {
        double x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = gv[(m0 + u < k ? m0 + u : k - 1) * sm];
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (m0 + u < k) av[(m0 + u) * sm] = xm + x[u];
      }
      if (A.det_run) av[(k + 1) * sm] = gv[(k + 1) * sm];
      if (A.rtps_out) A.rtps_out[pt + A.infl_sv * (long)v] = 1.0;
    }
    if (v == 0) {
      if (A.status) A.status[pt] = 0;
      if (A.nsweep) A.nsweep[pt] = 0;
      if (A.nobs_out) A.nobs_out[pt] = 0;
    }
    return;
  }

  bool qskip = false;
  if (A.q_update_top > 0.0) qskip = A.gues[pt * A.sp + k * sm + A.iv_p * A.sv] < A.q_update_top;
  // first variable of the class that is updated: its inflation slot is the point's rho (letkf_tools.f90:387-418)
  int v0 = 0;
  while (v0 < nv && (!((A.var_mask >> v0) & 1u) || (qskip && v0 >= A.iv_q_first && v0 <= A.iv_q_last))) ++v0;
  const double infl_old = v0 < nv ? A.infl[pt + A.infl_sv * (long)v0] : 1.0;
  const double lam = km1 / infl_old;                   // every eigenvalue of A = (k-1)/rho I
  const double sc1 = sqrt(km1 / lam);                  // T = sqrt(rho) I
  const double sc2 = 1.0 / lam;                        // Pa = rho/(k-1) I
  if (v == 0) {
    if (A.status) A.status[pt] = lam > 0.0 ? 0 : 2;    // common_mtx.f90:66-78
    if (A.nsweep) A.nsweep[pt] = 0;
    if (A.nobs_out) A.nobs_out[pt] = 0;
  }
  if (!inclass) return;

  const bool skip = qskip && v >= A.iv_q_first && v <= A.iv_q_last;
  const double xm = gv[k * sm];
  const double parm = A.relax_to_inflated_prior ? A.infl[pt + A.infl_sv * (long)v] : 1.0;   // :387-391, read before the update below
  double cfv = 1.0;
  if (A.relax_alpha != 0.0) {                          // RTPP (:1953-1966)
    cfv = 1.0 - A.relax_alpha;
  } else if (A.relax_alpha_spread != 0.0) {            // RTPS (:1971-2002) with Pa = sc2 I
    double var_g = 0.0;
    for (int m0 = 0; m0 < k; m0 += U) {
      double x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = gv[(m0 + u < k ? m0 + u : k - 1) * sm];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (m0 + u < k) var_g = fma(x[u], x[u], var_g);
    }
    const double var_a = var_g * sc2;
    if (var_g > 0.0 && var_a > 0.0)
      cfv = A.relax_alpha_spread * sqrt(var_g * parm / (var_a * km1)) - A.relax_alpha_spread + 1.0;
  }
  if (A.rtps_out)
    A.rtps_out[pt + A.infl_sv * (long)v] = (A.relax_alpha == 0.0 && A.relax_alpha_spread != 0.0 && !skip) ? cfv : 1.0;
  const double cdv = (!skip && A.relax_alpha != 0.0) ? A.relax_alpha * sqrt(parm) : 0.0;
  // one member's analysis value (letkf_tools.f90:472-487 with w-bar = 0)
  auto value = [&](const double x) { return skip ? xm + x : xm + beta * (cfv * (sc1 * x) + cdv * x) + (1.0 - beta) * x; };
  double q_mean = 0.0, q_sprd = 0.0;
  bool do_clamp = false;
  if (!skip && A.q_sprd_max > 0.0 && v == A.iv_q_first) {   // :500-513
    double s_ = 0.0;
    for (int m0 = 0; m0 < k; m0 += U) {
      double x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = gv[(m0 + u < k ? m0 + u : k - 1) * sm];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (m0 + u < k) s_ += value(x[u]);
    }
    q_mean = s_ / (double)k;
    double ss = 0.0;
    for (int m0 = 0; m0 < k; m0 += U) {
      double x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = gv[(m0 + u < k ? m0 + u : k - 1) * sm];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (m0 + u < k) {
          const double dq = value(x[u]) - q_mean;
          ss = fma(dq, dq, ss);
        }
    }
    q_sprd = sqrt(ss / km1) / q_mean;
    do_clamp = q_sprd > A.q_sprd_max;
  }
  for (int m0 = 0; m0 < k; m0 += U) {
    double x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = gv[(m0 + u < k ? m0 + u : k - 1) * sm];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (m0 + u < k) {
        double val = value(x[u]);
        if (do_clamp) val = q_mean + (val - q_mean) * A.q_sprd_max / q_sprd;
        av[(m0 + u) * sm] = val;
      }
  }
  if (A.det_run) av[(k + 1) * sm] = gv[(k + 1) * sm];   // :489-497 with w-bar_det = 0
  // :396-398: the class copies its first slot (no observation: the value is unchanged); every parm of this point was
  // read by the thread that writes it
  if (A.infl_adaptive && !skip) A.infl[pt + A.infl_sv * (long)v] = infl_old;
}

// which launches may hand their trivial points to this pass: the list-driven loop body without per-point matrix outputs
// (and with an analysis buffer of its own: the pass re-reads a variable's first guess after writing its analysis)
bool trivial_pass_supports(const PointArgs& a) {
  return a.mode == 0 && a.nv > 0 && a.gues && a.anal && a.gues != a.anal && !a.trans_out && !a.pa_out && !a.transm_out && !a.transmd_out;
}

hipError_t launch_trivial_points(const PointArgs& a, hipStream_t st) {
  const unsigned grid = (unsigned)((a.npts * a.nv + 255) / 256);
  if (grid == 0) return hipSuccess;
  hipLaunchKernelGGL(letkf_trivial_points_kernel, dim3(grid), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace letkf
