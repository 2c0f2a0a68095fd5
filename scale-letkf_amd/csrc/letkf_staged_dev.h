// letkf_staged_dev.h -- what the kernels of the staged path (letkf_staged.hip: Gram and apply stage; letkf_krylov.hip: the
// eigen-free stage) share: the per-point workspace slab and the view of a point's local observations.  Device code only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "letkf_device.h"

namespace letkf {
namespace staged_dev {

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}

constexpr int kMaxNb = 16;     // right-hand sides: nv + 2 <= 16

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));   // two doubles fetched as one 16-byte load from an 8-byte aligned address

// the points letkf_gram.hip's matrix-core Gram stage takes (loop body with lists): orders up to 512 (32 blocks of 16)
__host__ __device__ inline bool gram_mfma_takes(int n, int k) { return n >= 1 && (n < k ? n <= 512 : k <= 512); }

// Leading dimension of a point's m x m matrix in its slab (column c at c * staged_ld(m)): rows of 128-byte multiples for the
// orders whose matrices stream from beyond L2 in the eigen-free stage (A/B r3: C3-slab +2.3 %, C5-slab +1.8 %), m | 1 -- what
// rounds 1 and 2 used everywhere -- below that (k = 100: the padding to 112 cost 1.3 %).
#ifndef STAGED_LD_ALIGN
#define STAGED_LD_ALIGN 16
#endif
__host__ __device__ inline int staged_ld(int m) {
  return (STAGED_LD_ALIGN && m > 128) ? ((m + STAGED_LD_ALIGN - 1) / STAGED_LD_ALIGN) * STAGED_LD_ALIGN : (m | 1);
}

// Slab of one point of a batch (doubles; ldg = k | 1, nb = nv + 2):
//   G [(k + 1) ldg] | V0 [k] | V1 [k] | SC [16] | X [nv k] | TT [nb k] | PC [nb (k + 2)] | QQ [nb k] | OUT [nb k] | (W [k ldg]) | (H [hist])
// H: the residual history of the eigen-free stage (letkf_krylov.hip), lane-private layout.
struct Slab {
  double *G, *V0, *V1, *SC, *X, *TT, *PC, *QQ, *OUT, *W, *H;
};
__device__ __forceinline__ Slab slab_of(double* base, int k, int nv, int kkout) {
  const int ldg = staged_ld(k), nb = nv + 2;
  Slab s;
  s.G = base;
  s.V0 = s.G + (size_t)(k + 1) * ldg;   // one spare column: an odd order is padded with an inert zero column (letkf_eig.hip)
  s.V1 = s.V0 + k;
  s.SC = s.V1 + k;
  s.X = s.SC + 16;
  s.TT = s.X + (size_t)nv * k;
  s.PC = s.TT + (size_t)nb * k;
  s.QQ = s.PC + (size_t)nb * (k + 2);   // PC rows are k + 2 long (coefficients of up to k + 1 stored columns)
  s.OUT = s.QQ + (size_t)nb * k;
  s.W = kkout ? s.OUT + (size_t)nb * k : nullptr;
  const size_t hoff = (size_t)(s.OUT - base) + (size_t)nb * k + (kkout ? (size_t)k * ldg : 0);
  s.H = base + ((hoff + 15) & ~(size_t)15);   // (128-byte aligned: slabs are a multiple of 16 doubles apart, staged_ws_per_point)
  return s;
}

// where the point's observations come from: the obs table through the CSR lists (das_letkf body) or a dense
// column-major hdxb(nobs, ne) block (letkf_core batch)
struct ObsView {
  const PointArgs* A;
  long pt, o0;
  int n;
  __device__ __forceinline__ void weights(int i, double& w, double& d, double& dd, double& rl) const {
    if (A->mode == 0) {
      const long e = o0 + i;
      const int iob = A->obs_idx[e];
      rl = A->rloc_l[e];
      w = 1.0 / A->rdiag_l[e];
      d = A->dep[iob];
      dd = A->det_run ? A->ensval[(long)iob * A->kld + A->k] : 0.0;
    } else {
      const long e = pt * (long)A->nobs + i;
      rl = A->rloc[e];
      w = A->rdiag_wloc ? 1.0 / A->rdiag[e] : rl / A->rdiag[e];   // common_letkf.f90:111-123
      d = A->depv[e];
      dd = A->depd ? A->depd[e] : 0.0;
    }
  }
  // address of y_i[0] and the member stride
  __device__ __forceinline__ const double* row(int i, long& ms) const {
    if (A->mode == 0) {
      ms = 1;
      return A->ensval + (long)A->obs_idx[o0 + i] * A->kld;
    }
    ms = A->nobs;
    return A->hdxb + (size_t)pt * (size_t)A->nobs * (size_t)A->k + i;
  }
};

}  // namespace staged_dev
}  // namespace letkf
