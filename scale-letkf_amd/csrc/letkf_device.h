// letkf_device.h -- POD argument block shared by the host API (letkf_api.hip) and the device code
// (letkf_kernels.hip).  Internal: the public C ABI is include/letkf_amd.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/letkf_amd.h"

namespace letkf {

// LDS scratch of the block Jacobi (letkf_kernels.hip, jacobi_block_mfma), doubles per wave: 512 of the in-register
// 32 x 32 solver + its packed triangular factor (528).  The host sizes the obs tile region of the one-block kernel by it.
constexpr int kBlockJacobiScratch = 512 + 528;

// Plan of the solve kernel's dynamic run scheduling (letkf_wave.hip), worked out on the host for the grid that is
// launched and read by the kernel from its arguments (scalar loads): per XCD range x of the unit ids
//   base[x]  first unit id;  whole[x]  units handed out whole;  f[x]  units handed out in quarters (every t[x]-th of
//   the range, last);  nstat[x]  hand-out positions given to the waves by their position in the grid (no counter);
//   magic[x] = floor(2^40 / (t[x] - 1)) + 1: i / (t - 1) = (i * magic) >> 40 for i < 2^20
// ub = runs per unit, nruns = runs in all.
struct SchedPlan {
  int base[8], whole[8], f[8], t[8], nstat[8];
  unsigned long long magic[8];
  int ub;
  long nruns;
};

struct PointArgs {
  int k, nv;
  int mode;            // 0: CSR gather from the obs table (das_letkf body); 1: dense hdxb batch (letkf_core);
                       // 2: das_letkf body with obs_local fused in (no lists: the kernel walks the sorting mesh itself)
  int ldg, ldy, tn;    // leading dims of G / obs tile, obs rows per LDS tile
  long npts;
  // mode 0 observations
  const long* obs_off;
  const int* obs_idx;
  const double* rdiag_l;
  const double* rloc_l;
  const double* ensval;
  long kld;
  const double* dep;
  // mode 1 observations
  const int* nobsl;
  const double* hdxb;
  const double* rdiag;
  const double* rloc;
  const double* depv;
  const double* depd;
  int nobs;
  int rdiag_wloc;
  // switches
  int det_run, infl_adaptive, relax_to_inflated_prior;
  int iv_p, iv_q_first, iv_q_last;
  int add_wbar_to_trans;
  unsigned var_mask;   // variables of this variable-localisation class (all ones = every variable)
  int max_sweep;       // Jacobi sweep cap (60); lowered only by the LETKF_AMD_MAX_SWEEP profiling knob
  double relax_alpha, relax_alpha_spread, q_update_top, q_sprd_max;
  // state
  const double* beta;
  double* infl;
  long infl_sv;        // stride between the variables of infl / rtps_out (npts, or the whole field's size for a slab call)
  const double* gues;
  double* anal;
  long sp, sm, sv;
  // optional outputs
  double* trans_out;
  double* transm_out;
  double* transmd_out;
  double* pa_out;
  int* status;
  int* nsweep;
  double* rtps_out;    // [npts*nv] RTPS factor diagnostic (work3da), or null
  // large-k workspace
  double* ws;
  long ws_per_block;   // doubles
  int big_block;       // large-k path: block Jacobi on the matrix cores (0: streaming column-pair Jacobi)
  // mode 2: the search tables and the points' coordinates (rig1, rjg1, p mean, hgt1)
  letkf_search_tables stab;
  const double *pri, *prj, *prlev, *prz;
  int* nobs_out;       // [npts] local observation count (modes 2, 3), or null
  // mode 3: the horizontal survivors of each column (letkf_survivors_kernel): entry = 4 doubles (row | ctype << 32 as bits,
  // nd_h, v_obs, err) in surv[], column b of this launch owns entries sv_off[b] .. sv_off[b + 1]; the points of the launch are
  // p = pt0 + a * pt_stride + b, a < npts / warm_stride levels, b < warm_stride columns
  const long* sv_off;
  const double* surv;
  long pt_stride, pt0; // (0, 0: p = a * warm_stride + b as in every other mode)
  // ... and the per-wave slots of sl_cap entries in which a point's local list is assembled (obs_idx / rdiag_l / rloc_l point to
  // the same arrays; slot = the wave's number in the grid)
  int* sl_idx;
  double *sl_rd, *sl_rl;
  long sl_cap;
  // wave kernel: launch shape and warm-start workspace (wave_launch_shape)
  double* warm_ws;
  int run_len, wave_grid, warm_dbg;
  long warm_stride;           // >= 1: a run walks points p, p + warm_stride, ... (letkf_das_args.warm_stride)
  int skip_trivial;           // points with beta == 0 or without observations were done by letkf_trivial_points_kernel: skip them
  unsigned* sched;            // dynamic run scheduling: 8 counters 64 bytes apart, zeroed before the launch (null: static dealing)
  SchedPlan plan;             // ... and its plan, filled in by launch_wave_kernel for the grid it launches
  unsigned long long* prof;   // profiling build (-DLETKF_WAVE_PROF) only: per-phase s_memtime totals, else null
};

// Staged (three-kernel) path, letkf_staged.hip / letkf_eig.hip: per point of a batch one workspace slab
//   G [k * ldg] | V0 [k] | V1 [k] | SC [16] | X [nv k] | TT [nb k] | PC [nb k] | QQ [nb k] | OUT [nb k] | (W [k * ldg])
// (ldg = k | 1, nb = nv + 2; + the residual history H of the eigen-free stage), meta[2 it] = mode | solver << 8 (mode 0: no
// eigenproblem, 1: primal k x k, 2: dual n x n; solver 1: workgroup Jacobi, 2: block Jacobi, 3: eigen-free stage, which
// rewrites it to 1 / 2 for a point it gives up), meta[2 it + 1] = matrix order m, info[2 it] = sweeps (eigen-free:
// -(iterations)), info[2 it + 1] = converged.
struct EigArgs {
  double* ws;
  long ws_per_point;
  long npts;           // points in this batch
  long pt0;            // first point of the batch
  const int* meta;
  int* info;
  int max_sweep;
};
struct StagedArgs {
  PointArgs A;
  long pt0, nbatch;
  int* meta;
  int* info;
  int kkout;           // slab carries W (k x k outputs requested)
  int wg_max_order;    // largest order the workgroup Jacobi takes
  int gram_mfma;       // the matrix-core Gram stage (letkf_gram.hip) runs in front: letkf_stage_gram_kernel takes only what that one leaves
  int poly_max_n;      // eigen-free points (letkf_krylov.hip: conjugate gradients + Lanczos instead of an eigen-decomposition): largest order, 0 = off
};
int stage_apply_pcq_doubles(int k, int nv);
int stage_krylov_max_n(int k);
int stage_krylov_max_iter();
long stage_krylov_hist_doubles(int k);
hipError_t launch_stage_krylov(const StagedArgs& s, size_t lds_max, hipStream_t st);
size_t eig_wg_lds_bytes(int NP, int RP, int RBR, int SB);
int eig_wg_max_order();
hipError_t launch_eig_wg(const EigArgs& e, int mcap, int num_cu, hipStream_t st);
hipError_t launch_eig_block(const EigArgs& e, int kmax, int num_cu, hipStream_t st);
long staged_ws_per_point(int k, int nv, int kkout, long hist);
hipError_t launch_stage_gram(const StagedArgs& s, size_t lds_max, hipStream_t st);
hipError_t launch_stage_gram_mfma(const StagedArgs& s, hipStream_t st);
hipError_t launch_stage_apply(const StagedArgs& s, hipStream_t st);

struct LaunchPlan {
  bool big;
  int rmax;
  int grid, block;
  size_t lds_bytes;
};

struct SearchArgs {
  letkf_search_tables t;
  long npts;
  const double* ri;
  const double* rj;
  const double* rlev;
  const double* rz;
  int fill;
  int* counts;
  const long* obs_off;
  int* obs_idx;
  double* rdiag_l;
  double* rloc_l;
  int limited;         // some max_nobs > 0 (host knowledge: sizes the LDS candidate cache)
};
hipError_t launch_search(const SearchArgs& a, int num_cu, hipStream_t st);
// kref: null = the distance criterion's rings (nd_h^2); else [ngroup] reference offsets of the general ring key (criteria 2, 3)
hipError_t launch_ring_survivors(const letkf_search_tables& t, long col0, long ncol, const double* rig, const double* rjg, int fill,
                                 int* counts, const long* goff, double* sv, int* roff, const double* kref, int num_cu, hipStream_t st);
hipError_t launch_search_rings(const letkf_search_tables& t, long col0, long ncol, long nij1, int nlev, const double* rlev,
                               const double* rz, int fill, int* counts, const long* obs_off, int* obs_idx, double* rdiag_l,
                               double* rloc_l, int* nobs_ctype, double* cutd_ctype, const long* goff, const double* sv,
                               const int* roff, const double* kref, int num_cu, hipStream_t st);
hipError_t launch_ctype_min_err(const letkf_search_tables& t, double* out, hipStream_t st);
int search_rings_max_nobs();
int search_rings_count();
int search_rings_lds_survivors();
hipError_t launch_survivors(const letkf_search_tables& t, long col0, long ncol, const double* rig, const double* rjg, int fill,
                            int* counts, const long* sv_off, double* sv, int num_cu, hipStream_t st);
hipError_t launch_search_columns(const letkf_search_tables& t, long nij1, int nlev, const double* rig,
                                 const double* rjg, const double* rlev, const double* rz, int fill, int* counts,
                                 const long* obs_off, int* obs_idx, double* rdiag_l, double* rloc_l, int* nobs_ctype,
                                 int num_cu, hipStream_t st);

hipError_t launch_search_columns_limited(const letkf_search_tables& t, long nij1, int nlev, const double* rig,
                                         const double* rjg, const double* rlev, const double* rz, int fill, int* counts,
                                         const long* obs_off, int* obs_idx, double* rdiag_l, double* rloc_l,
                                         int* nobs_ctype, double* cutd_ctype, int num_cu, hipStream_t st);
hipError_t launch_point_kernel(const PointArgs& a, const LaunchPlan& p, hipStream_t st);
bool wave_kernel_supports(int k, int nv, int mode);
int wave_kernel_kr(int k);   // rows of the instantiation that serves k
void wave_launch_shape(int k, int mode, long npts, int num_cu, int run_req, long stride, int* run_len, int* grid,
                       size_t* ws_bytes);
hipError_t launch_wave_kernel(const PointArgs& a, int num_cu, hipStream_t st);
// three points per wave for k <= 20 (letkf_trio.hip)
bool trio_kernel_supports(const PointArgs& a);
hipError_t launch_trio_kernel(const PointArgs& a, int num_cu, hipStream_t st);
int trio_points_per_wave(int k);
int sched_plan_check(long npts, long stride, int run_len, int grid, int ppw, int resident_per_xcd, int ub_of);
bool trivial_pass_supports(const PointArgs& a);
hipError_t launch_trivial_points(const PointArgs& a, hipStream_t st);
hipError_t launch_obs_departure(const letkf_qc_params& p, long nobs, const int* elm, const double* dat, const double* err,
                                double* ensval, long kld, double* val, int* qc, int num_cu, hipStream_t st);
hipError_t obs_mesh_sort(const letkf_mesh& m, long nobs, const int* ctype, const double* ri, const double* rj,
                         const int* qc, int* n_cell, int* key, long* nsorted, void* scratch, size_t* scratch_bytes,
                         int num_cu, hipStream_t st);
hipError_t obs_halo_plan(const letkf_halo_layout& l, const int* n_all, int* ac_ext, int* src_row, long cap,
                         long* nobstotal, int num_cu, hipStream_t st);
hipError_t launch_gather_rows(long nrows, const int* src_row, int ncols, const double* src, long ld_src, double* dst,
                              long ld_dst, int num_cu, hipStream_t st);
hipError_t launch_gather_i32(long nrows, const int* src_row, const int* src, int* dst, int num_cu, hipStream_t st);
size_t monit_scratch_bytes(int nid, int num_cu);
hipError_t launch_monit_dep(int nid, const int* elem_uid, long nn, const int* elm, const double* dep, const int* qc,
                            int* nobs, double* bias, double* rmse, void* scratch, int num_cu, hipStream_t st);
hipError_t launch_additive(int k, int nv, long npts, long nij1, double* anal, const double* add, long sp, long sm,
                           long sv, double infl_add, const double* weight, const double* qmean, long q_sp, long q_sv,
                           int iv_q_first, int iv_q_last, const int* ishuf, int num_cu, hipStream_t st);
hipError_t launch_addinfl_weight(long nij1, const double* rig, const double* rjg, long nob, const double* ob_ri,
                                 const double* ob_rj, double dx, double dy, double hori_loc, double cut2, double* w,
                                 int num_cu, hipStream_t st);
int rccl_allgatherv(void* comm, int nranks, int myrank, const int64_t* counts, int64_t row_bytes, const void* send,
                    void* recv, hipStream_t st, const char** what);
int rccl_alltoallv(void* comm, int nranks, int myrank, const int64_t* scount, const int64_t* soff, const int64_t* rcount,
                   const int64_t* roff, int64_t row_bytes, const void* send, void* recv, hipStream_t st, const char** what);
hipError_t launch_block_slot(int dir, long npl, int nv3d, double* blk, double* x, long sp, long mo, long sv, hipStream_t st);
int rccl_allreduce_sum_i32(void* comm, int nranks, int64_t count, int32_t* buf, hipStream_t st, const char** what);
hipError_t launch_relax_beta(const letkf_beta_params& p, long nij1, int nlev, const double* rig, const double* rjg,
                             const double* hgt, double* beta, int num_cu, hipStream_t st);
hipError_t launch_infl_init(long n, double* w, double infl_mul, double infl_mul_min, int num_cu, hipStream_t st);
hipError_t launch_ens_to_pert(int k, int nv, long npts, double* x, long sp, long sm, long sv, hipStream_t st);
hipError_t launch_state_trans(const letkf_state_consts& c, int nlev, long nxy, int nv3d, double* v, int inverse,
                              hipStream_t st);
hipError_t launch_member_points(int dir, int nlev, int nlon, long nxy, int nv3d, int np, int rank, long nij1,
                                double* v3dg, double* x, long sp, long sm_m, long sv, hipStream_t st);
hipError_t launch_ens_spread(int k, int nv, long npts, const double* x, long sp, long sm, long sv, double* sprd,
                             hipStream_t st);
hipError_t launch_ens_mean(int k, int nv, long npts, double* x, long sp, long sm, long sv, hipStream_t st);

}  // namespace letkf
