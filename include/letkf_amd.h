/*
 * letkf_amd.h -- C ABI of the MI355X-native LETKF analysis core (libletkf_amd.so).
 *
 * Drop-in boundary for ONE path of gylien/scale-letkf: the per-grid-point
 * ensemble transform  common/common_letkf.f90::letkf_core  and its driver loop
 * scale/letkf/letkf_tools.f90::das_letkf (file:line below are relative to the
 * reference tree).  The reference has no FFI layer; the call boundary is the
 * Fortran module procedure itself, so every entry point here names the Fortran
 * interface it replaces.  The Fortran-side binding (ISO_C_BINDING module that
 * keeps the reference's letkf_core signature) is scale-letkf_amd/fortran/ and
 * is shown in INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes, no C++/torch types.  All matrices are
 * column-major (Fortran), reals are IEEE double (r_size = r_dble,
 * common/common.f90:18-24), integers 32-bit unless typed otherwise.
 * "dev" pointers are device (HBM) addresses valid on the context's GPU, "host"
 * pointers are ordinary host memory.  Caller owns every buffer; the library
 * keeps no pointer past a call (except the context's own workspace).
 * Every function returns LETKF_OK (0) or a negative LETKF_E_* host-side error;
 * per-problem numerical status codes (>0) are written to the status arrays.
 * There is NO CPU fallback: without a usable gfx950 device every compute entry
 * fails with LETKF_E_NO_DEVICE.
 */
#ifndef LETKF_AMD_H
#define LETKF_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LETKF_AMD_ABI_VERSION 7

/* host-side errors (function return values) */
#define LETKF_OK 0
#define LETKF_E_INVALID (-1)   /* bad argument (NULL, size, unsupported k) */
#define LETKF_E_HIP (-2)       /* a HIP runtime call failed: see letkf_amd_last_error() */
#define LETKF_E_NO_DEVICE (-3) /* no gfx950 device / context not bound to one */
#define LETKF_E_NUMERIC (-4)   /* at least one problem returned a status > 0 (host-pointer entries only) */

/* per-problem status (what the reference turns into WRITE + STOP 2,
 * common/common_mtx.f90:61-64,75-78; SURVEY.md section 8(b) "Errors") */
#define LETKF_ST_OK 0
#define LETKF_ST_NOT_CONVERGED 1 /* eigensolve did not converge (reference: rs ierr /= 0) */
#define LETKF_ST_NONPOSITIVE 2   /* largest eigenvalue <= 0 (reference: "All Eigenvalues are below 0") */
#define LETKF_ST_ILLCOND 3       /* lambda_max/lambda_min > 1/sqrt(eps): the reference would zero modes
                                    (common/common_mtx.f90:66-74) and then divide by zero; here the result
                                    is still computed without truncation and the point is flagged */

typedef struct letkf_ctx letkf_ctx;

int letkf_amd_abi_version(void);
const char *letkf_amd_last_error(void);

/* Context = device + stream + reusable device workspace.  device_id < 0: current device. */
int letkf_ctx_create(int device_id, letkf_ctx **ctx);
int letkf_ctx_destroy(letkf_ctx *ctx);
/* Run on a caller-owned hipStream_t (e.g. torch's current stream).  The handle is used as is: NULL selects HIP's
 * default (null) stream.  A fresh context runs on its own non-blocking stream until this is called. */
int letkf_ctx_set_stream(letkf_ctx *ctx, void *hip_stream);
int letkf_ctx_synchronize(letkf_ctx *ctx);
/* Options of a context.  LETKF_OPT_STAGED_POLY (default 1): a loop-body call (letkf_das_points_dev with lists) with
 * k >= 63, or with nv != 11 at any k, that returns no k x k matrix analyses its grid points WITHOUT an eigen-decomposition:
 * w-bar, the transform applied to the perturbations and the RTPS quadratic forms come from conjugate gradients and the
 * Lanczos tridiagonal of the point's matrix -- Z Z^T + (k-1)/rho I in observation space for a point with fewer local
 * observations than members, Z^T Z + (k-1)/rho I in member space otherwise, any order up to 512 -- applied to the nv + 2
 * right-hand sides on the matrix cores (same result to rounding).  No spectral bound and no degree cap
 * are involved: a point that does not converge within 128 iterations (cond(A) beyond ~1e3 with a flat spectrum, or a
 * matrix that is not positive definite to rounding) goes through the Jacobi eigen stage inside the same call.  nsweep
 * reports -(iterations) for a point analysed this way, the Jacobi's sweep count (> 0) otherwise.
 * 0: every point goes through the Jacobi (what replaces common/common_mtx.f90:41 mtx_eigen) -- 63 <= k <= 100 on the
 * two-wave register kernel, larger k on the staged path's eigen stage -- as calls with trans / Pa outputs always do. */
#define LETKF_OPT_STAGED_POLY 1
/* LETKF_OPT_COLUMN_SURVIVORS: letkf_das_columns_dev's list-free route (see (3c)).  2 (default): where the lists of all levels would
 * not fit `list_bytes` at once; 1: wherever the one-wave kernel serves the call; 0: never (always lists). */
#define LETKF_OPT_COLUMN_SURVIVORS 2
/* LETKF_OPT_LIMITED_RINGS: the column search under MAX_NOBS_PER_GRID (distance criterion) on DENSE observations -- a group's
 * horizontal survivors kept in global memory by rings of nd_h^2, a level takes the nearest rings only (letkf_search.hip).
 * 2 (default): where the survivors overflow the column kernel's buffer on average (weighed with a survivor count and two
 * read-backs; a "not dense" verdict is remembered for the same tables and columns -- by their addresses -- and not weighed again:
 * set 1 if the same buffers turn dense later); 1: wherever eligible (criterion 1, or 2 with one
 * variable-localisation factor per merged group; every limit <= 128; <= 64 combined types); 0: never. */
#define LETKF_OPT_LIMITED_RINGS 3
/* LETKF_OPT_RING_BATCH_MB (default 8192): device workspace of the ring route per batch of columns, in MiB. */
#define LETKF_OPT_RING_BATCH_MB 4
/* LETKF_OPT_RING_RELEASE (default 0): inside letkf_das_columns_dev the ring-ordered survivors of ALL columns are kept for the
 * calls of the entry where they fit half of the free device memory (allocated at their exact size; if that allocation fails the
 * entry falls back to batches of LETKF_OPT_RING_BATCH_MB).  0: the buffer stays with the context for the next analysis;
 * 1: whatever exceeds the batch budget is freed when the entry returns (a host model that needs the memory between analyses
 * pays the allocation again every call: ~1.7 s for 64 GB). */
#define LETKF_OPT_RING_RELEASE 5
/* LETKF_OPT_SMALL_K_TRIO (default 1): loop-body calls with lists, k <= 20, nv = 11 and no per-point matrix outputs run three grid
 * points per wavefront (csrc/letkf_trio.hip: one eigensolve for the three, cold starts); 0: the one-point register kernel with
 * its warm-started runs, as for every other ensemble size.  Same analysis to rounding either way. */
#define LETKF_OPT_SMALL_K_TRIO 6
int letkf_ctx_set_option(letkf_ctx *ctx, int option, int value);

/*---------------------------------------------------------------------------
 * (1) Fine boundary, host pointers, one problem:
 *     SUBROUTINE letkf_core(ne,nobs,nobsl,hdxb,rdiag,rloc,dep,parm_infl,trans,
 *                           transm,pao,rdiag_wloc,infl_update,depd,transmd)
 *     common/common_letkf.f90:52-68.  Absent OPTIONALs are NULL.  hdxb has
 *     leading dimension nobs, only rows 1..nobsl are read (:35-37).  transmd
 *     is computed only when depd AND transmd are given (:188); when transm is
 *     NULL, w-bar is added to every column of trans (:218-226).
 *     *status receives a LETKF_ST_* code (or a negative LETKF_E_*); NULL allowed.
 *     Thread-safe (the reference is called from inside an OpenMP region,
 *     scale/letkf/letkf_tools.f90:289): uses a per-thread context.
 *-------------------------------------------------------------------------*/
void letkf_core_c(int ne, int nobs, int nobsl, const double *hdxb, const double *rdiag,
                  const double *rloc, const double *dep, double *parm_infl, double *trans,
                  double *transm, double *pao, const int *rdiag_wloc, const int *infl_update,
                  const double *depd, double *transmd, int *status);

/*---------------------------------------------------------------------------
 * (1b) Fine boundary, batched, device pointers: nbatch independent letkf_core
 *      problems of identical (ne, nobs) with per-problem nobsl, laid out back
 *      to back (problem b at offset b*ne*nobs, b*nobs, b*ne*ne, b*ne).
 *-------------------------------------------------------------------------*/
typedef struct {
  int32_t ne, nobs;
  int64_t nbatch;
  const int32_t *nobsl;  /* dev [nbatch] */
  const double *hdxb;    /* dev [nbatch][ne][nobs] column-major per problem */
  const double *rdiag;   /* dev [nbatch][nobs] */
  const double *rloc;    /* dev [nbatch][nobs] */
  const double *dep;     /* dev [nbatch][nobs] */
  const double *depd;    /* dev [nbatch][nobs] or NULL */
  double *parm_infl;     /* dev [nbatch] INOUT */
  double *trans;         /* dev [nbatch][ne*ne] */
  double *transm;        /* dev [nbatch][ne] or NULL */
  double *pao;           /* dev [nbatch][ne*ne] or NULL */
  double *transmd;       /* dev [nbatch][ne] or NULL */
  int32_t rdiag_wloc;    /* 0/1 (reference default .false.) */
  int32_t infl_update;   /* 0/1 */
  int32_t *status;       /* dev [nbatch] or NULL */
  int32_t *nsweep;       /* dev [nbatch] or NULL: Jacobi sweeps used (diagnostic) */
} letkf_core_batch_args;
int letkf_core_batch_dev(letkf_ctx *ctx, const letkf_core_batch_args *args);

/*---------------------------------------------------------------------------
 * (2) Coarse boundary: the das_letkf main loop body for a batch of grid points,
 *     scale/letkf/letkf_tools.f90:313-527 (one variable-localisation class per
 *     call, see var_mask; 2-D variables, :530-659, are further variables of a
 *     level-1 call -- INTEGRATION.md "2-D variables").  For every point:
 *     local-obs gather by index -> letkf_core -> RTPP/RTPS relaxation (:457-469)
 *     -> total weight with beta (:472-477) -> analysis of the k members and of
 *     the deterministic member (:480-497) -> q-spread clamp (:500-513).
 *
 *     Observation table = obsda_sort (scale/common/common_obs_scale.f90:112-130):
 *       ensval[iob*kld + m], m = 0..k-1 member perturbations in obs space
 *       (member-fastest, as obsda_sort%ensval(1:MEMBER,iob)); slot m = k holds
 *       the deterministic departure ensval(mmdetobs,iob) when det_run;
 *       dep[iob] = obsda_sort%val(iob).
 *     Local lists = what obs_local returns (:1325): for point p the entries
 *       obs_off[p] .. obs_off[p+1]-1 of obs_idx (0-based rows of the table),
 *       rdiag_l (= err^2/rloc, :1903) and rloc_l.
 *     State: gues holds PERTURBATIONS in members 0..k-1, the ensemble mean in
 *       slot k (mmean) and the deterministic member in slot k+1 (mmdet)
 *       (:209-230, common_mpi_scale.f90:468-507).  Element (point p, member m,
 *       variable v) lives at p*sp + m*sm + v*sv doubles, for gues and anal
 *       alike; the reference's gues3d(nij1,nlev,nens,nv3d) is sp=1,
 *       sm=nij1*nlev, sv=nij1*nlev*nens.  anal slot k (mean) is NOT written
 *       (the reference fills it afterwards with ensmean_grd, letkf.f90:207).
 *     infl: work3d(ij,ilev,n) at p + npts*v; INOUT when infl_adaptive.
 *-------------------------------------------------------------------------*/
typedef struct {
  int32_t k;                 /* MEMBER */
  int32_t nv;                /* nv3d (11 in the reference, common_nml.f90:19) */
  int32_t det_run;           /* DET_RUN */
  int32_t infl_adaptive;     /* INFL_MUL_ADAPTIVE */
  int32_t relax_to_inflated_prior; /* RELAX_TO_INFLATED_PRIOR */
  int32_t iv_p;              /* 0-based index of pressure (iv3d_p) for Q_UPDATE_TOP */
  int32_t iv_q_first, iv_q_last;   /* 0-based inclusive iv3d_q .. iv3d_qg */
  int32_t warm_stride;       /* 0 / 1: a warm-start run (warm_run below) walks CONSECUTIVE points; S > 1: points p, p + S,
                                p + 2 S, ... (npts must be a multiple of S).  With gues3d's point order p = ij + nij1 * lev,
                                S = nij1 makes the runs go up a column -- vertical neighbours, whose local observations
                                are the same rows with slowly changing weights: a better starting point than the
                                horizontal neighbour (C2: off-diagonal norm of Q'AQ 0.013 against 0.023).  Same results
                                to rounding either way.  (reserved0 = 0 in ABI <= 3; ABI 4 libraries already read it as described here, the
                                version number followed in 5.) */
  double relax_alpha;        /* RELAX_ALPHA (RTPP), 0 = off */
  double relax_alpha_spread; /* RELAX_ALPHA_SPREAD (RTPS), 0 = off; RTPP wins when both set (:457) */
  double q_update_top;       /* Q_UPDATE_TOP, <= 0 = off */
  double q_sprd_max;         /* Q_SPRD_MAX, <= 0 = off */
  int64_t npts;
  const int64_t *obs_off;    /* dev [npts+1] */
  const int32_t *obs_idx;    /* dev [obs_off[npts]] */
  const double *rdiag_l;     /* dev, same length */
  const double *rloc_l;      /* dev, same length */
  const double *ensval;      /* dev [nobs_tot][kld] */
  int64_t kld;               /* >= k (+1 when det_run) */
  const double *dep;         /* dev [nobs_tot] */
  const double *beta;        /* dev [npts] or NULL (= 1 everywhere) */
  double *infl;              /* dev [npts*nv] */
  const double *gues;        /* dev */
  double *anal;              /* dev */
  int64_t sp, sm, sv;        /* strides in doubles */
  double *trans_out;         /* dev [npts][k*k] or NULL (diagnostic / parity) */
  double *transm_out;        /* dev [npts][k] or NULL */
  double *pa_out;            /* dev [npts][k*k] or NULL */
  int32_t *status;           /* dev [npts] or NULL */
  int32_t *nsweep;           /* dev [npts] or NULL: Jacobi sweeps; < 0: -(CG iterations) of a point analysed without an eigen stage (LETKF_OPT_STAGED_POLY) */
  double *rtps_infl_out;     /* dev [npts*nv] or NULL: the RTPS factor applied to T per variable, work3da of
                                RELAX_SPREAD_OUT (scale/letkf/letkf_tools.f90:271-276, 460-462, 735-759); 1 where
                                no RTPS factor applies (RTPP / none / beta = 0 / Q_UPDATE_TOP skip) */
  int32_t warm_run;          /* eigensolver warm start: consecutive points are solved in runs, each solve started
                                from the eigenvectors of the point before it (same results to rounding; fewer sweeps
                                when consecutive points are spatial neighbours, as in gues3d's ij-fastest order).
                                0 = library default (runs of up to 16), 1 = off, n > 1 = runs of n points */
  uint32_t var_mask;         /* variables (bit v, v < nv) of THIS variable-localisation class; 0 = all.  With several
                                classes (var_local_n2nc_max > 1, scale/letkf/letkf_tools.f90:130-157) the driver makes
                                one call per class with the class's local lists (its var_local factors enter rloc /
                                rdiag through letkf_search_tables.varloc); variables outside the mask, their infl and
                                rtps_infl_out entries are not touched */
  int64_t infl_sv;           /* stride between the variables of infl and rtps_infl_out: element (p, v) at p + infl_sv*v.
                                0 = npts (the call covers the whole field work3d(nij1*nlev, nv3d)); a call for a slab of
                                levels of a larger field passes infl + p0 and the FIELD's nij1*nlev here (ABI 5) */
} letkf_das_args;
int letkf_das_points_dev(letkf_ctx *ctx, const letkf_das_args *args);

/*---------------------------------------------------------------------------
 * (2b) Streaming passes either side of the loop, on the same strided layout:
 *      the perturbation pass scale/letkf/letkf_tools.f90:209-230 and
 *      ensmean_grd scale/common/common_scale.f90:1513-1552 (called at
 *      scale/letkf/letkf.f90:207 on anal3d).  x: dev, element (p,m,v) at
 *      p*sp + m*sm + v*sv; slot m = k is the mean.
 *-------------------------------------------------------------------------*/
int letkf_ens_to_perturbations_dev(letkf_ctx *ctx, int32_t k, int32_t nv, int64_t npts, double *x,
                                   int64_t sp, int64_t sm, int64_t sv);
int letkf_ens_mean_dev(letkf_ctx *ctx, int32_t k, int32_t nv, int64_t npts, double *x, int64_t sp,
                       int64_t sm, int64_t sv);


/*---------------------------------------------------------------------------
 * (3) Local-observation search on the device: obs_local
 *     scale/letkf/letkf_tools.f90:1325-1759 with obs_local_range (:1765), obs_local_cal (:1793-1906),
 *     ij_obsgrd_ext / obs_choose_ext (scale/letkf/letkf_obs.f90:1209-1285) and the top-N selection that the
 *     reference does with QUICKSELECT_arg (common/common_sort.f90:341).  Produces, for a batch of grid points, the
 *     CSR lists that letkf_das_points_dev consumes, so the lists never exist on the host.
 *
 *     Tables = what set_letkf_obs leaves behind (letkf_obs.f90:35-72): the combined obs types ("ctype"), their
 *     localisation scales, the per-ctype sorting mesh with prefix sums ac_ext, and the obs metadata in
 *     obsda_sort order.  All three selection modes: no limit (:1438-1476; list order = the reference's: ctype,
 *     mesh row j, candidate index), limit by distance (:1479-1660), limit by weight / error (:1663-1729).  With a
 *     limit the selected SET equals the reference's (the N best inside the cut-off; the incremental search of
 *     :1527-1602 is a pure speed heuristic) up to ties; the order inside the list is
 *     implementation-defined in the reference too (unstable quick-select).
 *     Dense observations under a limit (thousands of rows inside the horizontal cut-off): the column entry keeps a group's
 *     horizontal survivors in global memory by rings and takes the nearest rings only (LETKF_OPT_LIMITED_RINGS;
 *     letkf_search.hip) -- same selection up to ties.  Rings of nd_h^2 for the distance criterion (and the weight criterion
 *     where a group has one variable-localisation factor), of nd_h^2 + an offset per entry for the weight criterion with
 *     several factors in a group and for the error criterion (2 ln(err^2 / varloc)): all three criteria are served (ABI 7).
 *     The cut-off constants are the reference's single-precision literals (letkf_obs.f90:27-28).
 *-------------------------------------------------------------------------*/
typedef struct {
  int32_t nctype;
  int32_t ngroup;             /* merged groups (letkf_tools.f90:167-192); a lone ctype is a group of one */
  int32_t criterion;          /* MAX_NOBS_PER_GRID_CRITERION: 1 distance, 2 weight, 3 error */
  int32_t nlon, nlat;         /* interior size of the (sub)domain the mesh was built on */
  int32_t limit_hint;         /* what the HOST knows about max_nobs (a device array): 0 unknown -- the search entries read
                                 it back and synchronise the stream once per call; 1 no combined type has a limit;
                                 2 at least one has.  Set it to keep a pipeline of calls free of host synchronisation (the column
                                 entry still synchronises where it weighs the ring route for dense limited observations:
                                 it reads the limits and the survivor counts back to size its buffers). */
  double dx, dy;              /* DX, DY */
  double i_org, j_org;        /* ri - i_org is (ril - IHALO - 0.5) of ij_obsgrd_ext (letkf_obs.f90:1221) */
  double rain_base;           /* VERT_LOCAL_RAIN_BASE */
  /* per group [ngroup+1] / members (ctype ids, master first) */
  const int32_t *group_start;
  const int32_t *group_member;
  /* per ctype [nctype] */
  const int32_t *vmode;       /* vertical coordinate: 0 |dln p| (obs lev), 1 |dz| (type 22), 2 ps (obs dat), 3 rain base.
                                 -DH08, report type 23 (H08IRB, letkf_tools.f90:1859-1861): mode 0 with ob_lev = obsda_sort%lev
                                 of the row (the sensitive height) instead of obs%lev (which holds the band number there) */
  const double *hori_loc;
  const double *vert_loc;     /* 0 = no vertical localisation */
  const double *varloc;       /* var_local(nvar, uid_obs_varlocal(elm)) of the variable class; < tiny rejects */
  const int32_t *max_nobs;    /* MAX_NOBS_PER_GRID of the report type; <= 0: no limit */
  const int32_t *ngrd_i, *ngrd_j, *ngrdsch_i, *ngrdsch_j, *ngrdext_i, *ngrdext_j;
  const int64_t *ac_off;      /* start of the ctype's table inside ac_ext */
  const int32_t *ac_ext;      /* per ctype (ngrdext_i+1) x ngrdext_j: entry (i, j), i = 0..ngrdext_i, j = 1..ngrdext_j at
                                 ac_off + i + (ngrdext_i+1)*(j-1) = number of table rows before the end of cell (i, j)
                                 (obsgrd%ac_ext, letkf_obs.f90:60): cell (i,j) owns rows ac(i-1,j) .. ac(i,j)-1, 0-based */
  /* obs metadata in obsda_sort order [nobstotal] (obs(set)%{ri,rj,lev,dat,err}(idx)) */
  const double *ob_ri, *ob_rj, *ob_lev, *ob_dat, *ob_err;
} letkf_search_tables;

/* Points: fractional grid indices ri, rj (rig1, rjg1), pressure rlev (gues3d mean of iv3d_p), height rz (hgt1).
 * Two-phase CSR build: call with fill = 0 to get counts[npts]; exclusive-scan them into obs_off[npts+1] (any
 * scan; torch.cumsum in the harness); call again with fill = 1 to write obs_idx / rdiag_l / rloc_l.
 * All pointers are device pointers (the tables struct itself is passed by value from the host).  Synchronises the
 * stream once (max_nobs is read back: tables with a limit get an LDS candidate cache) unless tables->limit_hint is set. */
int letkf_obs_search_dev(letkf_ctx *ctx, const letkf_search_tables *tables, int64_t npts, const double *ri,
                         const double *rj, const double *rlev, const double *rz, int32_t fill, int32_t *counts,
                         const int64_t *obs_off, int32_t *obs_idx, double *rdiag_l, double *rloc_l);


/* (3a) Column-cooperative variant for the reference's own point layout: point p = ij + nij1*lev, rig / rjg [nij1]
 * (rig1, rjg1), rlev / rz [nij1*nlev] (gues3d(:,:,mmean,iv3d_p), hgt1).  One wavefront per horizontal point evaluates
 * the horizontal part of obs_local_cal once per observation and only the vertical part per level.
 * Without a limit: lists identical, entry for entry, to letkf_obs_search_dev on the expanded coordinates.
 * With MAX_NOBS_PER_GRID (all three criteria, merged groups under the master's limit, :1434-1436, :1479-1729): per
 * level the N best of the column's survivors are chosen by the same radix select -- the same selected SET as
 * letkf_obs_search_dev and the reference (up to ties), emitted in candidate order.
 * Same two-phase protocol (fill = 0: counts[nij1*nlev]; fill = 1: write).  Diagnostics, either phase, dev
 * [nij1*nlev][nctype] or NULL -- the inputs of NOBS_OUT (:440-447):
 *   nobs_ctype  nobsl_t of obs_local: accepted rows per combined type; for a limited group the selected count, on
 *               its master (:1633, :1713), 0 on the other members;
 *   cutd_ctype  cutd_t (:1384-1389, :1636-1640, :1716-1727) on the master of every group: criterion 1
 *               hori_loc * dist_zero_fac, or hori_loc * sqrt(largest selected distance) once the limit is hit;
 *               criterion 2 / 3: 0, or the smallest selected weight / largest selected error once it is hit.  (When
 *               exactly N observations lie inside the cut-off the reference reports the one its incremental search
 *               found LAST, :1637; here it is the farthest of the N.)
 * Synchronises the stream once unless tables->limit_hint says what max_nobs holds. */
int letkf_obs_search_columns_dev(letkf_ctx *ctx, const letkf_search_tables *tables, int64_t nij1, int32_t nlev,
                                 const double *rig, const double *rjg, const double *rlev, const double *rz,
                                 int32_t fill, int32_t *counts, const int64_t *obs_off, int32_t *obs_idx,
                                 double *rdiag_l, double *rloc_l, int32_t *nobs_ctype, double *cutd_ctype);

/* (3b) The loop body with obs_local FUSED IN: no local lists at all -- every wavefront walks the sorting mesh of
 * `tables` for its own point (same candidate order as letkf_obs_search_dev, so the results are bit-identical to
 * search + letkf_das_points_dev) and feeds the accepted rows straight into the Gram stage.  Saves the two search
 * passes and the list traffic (C2: 14 GB written and read back per analysis; at C4 sizes the lists do not fit).
 * Restrictions: no MAX_NOBS_PER_GRID limit on any ctype (no-limit mode, letkf_tools.f90:1438-1476), k <= 62 and
 * nv = 11 (the one-wave kernel); otherwise LETKF_E_INVALID -- use (3) + (2).  args->obs_off / obs_idx / rdiag_l /
 * rloc_l are ignored; ri, rj, rlev, rz as in (3); nobs_out (dev [npts] or NULL) receives nobsl of every point.
 * Synchronises the stream once (it checks max_nobs). */
int letkf_das_points_fused_dev(letkf_ctx *ctx, const letkf_das_args *args, const letkf_search_tables *tables,
                               const double *ri, const double *rj, const double *rlev, const double *rz,
                               int32_t *nobs_out);

/* (3c) das_letkf's main loop for a whole (sub)domain in ONE call, the local-observation lists never visible to the host:
 * obs_local for the points p = ij + nij1*lev by the column search (3a), then the loop body (2), slab of levels by slab of
 * levels -- the reference's level loop, scale/letkf/letkf_tools.f90:313 -- so that the lists of a slab fit `list_bytes` of
 * library workspace (0 = 8 GiB; 20 B per list entry; a single level that exceeds it still runs, with a larger workspace).
 * Sequence: one count pass over all levels, a device prefix sum, the level boundaries of the offsets read back (the
 * call's one synchronisation besides the search's own), then per slab the fill pass and the loop body on the context's
 * stream.  args: as for letkf_das_points_dev with npts = nij1*nlev; obs_off / obs_idx / rdiag_l / rloc_l are ignored;
 * trans_out / transm_out / pa_out must be NULL (per-point k x k outputs belong to (2)); warm-start runs go up the columns
 * of a slab.  This is the list-based route: any k, any MAX_NOBS_PER_GRID.
 * The LIST-FREE route (k <= 62, nv = 11, no combined type with a limit; LETKF_OPT_COLUMN_SURVIVORS, by default taken where
 * the lists of all levels would not fit list_bytes at once -- BASELINE configs[3]: 10 M points x ~4900 entries = 1 TB): the
 * horizontal half of obs_local once per COLUMN (the rows inside the horizontal cut-off, 32 B each, in batches of columns that
 * fit list_bytes), the vertical half inside the loop body kernel, which assembles each point's list in a slot of its own wave.
 * No count pass over the levels, nothing per (point, observation) in memory; same observations in the same order with the
 * same weights, hence the same analysis to the last bit (tests/test_gpu_columns.py).  (3b) walks the sorting mesh per POINT
 * instead and is superseded by this route wherever both apply.
 * nobs_out: dev [npts] or NULL, receives nobsl of every point -- 0 where beta = 0 on either route (the reference does not run
 * obs_local there, scale/letkf/letkf_tools.f90:333-359). */
int letkf_das_columns_dev(letkf_ctx *ctx, const letkf_das_args *args, const letkf_search_tables *tables, int64_t nij1,
                          int32_t nlev, const double *rig, const double *rjg, const double *rlev, const double *rz,
                          int64_t list_bytes, int32_t *nobs_out);

/*---------------------------------------------------------------------------
 * (4) The steps either side of the loop, all pure-bandwidth kernels:
 *     state_trans / state_trans_inv   scale/common/common_scale.f90:1181-1224 / :1229-1280
 *       pointwise (DENS,MOMX,MOMY,MOMZ,RHOT,Q*) <-> (U,V,W,T,P,Q*) on a member's subdomain field
 *       v3dg(nlev,nlon,nlat,nv3d) (level-fastest).  The SCALE-RM constants are not in the reference tree
 *       (scale_const / scale_tracer): the caller passes them.  Variable order as common_scale.f90:36-51:
 *       0 rho/u, 1 rhou... : see letkf_state_consts.  state_trans_inv applies the positive-definite clamps of
 *       :1243-1250 first when requested.
 *     member field <-> point-major ensemble   grd_to_buf / buf_to_grd + the ALLTOALL of read_ens_mpi / write_ens_mpi
 *       (scale/common/common_mpi_scale.f90:1099-1274, :1428-1480): the horizontal points of a subdomain are dealt
 *       cyclically over np ranks (local point i of rank r is subdomain point r + np*i, ilon = mod(j,nlon)); this
 *       entry moves ONE member's field into / out of slot m of gues3d(nij1,nlev,nens,nv3d) for one rank's share --
 *       on one node the all-to-all is np of these per member (or device-to-device copies between the GPUs).
 *     enssprd_grd                      scale/common/common_scale.f90:1570-1607
 *-------------------------------------------------------------------------*/
typedef struct {
  double rdry, rvap, cvdry, pre00;   /* CONST_Rdry, CONST_Rvap, CONST_CVdry, CONST_PRE00 */
  double tracer_cv[8];               /* TRACER_CV(1..nv3d-iv3d_q+1): specific heats of the moisture species */
  int32_t iv_rho, iv_rhou, iv_rhov, iv_rhow, iv_rhot;   /* 0-based slots of the prognostic variables ...        */
  int32_t iv_u, iv_v, iv_w, iv_t, iv_p;                 /* ... and of the analysis variables that replace them */
  int32_t iv_q;                      /* first moisture variable; moisture = iv_q .. nv3d-1 */
  int32_t positive_definite_q, positive_definite_qhyd;  /* POSITIVE_DEFINITE_Q / _QHYD (inverse only) */
  int32_t reserved0;
} letkf_state_consts;

int letkf_state_trans_dev(letkf_ctx *ctx, const letkf_state_consts *c, int32_t nlev, int32_t nlon, int32_t nlat,
                          int32_t nv3d, double *v3dg, int32_t inverse);

/* dir = 0: v3dg (member field, level-fastest) -> x slot m;  dir = 1: x slot m -> v3dg.
 * x element (point i, level k, member m, variable n) at (i + nij1*k)*sp + m*sm + n*sv (the strides of section 2). */
int letkf_member_points_dev(letkf_ctx *ctx, int32_t dir, int32_t nlev, int32_t nlon, int32_t nlat, int32_t nv3d,
                            int32_t np, int32_t rank, int32_t m, double *v3dg, double *x, int64_t nij1, int64_t sp,
                            int64_t sm, int64_t sv);

/* sprd[p + npts*v] = sqrt( sum_m (x_m - mean)^2 / (k-1) ), mean taken from slot k. */
int letkf_ens_spread_dev(letkf_ctx *ctx, int32_t k, int32_t nv, int64_t npts, const double *x, int64_t sp,
                         int64_t sm, int64_t sv, double *sprd);

/*---------------------------------------------------------------------------
 * (5) set_letkf_obs on the device (scale/letkf/letkf_obs.f90): the producer of the
 *     observation table the search (3) and the loop body (2) read.  Per rank (= GPU, = SCALE subdomain):
 *       departure + QC          letkf_obs.f90:361-561 (letkf_qc_params.h08 = 0: the reference built without -DH08;
 *                               1: its -DH08 build, Himawari-8 IR rows take the branches of :432-469, :480-487, :520-541)
 *       bucket sort on the mesh :762-822 (+ ij_obsgrd :1186-1203)
 *       [all-gather of the sorted buffers and cell counts over the ranks: RCCL, outside this library]
 *       extended-subdomain plan :922-976 and copy into obsda_sort :1036-1100
 *     Integer results (qc, counts, keys, ac_ext, row map) are identical to the reference's; the departures use the
 *     same sequential member sum and are bit-identical too.
 *-------------------------------------------------------------------------*/
typedef struct {
  int32_t member;                 /* MEMBER */
  int32_t det_run;                /* DET_RUN: column `member` of ensval holds H(x_det), becomes y - H(x_det) (:492-494) */
  int32_t use_radar_ref, use_radar_vr;                           /* USE_RADAR_REF, USE_RADAR_VR */
  int32_t min_radar_ref_member, min_radar_ref_member_obsref;     /* MIN_RADAR_REF_MEMBER(_OBSREF) */
  double radar_ref_thres_dbz;                                    /* RADAR_REF_THRES_DBZ */
  double gross_error, gross_error_rain, gross_error_radar_ref, gross_error_radar_vr, gross_error_radar_prh,
      gross_error_tcx, gross_error_tcy, gross_error_tcp;        /* GROSS_ERROR* */
  /* ---- the reference's -DH08 build (ABI 6).  h08 = 0: everything below is ignored and rows with elm = id_H08IR_obs (8800) are
   * ordinary rows (GROSS_ERROR), exactly what the default build does. */
  int32_t h08;                    /* 1: the -DH08 branches are compiled in */
  int32_t h08_min_cld_member;     /* H08_MIN_CLD_MEMBER: fewer cloudy members = clear sky (gross-error bound 1.0 err, :528-531) */
  double h08_limit_lev;           /* H08_LIMIT_LEV (Pa): rows whose sensitive height obsda%lev lies above it are iqc_obs_bad (:441) */
  double gross_error_h08;         /* GROSS_ERROR_H08 (cloudy sky; the namelist's "< 0: GROSS_ERROR" is the host's to resolve) */
  double h08_bt_min;              /* H08_BT_MIN: observed brightness temperatures below it are iqc_gross_err (:538-540) */
  const double *h08_lev;          /* dev [nobs]: obsda%lev -- read for elm = 8800 rows only */
  double *h08_val2;               /* dev [nobs] INOUT or NULL: obsda%val2, clear-sky BT in, CA = (|mean - clr| + |obs - clr|) / 2 out
                                     (:480-487; the reference computes it for EVERY live row of the -DH08 build, so does this) */
} letkf_qc_params;

/* elm[n] = obs(set)%elm(idx), dat / err likewise (gathered per local H(x) row by the caller); ensval[n*kld + m]
 * INOUT: H(x_m) -> H(x_m) - mean; val[n] OUT: y - mean; qc[n] INOUT (rows with qc > 0 are skipped).
 * h08 = 1, elm = 8800: cloudy members arrive as NEGATIVE brightness temperatures (the obs operator's flag); they are counted
 * and their sign restored before the mean is taken (:448-454). */
int letkf_obs_departure_dev(letkf_ctx *ctx, const letkf_qc_params *p, int64_t nobs, const int32_t *elm,
                            const double *dat, const double *err, double *ensval, int64_t kld, double *val,
                            int32_t *qc);

typedef struct {
  int32_t nctype;
  int32_t nlon, nlat;             /* subdomain size (without halo) */
  int32_t ihalo, jhalo;           /* IHALO, JHALO: ri, rj count the halo */
  int32_t rank_i, rank_j;         /* PRC_2Drank of this rank */
  int32_t fix_ij_obsgrd;          /* 0: ij_obsgrd as the reference has it -- rj scaled with ngrd_i (letkf_obs.f90:1200), so that
                                     on subdomains with ngrd_i /= ngrd_j the sort disagrees with the lookup ij_obsgrd_ext
                                     (:1223, ngrd_j) and grid points lose observations; 1: scale rj with ngrd_j (sort and
                                     lookup agree on any subdomain shape).  Identical on square subdomains. */
  const int32_t *ngrd_i, *ngrd_j; /* HOST [nctype]: obsgrd(ic)%ngrd_i / ngrd_j (letkf_obs.f90:668-669) */
} letkf_mesh;

/* ctype[n]: 0-based combined type of row n (ctype_elmtyp(uid_obs(elm), typ) - 1); ri, rj: global grid coordinates.
 * n_cell (dev, OUT): per ctype [ngrd_j][ngrd_i] counts of accepted rows, ctypes concatenated (obsgrd%n(:,:,myrank));
 * key (dev, OUT, [nobs]): key[0 .. *nsorted) = 0-based row numbers in (ctype, j, i, row) order (obsda%key).
 * Synchronises the stream (the count goes back to the host). */
int letkf_obs_mesh_sort_dev(letkf_ctx *ctx, const letkf_mesh *mesh, int64_t nobs, const int32_t *ctype,
                            const double *ri, const double *rj, const int32_t *qc, int32_t *n_cell, int32_t *key,
                            int64_t *nsorted);

typedef struct {
  int32_t nctype, nprocs, prc_num_x, myrank;      /* rank r sits at (mod(r, prc_num_x), r / prc_num_x) */
  const int32_t *ngrd_i, *ngrd_j, *ngrdsch_i, *ngrdsch_j;   /* HOST [nctype] */
} letkf_halo_layout;

/* n_all (dev): every rank's n_cell, rank-major (the MPI_ALLREDUCE of obsgrd%n, :826-831).  ac_ext (dev, OUT): per
 * ctype [ngrdext_j][ngrdext_i + 1], concatenated and cumulative over ctypes -- exactly letkf_search_tables.ac_ext.
 * src_row (dev, OUT, capacity cap): obsda_sort row r is row src_row[r] of the rank-major concatenation of all ranks'
 * sorted buffers (the ALLGATHERV receive buffer obsbufr, :1013-1025).  *nobstotal: rows of obsda_sort.
 * Synchronises the stream. */
int letkf_obs_halo_plan_dev(letkf_ctx *ctx, const letkf_halo_layout *layout, const int32_t *n_all, int32_t *ac_ext,
                            int32_t *src_row, int64_t cap, int64_t *nobstotal);

/* dst[r*ld_dst + c] = src[src_row[r]*ld_src + c], c < ncols (ensval rows, val, metadata); and the int32 fields. */
int letkf_obs_gather_rows_dev(letkf_ctx *ctx, int64_t nrows, const int32_t *src_row, int32_t ncols, const double *src,
                              int64_t ld_src, double *dst, int64_t ld_dst);
int letkf_obs_gather_i32_dev(letkf_ctx *ctx, int64_t nrows, const int32_t *src_row, const int32_t *src, int32_t *dst);

/*---------------------------------------------------------------------------
 * (6) After the loop
 *-------------------------------------------------------------------------*/
/* monit_dep, scale/common/common_obs_scale.f90:1851-1895: per observation element (elem_uid, HOST [nid], the ids of
 * common_obs_scale.f90:74-77 in that order) the count, mean and rms of dep over rows with qc == 0; Tv is counted as
 * T and RE0 as REF (:1871-1876); elements without rows get undef (-9.99e33).  elm[n] = NINT(obs%elm).  Outputs are
 * device arrays [nid].  Sums are taken in a fixed two-level order (reproducible run to run; differs from the
 * reference's sequential order in the last bits, below the ES12.3 the reference prints). */
int letkf_monit_dep_dev(letkf_ctx *ctx, int32_t nid, const int32_t *elem_uid, int64_t nn, const int32_t *elm,
                        const double *dep, const int32_t *qc, int32_t *nobs, double *bias, double *rmse);

/* Additive inflation, scale/letkf/letkf_tools.f90:884-913, on the strided ensemble layout of section 2:
 *   anal(p,m,v) += add(p,mshuf(m),v) * infl_add * weight(ij) [* qmean(p,v) for iv_q_first <= v <= iv_q_last]
 * add: the additive ensemble as PERTURBATIONS (read_ens_mpi_addiinfl + ensmean_grd + the subtraction of :862-871 =
 * letkf_ens_mean_dev + letkf_ens_to_perturbations_dev), same strides as anal; weight: [nij1] addinfl_weight or NULL
 * (= 1); qmean: INFL_ADD_Q_RATIO field, element (p,v) at p*q_sp + v*q_sv (pass gues + k*sm with q_sp = sp,
 * q_sv = sv for gues3d(:,:,mmean,:)), or NULL; ishuf: dev [k] 0-based INFL_ADD_SHUFFLE permutation or NULL. */
int letkf_additive_inflation_dev(letkf_ctx *ctx, int32_t k, int32_t nv, int64_t npts, int64_t nij1, double *anal,
                                 const double *add, int64_t sp, int64_t sm, int64_t sv, double infl_add,
                                 const double *weight, const double *qmean, int64_t q_sp, int64_t q_sv,
                                 int32_t iv_q_first, int32_t iv_q_last, const int32_t *ishuf);

/* addinfl_weight of INFL_ADD_REF_ONLY (:813-838): w(ij) = exp(-d2/2) with d2 = min over the reflectivity rows
 * [0, nob) of ((rig-ri)DX)^2 + ((rjg-rj)DY)^2, over hori_loc^2, and 0 beyond dist_zero_fac_square. */
int letkf_addinfl_weight_dev(letkf_ctx *ctx, int64_t nij1, const double *rig, const double *rjg, int64_t nob,
                             const double *ob_ri, const double *ob_rj, double dx, double dy, double hori_loc,
                             double *weight);

/*---------------------------------------------------------------------------
 * (7) The set-up das_letkf does before its loop,
 *     scale/letkf/letkf_tools.f90:130-267 and relax_beta :1911-1948.
 *     The three table derivations are HOST functions (tens of integers, once per analysis; they need no device and
 *     return LETKF_OK / LETKF_E_INVALID); relax_beta and the inflation field are device passes over the points.
 *-------------------------------------------------------------------------*/
/* Variable-localisation classes, :130-157.  var_local: column-major var_local(nvar, nlt) as the reference fills it
 * from VAR_LOCAL_UV .. VAR_LOCAL_H08 (:130-138; nlt = 9).  Variables whose rows agree within tiny() share a class:
 * n2nc[n] = 0-based class of variable n (var_local_n2nc - 1), n2n[n] = 0-based first variable of that class
 * (var_local_n2n - 1), *nclass = var_local_n2nc_max.  One letkf_das_points_dev call per class with var_mask = the
 * class's variables and letkf_search_tables.varloc[ic] = var_local(n2n, uid_obs_varlocal(elm of ctype ic)). */
int letkf_var_local_classes(int32_t nvar, int32_t nlt, const double *var_local, int32_t *n2nc, int32_t *n2n,
                            int32_t *nclass);

/* Merge groups of the obs-number limit, :167-192.  elm_u_ctype / typ_ctype [nctype]: 1-based uid_obs and report type
 * of every combined type (letkf_obs.f90:35-41); ctype_merge: column-major ctype_merge(nid_obs, nobtype), > 0 = merge
 * class (the reference sets (uid_obs(id_radar_ref_obs), 22) = (uid_obs(id_radar_ref_zero_obs), 22) = 1).  Output in the
 * form letkf_search_tables takes: group g owns group_member[group_start[g] .. group_start[g+1]), 0-based ctypes, the
 * master (whose MAX_NOBS_PER_GRID and search radius apply, :1434-1436) first; groups in the order of their masters. */
int letkf_ctype_merge_groups(int32_t nctype, const int32_t *elm_u_ctype, const int32_t *typ_ctype, int32_t nid_obs,
                             int32_t nobtype, const int32_t *ctype_merge, int32_t *group_start, int32_t *group_member,
                             int32_t *ngroup);

/* radar_only, :197-203: 1 when every combined type is of report type typ_radar (22 = 'PHARAD',
 * common_obs_scale.f90:87-92), else 0 (also for nctype = 0 the reference leaves .true.: returned as 1). */
int letkf_radar_only(int32_t nctype, const int32_t *typ_ctype, int32_t typ_radar);

typedef struct {
  int32_t radar_only;            /* letkf_radar_only() */
  int32_t ihalo, jhalo;          /* IHALO, JHALO */
  int32_t nlong, nlatg;          /* global interior size (common_scale.f90:117-121) */
  int32_t reserved0;
  double radar_zmax;             /* RADAR_ZMAX */
  double vert_local_radar;       /* max(VERT_LOCAL(22), VERT_LOCAL_RADAR_VR) */
  double boundary_buffer_width;  /* BOUNDARY_BUFFER_WIDTH, <= 0 = off */
  double dx, dy;                 /* DX, DY */
} letkf_beta_params;

/* relax_beta, :1911-1948, for the points p = ij + nij1*lev: beta[p] from rig[ij], rjg[ij] (rig1, rjg1) and
 * hgt[p] (hgt1(ij, ilev)); the argument `beta` of letkf_das_points_dev.  All device pointers. */
int letkf_relax_beta_dev(letkf_ctx *ctx, const letkf_beta_params *p, int64_t nij1, int32_t nlev, const double *rig,
                         const double *rjg, const double *hgt, double *beta);

/* Multiplicative-inflation field, :237-267: infl_mul > 0 -> work3d = INFL_MUL, else work3d keeps the field the caller
 * read in (INFL_MUL_IN_BASENAME); then work3d = max(work3d, INFL_MUL_MIN) when infl_mul_min > 0.  n = nij1*nlev*nv3d. */
int letkf_infl_init_dev(letkf_ctx *ctx, int64_t n, double *work3d, double infl_mul, double infl_mul_min);

/*---------------------------------------------------------------------------
 * (8) The path's one exchange: MPI_ALLGATHERV of the sorted observation buffers (and, with the same call, of the
 *     mesh-cell counts) over the subdomain ranks, scale/letkf/letkf_obs.f90:1036-1046 / :826-831, on an RCCL
 *     communicator owned by the host (nccl_comm = the host's ncclComm_t, created with ncclCommInitRank; one rank per
 *     GPU).  Grouped ncclSend / ncclRecv with the TRUE row counts on the context's stream: rank r's counts[r] rows of
 *     row_bytes bytes land in recv behind the rows of ranks 0 .. r-1 (rank-major = the receive buffer obsbufr the
 *     extended-subdomain plan of section 5 indexes).  counts: HOST [nranks] (every rank knows them from the
 *     all-reduced cell counts, as in the reference).  send: dev [counts[myrank] * row_bytes], recv: dev
 *     [sum(counts) * row_bytes].  Asynchronous on the stream; LETKF_E_INVALID when RCCL is not loadable.
 *-------------------------------------------------------------------------*/
int letkf_obs_allgatherv_dev(letkf_ctx *ctx, void *nccl_comm, int32_t nranks, int32_t myrank, const int64_t *counts,
                             int64_t row_bytes, const void *send, void *recv);

/* (8b, ABI 7) The other exchanges of the path on the same host-owned communicator, so that a Fortran host reaches every one of
 * them through this library (round 3 had them as torch code only):
 *   letkf_alltoallv_dev       pairwise exchange with TRUE counts (MPI_ALLTOALLV): send_counts[q] rows from row send_offs[q] of
 *                             `send` go to rank q, recv_counts[q] rows from rank q land at row recv_offs[q] of `recv`; ONE group of
 *                             ncclSend / ncclRecv on the context's stream, the own block a device copy (send_counts[myrank] must
 *                             equal recv_counts[myrank]).  Counts and offsets: HOST [nranks], in rows of row_bytes bytes.  This is
 *                             the HALO-ONLY exchange of the observation table -- every rank sends rank q just the rows that fall
 *                             into q's extended subdomain (scale/letkf/letkf_obs.f90:922-976, 1059-1109) instead of the
 *                             ALLGATHERV of everything (:1036-1046) -- and the transport of the member <-> point transpose below.
 *                             On a fully connected xGMI node the pairs use different links at the same time.
 *   letkf_allreduce_sum_i32_dev  MPI_ALLREDUCE(MPI_SUM) in place on dev int32 [count]: the per-mesh-cell observation counts
 *                             of :826-833, from which every rank derives every rank's row counts (ncclAllReduce).
 *   letkf_members_alltoall_dev  scatter_grd_mpi_alltoall (dir 0) / gather_grd_mpi_alltoall (dir 1),
 *                             scale/common/common_mpi_scale.f90:1279-1396: ranks 0 .. mcount-1 each hold ONE member's field
 *                             v3dg(nlev,nlon,nlat,nv3d) (member mstart + rank; NULL elsewhere); every rank owns the points
 *                             ij = myrank, myrank + nranks, ... (grd_to_buf, :1428-1455) and keeps, for them, the members in the
 *                             slots mstart .. mstart + mcount - 1 of x (element (i, lev, m, v) at (i + nij1*lev)*sp + m*sm + v*sv).
 *                             dir 0 deals the fields to the slots, dir 1 assembles the fields from them; blocks of true size
 *                             (the reference pads to nij1max), packed / unpacked by the kernel of letkf_member_points_dev.
 * nranks = 1 needs no communicator (NULL): everything is the own block.  LETKF_E_INVALID when RCCL is not loadable. */
int letkf_alltoallv_dev(letkf_ctx *ctx, void *nccl_comm, int32_t nranks, int32_t myrank, const int64_t *send_counts,
                        const int64_t *send_offs, const int64_t *recv_counts, const int64_t *recv_offs, int64_t row_bytes,
                        const void *send, void *recv);
int letkf_allreduce_sum_i32_dev(letkf_ctx *ctx, void *nccl_comm, int32_t nranks, int64_t count, int32_t *buf);
int letkf_members_alltoall_dev(letkf_ctx *ctx, void *nccl_comm, int32_t nranks, int32_t myrank, int32_t dir, int32_t nlev,
                               int32_t nlon, int32_t nlat, int32_t nv3d, int32_t mstart, int32_t mcount, double *v3dg, double *x,
                               int64_t sp, int64_t sm, int64_t sv);

/* Name(s) of the kernel(s) the context's last letkf_das_points*_dev / letkf_core_batch_dev call went through, as a
 * NUL-terminated string (truncated to len): what bench.py reports as roofline.kernel. */
int letkf_ctx_last_path(letkf_ctx *ctx, char *buf, int32_t len);
/* Kernel timing helper for bench.py: average duration (ms) of the last
 * letkf_das_points_dev / letkf_core_batch_dev launches measured with HIP events on the
 * context's stream since the previous reset; *nlaunch receives the count. */
int letkf_ctx_timing_enable(letkf_ctx *ctx, int enable);
int letkf_ctx_timing_read(letkf_ctx *ctx, double *avg_ms, int64_t *nlaunch, int reset);

/* Self-check of the solve kernel's run scheduling (host only, no device needed): builds the plan that a launch over
 * npts points with warm-start stride `stride`, runs of run_len points and a grid of `grid` workgroups (ppw wave-slots
 * each, resident_per_xcd of them in flight per XCD) would use, walks every hand-out position of every range through the
 * same code the kernel runs, and returns 0 iff every run is handed out exactly once -- whole, or as its four quarters.
 * For the CPU test suite (tests/test_sched_plan.py). */
int letkf_sched_plan_check(int64_t npts, int64_t stride, int32_t run_len, int32_t grid, int32_t ppw, int32_t resident_per_xcd);
/* ... with units that are whole multiples of ub_of runs (csrc/letkf_trio.hip walks three runs in step: ub_of = 3). */
int letkf_sched_plan_check_units(int64_t npts, int64_t stride, int32_t run_len, int32_t grid, int32_t ppw, int32_t resident_per_xcd, int32_t ub_of);

#ifdef __cplusplus
}
#endif
#endif /* LETKF_AMD_H */
