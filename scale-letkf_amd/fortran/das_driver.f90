!===============================================================================
! das_driver.f90 -- the batched loop body called from Fortran (INTEGRATION.md level 2): what a restructured das_letkf
! does for one level slab.  Reads a case written by tests/test_fortran_shim.py, uploads it, runs the perturbation pass
! and ONE letkf_das_points_dev call for all points, downloads the analysis and writes it back.
!   file layout (little endian, stream):
!     int32 k, nv, npts, nobs, kld, det_run, relax(0 none / 1 rtpp / 2 rtps) ; real64 alpha
!     int64 obs_off(npts+1) ; int32 obs_idx(nnz) ; real64 rdiag(nnz), rloc(nnz)
!     real64 ensval(kld,nobs) [member-fastest], dep(nobs), beta(npts), infl(npts*nv)
!     real64 gues(npts,nens,nv)  FULL members in slots 1..k (the driver takes the mean and the perturbations)
!===============================================================================
PROGRAM das_driver
  USE letkf_amd_api
  IMPLICIT NONE
  INTEGER(c_int32_t) :: k, nv, npts, nobs, kld, det_run, relax
  REAL(c_double) :: alpha
  INTEGER(c_int64_t), ALLOCATABLE, TARGET :: obs_off(:)
  INTEGER(c_int32_t), ALLOCATABLE, TARGET :: obs_idx(:), status(:)
  REAL(c_double), ALLOCATABLE, TARGET :: rdiag(:), rloc(:), ensval(:, :), dep(:), beta(:), infl(:), gues(:, :, :), &
                                         anal(:, :, :)
  INTEGER :: u, ios, nens
  INTEGER(c_int64_t) :: nnz
  INTEGER(c_int) :: rc
  TYPE(c_ptr) :: ctx, d_off, d_idx, d_rd, d_rl, d_ens, d_dep, d_beta, d_infl, d_gues, d_anal, d_st
  TYPE(letkf_das_args) :: a
  CHARACTER(len=512) :: fin, fout

  CALL get_command_argument(1, fin)
  CALL get_command_argument(2, fout)
  OPEN (newunit=u, file=trim(fin), access='stream', form='unformatted', status='old', iostat=ios)
  IF (ios /= 0) STOP 3
  READ (u) k, nv, npts, nobs, kld, det_run, relax
  READ (u) alpha
  nens = k + 1 + det_run
  ALLOCATE (obs_off(npts + 1))
  READ (u) obs_off
  nnz = obs_off(npts + 1)
  ALLOCATE (obs_idx(max(nnz, 1_c_int64_t)), rdiag(max(nnz, 1_c_int64_t)), rloc(max(nnz, 1_c_int64_t)))
  IF (nnz > 0) READ (u) obs_idx(1:nnz), rdiag(1:nnz), rloc(1:nnz)
  ALLOCATE (ensval(kld, nobs), dep(nobs), beta(npts), infl(npts*nv), gues(npts, nens, nv), anal(npts, nens, nv), &
            status(npts))
  READ (u) ensval, dep, beta, infl, gues
  CLOSE (u)

  IF (letkf_amd_abi_version() < 3) STOP 4
  CALL chk(letkf_ctx_create(0_c_int, ctx), 'ctx_create')
  d_off = up(c_loc(obs_off), 8_c_size_t*(npts + 1))
  d_idx = up(c_loc(obs_idx), 4_c_size_t*max(nnz, 1_c_int64_t))
  d_rd = up(c_loc(rdiag), 8_c_size_t*max(nnz, 1_c_int64_t))
  d_rl = up(c_loc(rloc), 8_c_size_t*max(nnz, 1_c_int64_t))
  d_ens = up(c_loc(ensval), 8_c_size_t*kld*nobs)
  d_dep = up(c_loc(dep), 8_c_size_t*nobs)
  d_beta = up(c_loc(beta), 8_c_size_t*npts)
  d_infl = up(c_loc(infl), 8_c_size_t*npts*nv)
  d_gues = up(c_loc(gues), 8_c_size_t*npts*nens*nv)
  anal = 0.0d0
  d_anal = up(c_loc(anal), 8_c_size_t*npts*nens*nv)
  status = -1
  d_st = up(c_loc(status), 4_c_size_t*npts)

  ! ensmean_grd + the perturbation pass (letkf_tools.f90:209-230) on the device: gues3d(nij1*nlev, nens, nv3d)
  CALL chk(letkf_ens_mean_dev(ctx, k, nv, INT(npts, c_int64_t), d_gues, 1_c_int64_t, INT(npts, c_int64_t), &
                              INT(npts, c_int64_t)*nens), 'ens_mean')
  CALL chk(letkf_ens_to_perturbations_dev(ctx, k, nv, INT(npts, c_int64_t), d_gues, 1_c_int64_t, &
                                          INT(npts, c_int64_t), INT(npts, c_int64_t)*nens), 'to_perturbations')

  a%k = k; a%nv = nv; a%det_run = det_run; a%infl_adaptive = 0; a%relax_to_inflated_prior = 0
  a%iv_p = 4; a%iv_q_first = 5; a%iv_q_last = MIN(10, nv - 1); a%warm_stride = 0
  a%relax_alpha = 0.0d0; a%relax_alpha_spread = 0.0d0
  IF (relax == 1) a%relax_alpha = alpha
  IF (relax == 2) a%relax_alpha_spread = alpha
  a%q_update_top = 0.0d0; a%q_sprd_max = 0.0d0
  a%npts = npts
  a%obs_off = d_off; a%obs_idx = d_idx; a%rdiag_l = d_rd; a%rloc_l = d_rl; a%ensval = d_ens; a%kld = kld
  a%dep = d_dep; a%beta = d_beta; a%infl = d_infl; a%gues = d_gues; a%anal = d_anal
  a%sp = 1; a%sm = npts; a%sv = INT(npts, c_int64_t)*nens
  a%trans_out = c_null_ptr; a%transm_out = c_null_ptr; a%pa_out = c_null_ptr
  a%status = d_st; a%nsweep = c_null_ptr; a%rtps_infl_out = c_null_ptr
  a%warm_run = 0; a%var_mask = 0; a%infl_sv = 0
  CALL chk(letkf_das_points_dev(ctx, a), 'das_points')
  CALL chk(letkf_ctx_synchronize(ctx), 'synchronize')

  CALL chk(hipMemcpy(c_loc(anal), d_anal, 8_c_size_t*npts*nens*nv, hipMemcpyDeviceToHost), 'download anal')
  CALL chk(hipMemcpy(c_loc(status), d_st, 4_c_size_t*npts, hipMemcpyDeviceToHost), 'download status')
  IF (ANY(status /= 0)) THEN        ! the reference's behaviour: print and STOP 2 (common_mtx.f90:61-64)
    WRITE (6, *) 'letkf_das_points_dev: non-zero status at', COUNT(status /= 0), 'points:', PACK(status, status /= 0)
    STOP 2
  END IF
  OPEN (newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
  WRITE (u) anal
  CLOSE (u)
  rc = letkf_ctx_destroy(ctx)

CONTAINS

  FUNCTION up(host, nbytes) RESULT(d)
    TYPE(c_ptr), INTENT(IN) :: host
    INTEGER(c_size_t), INTENT(IN) :: nbytes
    TYPE(c_ptr) :: d
    CALL chk(hipMalloc(d, nbytes), 'hipMalloc')
    CALL chk(hipMemcpy(d, host, nbytes, hipMemcpyHostToDevice), 'hipMemcpy H2D')
  END FUNCTION up

  SUBROUTINE chk(rc_, what)
    INTEGER(c_int), INTENT(IN) :: rc_
    CHARACTER(*), INTENT(IN) :: what
    IF (rc_ /= 0) THEN
      WRITE (6, *) 'error', rc_, 'in ', what
      STOP 5
    END IF
  END SUBROUTINE chk

END PROGRAM das_driver
