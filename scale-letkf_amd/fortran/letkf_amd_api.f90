!===============================================================================
! letkf_amd_api.f90 -- ISO_C_BINDING view of the batched entry points of include/letkf_amd.h, for a das_letkf that is
! restructured into "search, then one call per level slab" (INTEGRATION.md, level 2).  TYPE letkf_das_args mirrors the
! C struct field by field (same order, same kinds); device buffers are C pointers obtained from hipMalloc (bound
! below straight from libamdhip64 -- hipfort is not needed).
!===============================================================================
MODULE letkf_amd_api
  USE, INTRINSIC :: iso_c_binding
  IMPLICIT NONE
  PUBLIC

  INTEGER(c_int), PARAMETER :: hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2

  TYPE, BIND(C) :: letkf_das_args
    INTEGER(c_int32_t) :: k, nv, det_run, infl_adaptive, relax_to_inflated_prior, iv_p, iv_q_first, iv_q_last, reserved0
    REAL(c_double)     :: relax_alpha, relax_alpha_spread, q_update_top, q_sprd_max
    INTEGER(c_int64_t) :: npts
    TYPE(c_ptr)        :: obs_off, obs_idx, rdiag_l, rloc_l, ensval
    INTEGER(c_int64_t) :: kld
    TYPE(c_ptr)        :: dep, beta, infl, gues, anal
    INTEGER(c_int64_t) :: sp, sm, sv
    TYPE(c_ptr)        :: trans_out, transm_out, pa_out, status, nsweep, rtps_infl_out
    INTEGER(c_int32_t) :: warm_run
    INTEGER(c_int32_t) :: var_mask
  END TYPE letkf_das_args

  INTERFACE
    FUNCTION letkf_amd_abi_version() BIND(C, name='letkf_amd_abi_version') RESULT(v)
      IMPORT :: c_int
      INTEGER(c_int) :: v
    END FUNCTION
    FUNCTION letkf_ctx_create(device_id, ctx) BIND(C, name='letkf_ctx_create') RESULT(rc)
      IMPORT :: c_int, c_ptr
      INTEGER(c_int), VALUE :: device_id
      TYPE(c_ptr), INTENT(OUT) :: ctx
      INTEGER(c_int) :: rc
    END FUNCTION
    FUNCTION letkf_ctx_destroy(ctx) BIND(C, name='letkf_ctx_destroy') RESULT(rc)
      IMPORT :: c_int, c_ptr
      TYPE(c_ptr), VALUE :: ctx
      INTEGER(c_int) :: rc
    END FUNCTION
    FUNCTION letkf_ctx_synchronize(ctx) BIND(C, name='letkf_ctx_synchronize') RESULT(rc)
      IMPORT :: c_int, c_ptr
      TYPE(c_ptr), VALUE :: ctx
      INTEGER(c_int) :: rc
    END FUNCTION
    FUNCTION letkf_das_points_dev(ctx, args) BIND(C, name='letkf_das_points_dev') RESULT(rc)
      IMPORT :: c_int, c_ptr, letkf_das_args
      TYPE(c_ptr), VALUE :: ctx
      TYPE(letkf_das_args), INTENT(IN) :: args
      INTEGER(c_int) :: rc
    END FUNCTION
    ! x(p,m,v) at p*sp + m*sm + v*sv: ensmean_grd into slot k / the perturbation pass of das_letkf
    FUNCTION letkf_ens_mean_dev(ctx, k, nv, npts, x, sp, sm, sv) BIND(C, name='letkf_ens_mean_dev') RESULT(rc)
      IMPORT :: c_int, c_ptr, c_int32_t, c_int64_t
      TYPE(c_ptr), VALUE :: ctx, x
      INTEGER(c_int32_t), VALUE :: k, nv
      INTEGER(c_int64_t), VALUE :: npts, sp, sm, sv
      INTEGER(c_int) :: rc
    END FUNCTION
    FUNCTION letkf_ens_to_perturbations_dev(ctx, k, nv, npts, x, sp, sm, sv) &
        BIND(C, name='letkf_ens_to_perturbations_dev') RESULT(rc)
      IMPORT :: c_int, c_ptr, c_int32_t, c_int64_t
      TYPE(c_ptr), VALUE :: ctx, x
      INTEGER(c_int32_t), VALUE :: k, nv
      INTEGER(c_int64_t), VALUE :: npts, sp, sm, sv
      INTEGER(c_int) :: rc
    END FUNCTION
    ! device memory, straight from the HIP runtime
    FUNCTION hipMalloc(ptr, nbytes) BIND(C, name='hipMalloc') RESULT(rc)
      IMPORT :: c_int, c_ptr, c_size_t
      TYPE(c_ptr), INTENT(OUT) :: ptr
      INTEGER(c_size_t), VALUE :: nbytes
      INTEGER(c_int) :: rc
    END FUNCTION
    FUNCTION hipFree(ptr) BIND(C, name='hipFree') RESULT(rc)
      IMPORT :: c_int, c_ptr
      TYPE(c_ptr), VALUE :: ptr
      INTEGER(c_int) :: rc
    END FUNCTION
    FUNCTION hipMemcpy(dst, src, nbytes, kind) BIND(C, name='hipMemcpy') RESULT(rc)
      IMPORT :: c_int, c_ptr, c_size_t
      TYPE(c_ptr), VALUE :: dst, src
      INTEGER(c_size_t), VALUE :: nbytes
      INTEGER(c_int), VALUE :: kind
      INTEGER(c_int) :: rc
    END FUNCTION
  END INTERFACE

END MODULE letkf_amd_api
