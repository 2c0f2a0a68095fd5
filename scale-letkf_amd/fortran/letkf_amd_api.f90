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
    INTEGER(c_int32_t) :: k, nv, det_run, infl_adaptive, relax_to_inflated_prior, iv_p, iv_q_first, iv_q_last, warm_stride
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

  ! include/letkf_amd.h section 7
  TYPE, BIND(C) :: letkf_beta_params
    INTEGER(c_int32_t) :: radar_only, ihalo, jhalo, nlong, nlatg, reserved0
    REAL(c_double)     :: radar_zmax, vert_local_radar, boundary_buffer_width, dx, dy
  END TYPE letkf_beta_params

  INTERFACE
    ! das_letkf set-up (scale/letkf/letkf_tools.f90:130-267, relax_beta :1911-1948); the first three are host functions
    FUNCTION letkf_var_local_classes(nvar, nlt, var_local, n2nc, n2n, nclass) &
        BIND(C, name='letkf_var_local_classes') RESULT(rc)
      IMPORT :: c_int, c_int32_t, c_double
      INTEGER(c_int32_t), VALUE :: nvar, nlt
      REAL(c_double), INTENT(IN) :: var_local(nvar, nlt)
      INTEGER(c_int32_t), INTENT(OUT) :: n2nc(nvar), n2n(nvar), nclass
      INTEGER(c_int) :: rc
    END FUNCTION
    FUNCTION letkf_ctype_merge_groups(nctype, elm_u_ctype, typ_ctype, nid_obs, nobtype, ctype_merge, group_start, &
                                      group_member, ngroup) BIND(C, name='letkf_ctype_merge_groups') RESULT(rc)
      IMPORT :: c_int, c_int32_t
      INTEGER(c_int32_t), VALUE :: nctype, nid_obs, nobtype
      INTEGER(c_int32_t), INTENT(IN) :: elm_u_ctype(nctype), typ_ctype(nctype), ctype_merge(nid_obs, nobtype)
      INTEGER(c_int32_t), INTENT(OUT) :: group_start(nctype + 1), group_member(nctype), ngroup
      INTEGER(c_int) :: rc
    END FUNCTION
    FUNCTION letkf_radar_only(nctype, typ_ctype, typ_radar) BIND(C, name='letkf_radar_only') RESULT(flag)
      IMPORT :: c_int, c_int32_t
      INTEGER(c_int32_t), VALUE :: nctype, typ_radar
      INTEGER(c_int32_t), INTENT(IN) :: typ_ctype(nctype)
      INTEGER(c_int) :: flag
    END FUNCTION
    FUNCTION letkf_relax_beta_dev(ctx, p, nij1, nlev, rig, rjg, hgt, beta) &
        BIND(C, name='letkf_relax_beta_dev') RESULT(rc)
      IMPORT :: c_int, c_ptr, c_int32_t, c_int64_t, letkf_beta_params
      TYPE(c_ptr), VALUE :: ctx, rig, rjg, hgt, beta
      TYPE(letkf_beta_params), INTENT(IN) :: p
      INTEGER(c_int64_t), VALUE :: nij1
      INTEGER(c_int32_t), VALUE :: nlev
      INTEGER(c_int) :: rc
    END FUNCTION
    FUNCTION letkf_infl_init_dev(ctx, n, work3d, infl_mul, infl_mul_min) &
        BIND(C, name='letkf_infl_init_dev') RESULT(rc)
      IMPORT :: c_int, c_ptr, c_int64_t, c_double
      TYPE(c_ptr), VALUE :: ctx, work3d
      INTEGER(c_int64_t), VALUE :: n
      REAL(c_double), VALUE :: infl_mul, infl_mul_min
      INTEGER(c_int) :: rc
    END FUNCTION
    ! the exchange of letkf_obs.f90:1036-1046 on the host's RCCL communicator (section 8); counts is a HOST array
    FUNCTION letkf_obs_allgatherv_dev(ctx, nccl_comm, nranks, myrank, counts, row_bytes, send, recv) &
        BIND(C, name='letkf_obs_allgatherv_dev') RESULT(rc)
      IMPORT :: c_int, c_ptr, c_int32_t, c_int64_t
      TYPE(c_ptr), VALUE :: ctx, nccl_comm, send, recv
      INTEGER(c_int32_t), VALUE :: nranks, myrank
      INTEGER(c_int64_t), INTENT(IN) :: counts(nranks)
      INTEGER(c_int64_t), VALUE :: row_bytes
      INTEGER(c_int) :: rc
    END FUNCTION
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
    FUNCTION letkf_ctx_set_option(ctx, option, value) BIND(C, name='letkf_ctx_set_option') RESULT(rc)
      IMPORT :: c_int, c_ptr
      TYPE(c_ptr), VALUE :: ctx
      INTEGER(c_int), VALUE :: option, value      ! LETKF_OPT_STAGED_POLY = 1 (include/letkf_amd.h)
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
