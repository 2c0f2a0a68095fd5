!===============================================================================
! letkf_amd_api.f90 -- ISO_C_BINDING view of EVERY device entry point of include/letkf_amd.h (sections 1b - 8), for a
! das_letkf that runs "search, then one call per level slab" on the device (letkf_tools_amd.f90 is that routine;
! INTEGRATION.md, level 2).  The derived types mirror the C structs field by field (same order, same kinds:
! tests/test_fortran_shim.py reads every field back through C); device buffers are C pointers obtained from hipMalloc
! (bound below straight from libamdhip64 -- hipfort is not needed).
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
    INTEGER(c_int64_t) :: infl_sv
  END TYPE letkf_das_args

  ! include/letkf_amd.h section 7
  TYPE, BIND(C) :: letkf_beta_params
    INTEGER(c_int32_t) :: radar_only, ihalo, jhalo, nlong, nlatg, reserved0
    REAL(c_double)     :: radar_zmax, vert_local_radar, boundary_buffer_width, dx, dy
  END TYPE letkf_beta_params


  ! include/letkf_amd.h section 1b
  TYPE, BIND(C) :: letkf_core_batch_args
    INTEGER(c_int32_t) :: ne, nobs
    INTEGER(c_int64_t) :: nbatch
    TYPE(c_ptr)        :: nobsl, hdxb, rdiag, rloc, dep, depd, parm_infl, trans, transm, pao, transmd
    INTEGER(c_int32_t) :: rdiag_wloc, infl_update
    TYPE(c_ptr)        :: status, nsweep
  END TYPE letkf_core_batch_args

  ! section 3: what set_letkf_obs leaves behind (scale/letkf/letkf_obs.f90:35-72); every pointer is a DEVICE pointer, the
  ! struct itself is passed by reference from the host
  TYPE, BIND(C) :: letkf_search_tables
    INTEGER(c_int32_t) :: nctype, ngroup, criterion, nlon, nlat, limit_hint
    REAL(c_double)     :: dx, dy, i_org, j_org, rain_base
    TYPE(c_ptr)        :: group_start, group_member
    TYPE(c_ptr)        :: vmode, hori_loc, vert_loc, varloc, max_nobs
    TYPE(c_ptr)        :: ngrd_i, ngrd_j, ngrdsch_i, ngrdsch_j, ngrdext_i, ngrdext_j
    TYPE(c_ptr)        :: ac_off, ac_ext
    TYPE(c_ptr)        :: ob_ri, ob_rj, ob_lev, ob_dat, ob_err
  END TYPE letkf_search_tables

  ! section 4
  TYPE, BIND(C) :: letkf_state_consts
    REAL(c_double)     :: rdry, rvap, cvdry, pre00
    REAL(c_double)     :: tracer_cv(8)
    INTEGER(c_int32_t) :: iv_rho, iv_rhou, iv_rhov, iv_rhow, iv_rhot
    INTEGER(c_int32_t) :: iv_u, iv_v, iv_w, iv_t, iv_p
    INTEGER(c_int32_t) :: iv_q
    INTEGER(c_int32_t) :: positive_definite_q, positive_definite_qhyd
    INTEGER(c_int32_t) :: reserved0
  END TYPE letkf_state_consts

  ! section 5
  TYPE, BIND(C) :: letkf_qc_params
    INTEGER(c_int32_t) :: member, det_run, use_radar_ref, use_radar_vr, min_radar_ref_member, min_radar_ref_member_obsref
    REAL(c_double)     :: radar_ref_thres_dbz
    REAL(c_double)     :: gross_error, gross_error_rain, gross_error_radar_ref, gross_error_radar_vr, gross_error_radar_prh, &
                          gross_error_tcx, gross_error_tcy, gross_error_tcp
    ! the -DH08 build of the reference (ABI 6).  h08 = 0: compiled out -- SET IT, a derived type has no zero default
    INTEGER(c_int32_t) :: h08, h08_min_cld_member
    REAL(c_double)     :: h08_limit_lev, gross_error_h08, h08_bt_min
    TYPE(c_ptr)        :: h08_lev, h08_val2
  END TYPE letkf_qc_params
  TYPE, BIND(C) :: letkf_mesh
    INTEGER(c_int32_t) :: nctype, nlon, nlat, ihalo, jhalo, rank_i, rank_j, fix_ij_obsgrd
    TYPE(c_ptr)        :: ngrd_i, ngrd_j                 ! HOST [nctype]
  END TYPE letkf_mesh
  TYPE, BIND(C) :: letkf_halo_layout
    INTEGER(c_int32_t) :: nctype, nprocs, prc_num_x, myrank
    TYPE(c_ptr)        :: ngrd_i, ngrd_j, ngrdsch_i, ngrdsch_j   ! HOST [nctype]
  END TYPE letkf_halo_layout

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
    ! (8b) the pairwise exchange with true counts (halo-only observation exchange; transport of the member <-> point transpose),
    ! the all-reduce of the mesh-cell counts (letkf_obs.f90:826-833) and scatter / gather_grd_mpi_alltoall
    ! (common_mpi_scale.f90:1279-1396) on the host's RCCL communicator; counts / offsets are HOST arrays
    FUNCTION letkf_alltoallv_dev(ctx, nccl_comm, nranks, myrank, send_counts, send_offs, recv_counts, recv_offs, row_bytes, &
        send, recv) BIND(C, name='letkf_alltoallv_dev') RESULT(rc)
      IMPORT :: c_int, c_ptr, c_int32_t, c_int64_t
      TYPE(c_ptr), VALUE :: ctx, nccl_comm, send, recv
      INTEGER(c_int32_t), VALUE :: nranks, myrank
      INTEGER(c_int64_t), INTENT(IN) :: send_counts(nranks), send_offs(nranks), recv_counts(nranks), recv_offs(nranks)
      INTEGER(c_int64_t), VALUE :: row_bytes
      INTEGER(c_int) :: rc
    END FUNCTION
    FUNCTION letkf_allreduce_sum_i32_dev(ctx, nccl_comm, nranks, count, buf) &
        BIND(C, name='letkf_allreduce_sum_i32_dev') RESULT(rc)
      IMPORT :: c_int, c_ptr, c_int32_t, c_int64_t
      TYPE(c_ptr), VALUE :: ctx, nccl_comm, buf
      INTEGER(c_int32_t), VALUE :: nranks
      INTEGER(c_int64_t), VALUE :: count
      INTEGER(c_int) :: rc
    END FUNCTION
    FUNCTION letkf_members_alltoall_dev(ctx, nccl_comm, nranks, myrank, dir, nlev, nlon, nlat, nv3d, mstart, mcount, v3dg, x, &
        sp, sm, sv) BIND(C, name='letkf_members_alltoall_dev') RESULT(rc)
      IMPORT :: c_int, c_ptr, c_int32_t, c_int64_t
      TYPE(c_ptr), VALUE :: ctx, nccl_comm, v3dg, x
      INTEGER(c_int32_t), VALUE :: nranks, myrank, dir, nlev, nlon, nlat, nv3d, mstart, mcount
      INTEGER(c_int64_t), VALUE :: sp, sm, sv
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
      INTEGER(c_int), VALUE :: option, value      ! LETKF_OPT_STAGED_POLY = 1, LETKF_OPT_COLUMN_SURVIVORS = 2, LETKF_OPT_LIMITED_RINGS = 3, LETKF_OPT_RING_BATCH_MB = 4, LETKF_OPT_RING_RELEASE = 5, LETKF_OPT_SMALL_K_TRIO = 6 (include/letkf_amd.h)
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
    ! ---- section 1b: batched letkf_core on device pointers
    FUNCTION letkf_core_batch_dev(ctx, args) BIND(C, name='letkf_core_batch_dev') RESULT(rc)
      IMPORT :: c_int, c_ptr, letkf_core_batch_args
      TYPE(c_ptr), VALUE :: ctx
      TYPE(letkf_core_batch_args), INTENT(IN) :: args
      INTEGER(c_int) :: rc
    END FUNCTION
    FUNCTION letkf_ctx_set_stream(ctx, hip_stream) BIND(C, name='letkf_ctx_set_stream') RESULT(rc)
      IMPORT :: c_int, c_ptr
      TYPE(c_ptr), VALUE :: ctx, hip_stream
      INTEGER(c_int) :: rc
    END FUNCTION
    ! ---- section 3: obs_local on the device (scale/letkf/letkf_tools.f90:1325-1759).  Two-phase CSR build: fill = 0 ->
    ! counts; scan them into obs_off (npts + 1, int64); fill = 1 -> obs_idx / rdiag_l / rloc_l
    FUNCTION letkf_obs_search_dev(ctx, tables, npts, ri, rj, rlev, rz, fill, counts, obs_off, obs_idx, rdiag_l, rloc_l) &
        BIND(C, name='letkf_obs_search_dev') RESULT(rc)
      IMPORT :: c_int, c_ptr, c_int32_t, c_int64_t, letkf_search_tables
      TYPE(c_ptr), VALUE :: ctx, ri, rj, rlev, rz, counts, obs_off, obs_idx, rdiag_l, rloc_l
      TYPE(letkf_search_tables), INTENT(IN) :: tables
      INTEGER(c_int64_t), VALUE :: npts
      INTEGER(c_int32_t), VALUE :: fill
      INTEGER(c_int) :: rc
    END FUNCTION
    ! points p = ij + nij1*lev (gues3d's order): rig / rjg [nij1], rlev / rz [nij1*nlev]; nobs_ctype / cutd_ctype: the
    ! NOBS_OUT inputs [nij1*nlev][nctype] or c_null_ptr
    FUNCTION letkf_obs_search_columns_dev(ctx, tables, nij1, nlev, rig, rjg, rlev, rz, fill, counts, obs_off, obs_idx, &
                                          rdiag_l, rloc_l, nobs_ctype, cutd_ctype) &
        BIND(C, name='letkf_obs_search_columns_dev') RESULT(rc)
      IMPORT :: c_int, c_ptr, c_int32_t, c_int64_t, letkf_search_tables
      TYPE(c_ptr), VALUE :: ctx, rig, rjg, rlev, rz, counts, obs_off, obs_idx, rdiag_l, rloc_l, nobs_ctype, cutd_ctype
      TYPE(letkf_search_tables), INTENT(IN) :: tables
      INTEGER(c_int64_t), VALUE :: nij1
      INTEGER(c_int32_t), VALUE :: nlev, fill
      INTEGER(c_int) :: rc
    END FUNCTION
    FUNCTION letkf_das_points_fused_dev(ctx, args, tables, ri, rj, rlev, rz, nobs_out) &
        BIND(C, name='letkf_das_points_fused_dev') RESULT(rc)
      IMPORT :: c_int, c_ptr, letkf_das_args, letkf_search_tables
      TYPE(c_ptr), VALUE :: ctx, ri, rj, rlev, rz, nobs_out
      TYPE(letkf_das_args), INTENT(IN) :: args
      TYPE(letkf_search_tables), INTENT(IN) :: tables
      INTEGER(c_int) :: rc
    END FUNCTION
    ! das_letkf's main loop for a whole subdomain in one call: column search + loop body by slabs of levels (section 3c)
    FUNCTION letkf_das_columns_dev(ctx, args, tables, nij1, nlev, rig, rjg, rlev, rz, list_bytes, nobs_out) &
        BIND(C, name='letkf_das_columns_dev') RESULT(rc)
      IMPORT :: c_int, c_ptr, c_int32_t, c_int64_t, letkf_das_args, letkf_search_tables
      TYPE(c_ptr), VALUE :: ctx, rig, rjg, rlev, rz, nobs_out
      TYPE(letkf_das_args), INTENT(IN) :: args
      TYPE(letkf_search_tables), INTENT(IN) :: tables
      INTEGER(c_int64_t), VALUE :: nij1, list_bytes
      INTEGER(c_int32_t), VALUE :: nlev
      INTEGER(c_int) :: rc
    END FUNCTION
    ! ---- section 4: the steps either side of the loop (row f3)
    FUNCTION letkf_state_trans_dev(ctx, c, nlev, nlon, nlat, nv3d, v3dg, inverse) &
        BIND(C, name='letkf_state_trans_dev') RESULT(rc)
      IMPORT :: c_int, c_ptr, c_int32_t, letkf_state_consts
      TYPE(c_ptr), VALUE :: ctx, v3dg
      TYPE(letkf_state_consts), INTENT(IN) :: c
      INTEGER(c_int32_t), VALUE :: nlev, nlon, nlat, nv3d, inverse
      INTEGER(c_int) :: rc
    END FUNCTION
    FUNCTION letkf_member_points_dev(ctx, dir, nlev, nlon, nlat, nv3d, np, rank, m, v3dg, x, nij1, sp, sm, sv) &
        BIND(C, name='letkf_member_points_dev') RESULT(rc)
      IMPORT :: c_int, c_ptr, c_int32_t, c_int64_t
      TYPE(c_ptr), VALUE :: ctx, v3dg, x
      INTEGER(c_int32_t), VALUE :: dir, nlev, nlon, nlat, nv3d, np, rank, m
      INTEGER(c_int64_t), VALUE :: nij1, sp, sm, sv
      INTEGER(c_int) :: rc
    END FUNCTION
    FUNCTION letkf_ens_spread_dev(ctx, k, nv, npts, x, sp, sm, sv, sprd) BIND(C, name='letkf_ens_spread_dev') RESULT(rc)
      IMPORT :: c_int, c_ptr, c_int32_t, c_int64_t
      TYPE(c_ptr), VALUE :: ctx, x, sprd
      INTEGER(c_int32_t), VALUE :: k, nv
      INTEGER(c_int64_t), VALUE :: npts, sp, sm, sv
      INTEGER(c_int) :: rc
    END FUNCTION
    ! ---- section 5: set_letkf_obs on the device (row f2)
    FUNCTION letkf_obs_departure_dev(ctx, p, nobs, elm, dat, err, ensval, kld, val, qc) &
        BIND(C, name='letkf_obs_departure_dev') RESULT(rc)
      IMPORT :: c_int, c_ptr, c_int64_t, letkf_qc_params
      TYPE(c_ptr), VALUE :: ctx, elm, dat, err, ensval, val, qc
      TYPE(letkf_qc_params), INTENT(IN) :: p
      INTEGER(c_int64_t), VALUE :: nobs, kld
      INTEGER(c_int) :: rc
    END FUNCTION
    FUNCTION letkf_obs_mesh_sort_dev(ctx, mesh, nobs, ctype, ri, rj, qc, n_cell, key, nsorted) &
        BIND(C, name='letkf_obs_mesh_sort_dev') RESULT(rc)
      IMPORT :: c_int, c_ptr, c_int64_t, letkf_mesh
      TYPE(c_ptr), VALUE :: ctx, ctype, ri, rj, qc, n_cell, key
      TYPE(letkf_mesh), INTENT(IN) :: mesh
      INTEGER(c_int64_t), VALUE :: nobs
      INTEGER(c_int64_t), INTENT(OUT) :: nsorted
      INTEGER(c_int) :: rc
    END FUNCTION
    FUNCTION letkf_obs_halo_plan_dev(ctx, layout, n_all, ac_ext, src_row, cap, nobstotal) &
        BIND(C, name='letkf_obs_halo_plan_dev') RESULT(rc)
      IMPORT :: c_int, c_ptr, c_int64_t, letkf_halo_layout
      TYPE(c_ptr), VALUE :: ctx, n_all, ac_ext, src_row
      TYPE(letkf_halo_layout), INTENT(IN) :: layout
      INTEGER(c_int64_t), VALUE :: cap
      INTEGER(c_int64_t), INTENT(OUT) :: nobstotal
      INTEGER(c_int) :: rc
    END FUNCTION
    FUNCTION letkf_obs_gather_rows_dev(ctx, nrows, src_row, ncols, src, ld_src, dst, ld_dst) &
        BIND(C, name='letkf_obs_gather_rows_dev') RESULT(rc)
      IMPORT :: c_int, c_ptr, c_int32_t, c_int64_t
      TYPE(c_ptr), VALUE :: ctx, src_row, src, dst
      INTEGER(c_int64_t), VALUE :: nrows, ld_src, ld_dst
      INTEGER(c_int32_t), VALUE :: ncols
      INTEGER(c_int) :: rc
    END FUNCTION
    FUNCTION letkf_obs_gather_i32_dev(ctx, nrows, src_row, src, dst) BIND(C, name='letkf_obs_gather_i32_dev') RESULT(rc)
      IMPORT :: c_int, c_ptr, c_int64_t
      TYPE(c_ptr), VALUE :: ctx, src_row, src, dst
      INTEGER(c_int64_t), VALUE :: nrows
      INTEGER(c_int) :: rc
    END FUNCTION
    ! ---- section 6: after the loop (row f4); elem_uid is a HOST array
    FUNCTION letkf_monit_dep_dev(ctx, nid, elem_uid, nn, elm, dep, qc, nobs, bias, rmse) &
        BIND(C, name='letkf_monit_dep_dev') RESULT(rc)
      IMPORT :: c_int, c_ptr, c_int32_t, c_int64_t
      TYPE(c_ptr), VALUE :: ctx, elm, dep, qc, nobs, bias, rmse
      INTEGER(c_int32_t), VALUE :: nid
      INTEGER(c_int32_t), INTENT(IN) :: elem_uid(nid)
      INTEGER(c_int64_t), VALUE :: nn
      INTEGER(c_int) :: rc
    END FUNCTION
    FUNCTION letkf_additive_inflation_dev(ctx, k, nv, npts, nij1, anal, add, sp, sm, sv, infl_add, weight, qmean, q_sp, &
                                          q_sv, iv_q_first, iv_q_last, ishuf) &
        BIND(C, name='letkf_additive_inflation_dev') RESULT(rc)
      IMPORT :: c_int, c_ptr, c_int32_t, c_int64_t, c_double
      TYPE(c_ptr), VALUE :: ctx, anal, add, weight, qmean, ishuf
      INTEGER(c_int32_t), VALUE :: k, nv, iv_q_first, iv_q_last
      INTEGER(c_int64_t), VALUE :: npts, nij1, sp, sm, sv, q_sp, q_sv
      REAL(c_double), VALUE :: infl_add
      INTEGER(c_int) :: rc
    END FUNCTION
    FUNCTION letkf_addinfl_weight_dev(ctx, nij1, rig, rjg, nob, ob_ri, ob_rj, dx, dy, hori_loc, weight) &
        BIND(C, name='letkf_addinfl_weight_dev') RESULT(rc)
      IMPORT :: c_int, c_ptr, c_int64_t, c_double
      TYPE(c_ptr), VALUE :: ctx, rig, rjg, ob_ri, ob_rj, weight
      INTEGER(c_int64_t), VALUE :: nij1, nob
      REAL(c_double), VALUE :: dx, dy, hori_loc
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
