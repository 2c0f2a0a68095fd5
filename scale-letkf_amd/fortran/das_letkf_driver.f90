!===============================================================================
! das_letkf_driver.f90 -- a Fortran host calling das_letkf_amd (letkf_tools_amd.f90) the way scale/letkf/letkf.f90:196
! calls das_letkf: fills the two derived types from a case file written by tests/test_fortran_das.py (what the
! reference's modules would hold after set_letkf_obs and read_ens_mpi), calls the routine, writes anal3d, the
! perturbations gues3d comes back as, the inflation field and the local-observation counts.
!   file (little endian, stream): int32 hdr(19); real64 r(13); then the arrays in the order read below
!===============================================================================
PROGRAM das_letkf_driver
  USE, INTRINSIC :: iso_c_binding
  USE letkf_amd_api
  USE letkf_tools_amd
  IMPLICIT NONE
  INTEGER(c_int32_t) :: hdr(19)
  REAL(c_double) :: r(13)
  INTEGER :: member, det, nij1, nlev, nv3d, nctype, nobstotal, nensobs, nens, relax, nid_obs, nobtype, nac, u, ios
  TYPE(letkf_das_nml) :: nml
  TYPE(letkf_obs_tables) :: obs
  TYPE(c_ptr) :: ctx
  INTEGER(c_int32_t), ALLOCATABLE :: i32(:), merge32(:, :), nobs_point(:, :)
  REAL(c_double), ALLOCATABLE :: rig1(:), rjg1(:), hgt1(:, :), gues3d(:, :, :, :), anal3d(:, :, :, :), work3d(:, :, :)
  CHARACTER(len=512) :: fin, fout
  INTEGER(c_int) :: rc

  CALL get_command_argument(1, fin)
  CALL get_command_argument(2, fout)
  OPEN (newunit=u, file=trim(fin), access='stream', form='unformatted', status='old', iostat=ios)
  IF (ios /= 0) STOP 3
  READ (u) hdr
  READ (u) r
  member = hdr(1); det = hdr(2); nij1 = hdr(3); nlev = hdr(4); nv3d = hdr(5); nctype = hdr(6); nobstotal = hdr(7)
  nensobs = hdr(8); relax = hdr(15); nid_obs = hdr(17); nobtype = hdr(18); nac = hdr(19)
  nens = member + 1 + det
  nml%member = member; nml%det_run = det /= 0
  nml%max_nobs_per_grid_criterion = hdr(9); nml%nlon = hdr(10); nml%nlat = hdr(11); nml%nlong = hdr(12); nml%nlatg = hdr(13)
  nml%ihalo = hdr(14); nml%jhalo = hdr(14); nml%infl_mul_adaptive = hdr(16) /= 0
  IF (relax == 1) nml%relax_alpha = r(1)
  IF (relax == 2) nml%relax_alpha_spread = r(1)
  nml%infl_mul = r(2); nml%infl_mul_min = r(3); nml%q_update_top = r(4); nml%q_sprd_max = r(5)
  nml%boundary_buffer_width = r(6); nml%radar_zmax = r(7); nml%vert_local_radar = r(8); nml%vert_local_rain_base = r(9)
  nml%dx = r(10); nml%dy = r(11); nml%i_org = r(12); nml%j_org = r(13)
  nml%iv3d_p = 5; nml%iv3d_q = 6; nml%iv3d_qlast = MIN(11, nv3d)
  ALLOCATE (nml%var_local(nv3d, 9), merge32(nid_obs, nobtype), nml%ctype_merge(nid_obs, nobtype))
  READ (u) nml%var_local
  READ (u) merge32
  nml%ctype_merge = merge32
  obs%nctype = nctype; obs%nobstotal = nobstotal; obs%nensobs = nensobs
  ALLOCATE (obs%elm_ctype(nctype), obs%elm_u_ctype(nctype), obs%typ_ctype(nctype), obs%uid_varlocal_ctype(nctype), &
            obs%max_nobs_ctype(nctype), obs%ngrd_i(nctype), obs%ngrd_j(nctype), obs%ngrdsch_i(nctype), obs%ngrdsch_j(nctype), &
            obs%ngrdext_i(nctype), obs%ngrdext_j(nctype), obs%ac_off(nctype), obs%ac_ext(nac), obs%hori_loc_ctype(nctype), &
            obs%vert_loc_ctype(nctype), i32(nctype))
  READ (u) i32; obs%elm_ctype = i32
  READ (u) i32; obs%elm_u_ctype = i32
  READ (u) i32; obs%typ_ctype = i32
  READ (u) i32; obs%uid_varlocal_ctype = i32
  READ (u) i32; obs%max_nobs_ctype = i32
  READ (u) obs%ngrd_i, obs%ngrd_j, obs%ngrdsch_i, obs%ngrdsch_j, obs%ngrdext_i, obs%ngrdext_j
  READ (u) obs%ac_off
  READ (u) obs%ac_ext
  READ (u) obs%hori_loc_ctype, obs%vert_loc_ctype
  ALLOCATE (obs%ob_ri(nobstotal), obs%ob_rj(nobstotal), obs%ob_lev(nobstotal), obs%ob_dat(nobstotal), obs%ob_err(nobstotal), &
            obs%ensval(nensobs, nobstotal), obs%val(nobstotal))
  READ (u) obs%ob_ri, obs%ob_rj, obs%ob_lev, obs%ob_dat, obs%ob_err
  READ (u) obs%ensval
  READ (u) obs%val
  ALLOCATE (rig1(nij1), rjg1(nij1), hgt1(nij1, nlev), gues3d(nij1, nlev, nens, nv3d), anal3d(nij1, nlev, nens, nv3d), &
            work3d(nij1, nlev, nv3d), nobs_point(nij1, nlev))
  READ (u) rig1, rjg1, hgt1
  READ (u) gues3d
  CLOSE (u)

  IF (letkf_amd_abi_version() < 5) STOP 4
  rc = letkf_ctx_create(0_c_int, ctx)
  IF (rc /= 0) STOP 5
  ! ---- what replaces CALL das_letkf(gues3d,gues2d,anal3d,anal2d) of scale/letkf/letkf.f90:196
  CALL das_letkf_amd(ctx, nml, obs, nij1, nlev, nens, nv3d, rig1, rjg1, hgt1, gues3d, anal3d, work3d_out=work3d, &
                     nobs_point=nobs_point)
  rc = letkf_ctx_destroy(ctx)

  OPEN (newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
  WRITE (u) anal3d
  WRITE (u) gues3d
  WRITE (u) work3d
  WRITE (u) nobs_point
  CLOSE (u)
END PROGRAM das_letkf_driver
