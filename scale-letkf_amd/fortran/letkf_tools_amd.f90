!===============================================================================
! letkf_tools_amd.f90 -- das_letkf on the MI355X: the Fortran routine a maintainer CALLs instead of the reference's
!   CALL das_letkf(gues3d,gues2d,anal3d,anal2d)        scale/letkf/letkf.f90:196, scale/letkf/letkf_tools.f90:50-58
! What the reference's routine reads from module state (common_mpi_scale.f90:37-66 nij1, rig1, rjg1, hgt1, nens, mmean,
! mmdet; common_scale.f90:117-121; letkf_obs.f90:35-72 obs / obsda_sort / obsgrd / *_ctype; ~25 namelist variables)
! arrives here as explicit arguments: two derived types that the caller fills from those modules once per analysis.
! The routine itself follows das_letkf step by step, every step on the device through include/letkf_amd.h:
!   :130-157  variable-localisation classes            letkf_var_local_classes          (host helper)
!   :167-192  merge groups of the obs-number limit      letkf_ctype_merge_groups         (host helper)
!   :197-203  radar_only                                letkf_radar_only                 (host helper)
!   :209-230  members -> perturbations                  letkf_ens_to_perturbations_dev
!   :237-267  multiplicative-inflation field            letkf_infl_init_dev
!   :1911-1948 relax_beta for every point               letkf_relax_beta_dev
!   :313-527  main loop: obs_local + letkf_core + relaxation + transform, per variable-localisation class
!                                                      letkf_das_columns_dev (column search + loop body by slabs of levels)
!   letkf.f90:207  ensmean_grd on the analysis          letkf_ens_mean_dev
! gues3d / anal3d keep the reference's shape and meaning: gues3d(nij1,nlev,nens,nv3d) INOUT (members 1..MEMBER come back
! as perturbations, slot mmean = MEMBER+1 holds the mean, mmdet = MEMBER+2 the deterministic member), anal3d OUT.
! 2-D variables: the reference is built with nv2d = 0 (common_scale.f90:53); they are not arguments here (INTEGRATION.md
! has the level-1 recipe).  Errors: a non-zero point status prints and STOPs 2, as common_mtx.f90:61-64 does.
!===============================================================================
MODULE letkf_tools_amd
  USE, INTRINSIC :: iso_c_binding
  USE letkf_amd_api
  IMPLICIT NONE
  PRIVATE
  PUBLIC :: das_letkf_amd, letkf_das_nml, letkf_obs_tables, letkf_vmode

  INTEGER, PARAMETER :: r_size = c_double
  INTEGER, PARAMETER :: nlt = 9               ! columns of var_local: VAR_LOCAL_UV .. VAR_LOCAL_H08 (letkf_tools.f90:130-138)
  ! common_obs_scale.f90:57-58, :87-92
  INTEGER, PARAMETER :: id_ps_obs = 14593, id_rain_obs = 19999, typ_pharad = 22

  ! namelist values and grid constants das_letkf reads (common_nml.f90, scale_grid_index, common_scale.f90)
  TYPE :: letkf_das_nml
    INTEGER :: member = 0                     ! MEMBER
    LOGICAL :: det_run = .FALSE.              ! DET_RUN
    REAL(r_size) :: infl_mul = 1.0d0          ! INFL_MUL (> 0: constant field; <= 0: work3d_in is used, :237-267)
    REAL(r_size) :: infl_mul_min = -1.0d0     ! INFL_MUL_MIN
    LOGICAL :: infl_mul_adaptive = .FALSE.    ! INFL_MUL_ADAPTIVE
    REAL(r_size) :: relax_alpha = 0.0d0       ! RELAX_ALPHA (RTPP)
    REAL(r_size) :: relax_alpha_spread = 0.0d0   ! RELAX_ALPHA_SPREAD (RTPS)
    LOGICAL :: relax_to_inflated_prior = .FALSE.
    REAL(r_size) :: q_update_top = 0.0d0, q_sprd_max = -1.0d0
    REAL(r_size) :: boundary_buffer_width = 0.0d0, radar_zmax = 99.0d3
    REAL(r_size) :: vert_local_radar = 0.0d0  ! MAX(VERT_LOCAL(22), VERT_LOCAL_RADAR_VR) (:1925)
    REAL(r_size) :: vert_local_rain_base = 85000.0d0
    INTEGER :: max_nobs_per_grid_criterion = 1
    REAL(r_size) :: dx = 0.0d0, dy = 0.0d0    ! DX, DY
    INTEGER :: ihalo = 2, jhalo = 2           ! IHALO, JHALO
    INTEGER :: nlon = 0, nlat = 0             ! subdomain interior size
    INTEGER :: nlong = 0, nlatg = 0           ! global interior size
    REAL(r_size) :: i_org = 0.0d0, j_org = 0.0d0   ! ri - i_org = ril - IHALO - 0.5 of ij_obsgrd_ext (letkf_obs.f90:1221)
    INTEGER :: iv3d_p = 5, iv3d_q = 6, iv3d_qlast = 11   ! 1-based (common_scale.f90:36-51)
    INTEGER(c_int64_t) :: list_bytes = 0                 ! device workspace for the local-observation lists of a slab of levels, or -- list-free route, include/letkf_amd.h (3c) -- the horizontal survivors of a batch of columns (0: 8 GiB)
    REAL(r_size), ALLOCATABLE :: var_local(:, :)          ! (nv3d, 9)
    INTEGER, ALLOCATABLE :: ctype_merge(:, :)             ! (nid_obs, nobtype), > 0 = merge class (:167-178)
  END TYPE letkf_das_nml

  ! what set_letkf_obs leaves behind (letkf_obs.f90:35-72; common_obs_scale.f90:112-130), host arrays
  TYPE :: letkf_obs_tables
    INTEGER :: nctype = 0, nobstotal = 0, nensobs = 0
    INTEGER, ALLOCATABLE :: elm_ctype(:), elm_u_ctype(:), typ_ctype(:)   ! obs element id, uid_obs(elm), report type per combined type
    INTEGER, ALLOCATABLE :: uid_varlocal_ctype(:)         ! uid_obs_varlocal(elm): column of var_local, 1..9
    INTEGER, ALLOCATABLE :: max_nobs_ctype(:)             ! MAX_NOBS_PER_GRID(typ)
    REAL(r_size), ALLOCATABLE :: hori_loc_ctype(:), vert_loc_ctype(:)
    INTEGER(c_int32_t), ALLOCATABLE :: ngrd_i(:), ngrd_j(:), ngrdsch_i(:), ngrdsch_j(:), ngrdext_i(:), ngrdext_j(:)
    INTEGER(c_int64_t), ALLOCATABLE :: ac_off(:)          ! start of every ctype's obsgrd%ac_ext inside ac_ext (0-based)
    INTEGER(c_int32_t), ALLOCATABLE :: ac_ext(:)          ! obsgrd(ic)%ac_ext(0:ngrdext_i, 1:ngrdext_j), ctypes concatenated, cumulative over ctypes
    REAL(r_size), ALLOCATABLE :: ob_ri(:), ob_rj(:), ob_lev(:), ob_dat(:), ob_err(:)   ! obs(set)%...(idx) in obsda_sort order
    REAL(r_size), ALLOCATABLE :: ensval(:, :)             ! obsda_sort%ensval(nensobs, nobstotal): 1..MEMBER perturbations, mmdetobs = MEMBER+1
    REAL(r_size), ALLOCATABLE :: val(:)                   ! obsda_sort%val
  END TYPE letkf_obs_tables

CONTAINS

  ! vertical coordinate of a combined type (obs_local_cal, letkf_tools.f90:1851-1865): 0 |dln p| (obs lev), 1 |dz| (type 22),
  ! 2 ps (obs dat), 3 rain base.  The -DH08 build's report type 23 (H08IRB, :1859-1861) is mode 0 as well: its rows carry
  ! obsda_sort%lev (the sensitive height) in letkf_obs_tables%lev instead of obs%lev -- the tables are per obsda_sort row.
  PURE FUNCTION letkf_vmode(elm, typ) RESULT(vm)
    INTEGER, INTENT(IN) :: elm, typ
    INTEGER(c_int32_t) :: vm
    IF (elm == id_ps_obs) THEN
      vm = 2
    ELSE IF (elm == id_rain_obs) THEN
      vm = 3
    ELSE IF (typ == typ_pharad) THEN
      vm = 1
    ELSE
      vm = 0
    END IF
  END FUNCTION letkf_vmode

  SUBROUTINE das_letkf_amd(ctx, nml, obs, nij1, nlev, nens, nv3d, rig1, rjg1, hgt1, gues3d, anal3d, work3d_in, work3d_out, nobs_point)
    TYPE(c_ptr), INTENT(IN) :: ctx                        ! letkf_ctx_create
    TYPE(letkf_das_nml), INTENT(IN) :: nml
    TYPE(letkf_obs_tables), INTENT(IN), TARGET :: obs
    INTEGER, INTENT(IN) :: nij1, nlev, nens, nv3d
    REAL(r_size), INTENT(IN), TARGET :: rig1(nij1), rjg1(nij1), hgt1(nij1, nlev)
    REAL(r_size), INTENT(INOUT), TARGET :: gues3d(nij1, nlev, nens, nv3d)
    REAL(r_size), INTENT(OUT), TARGET :: anal3d(nij1, nlev, nens, nv3d)
    REAL(r_size), INTENT(IN), OPTIONAL, TARGET :: work3d_in(nij1, nlev, nv3d)    ! INFL_MUL_IN field (INFL_MUL <= 0)
    REAL(r_size), INTENT(OUT), OPTIONAL, TARGET :: work3d_out(nij1, nlev, nv3d)  ! the (adaptively updated) inflation field
    INTEGER(c_int32_t), INTENT(OUT), OPTIONAL, TARGET :: nobs_point(nij1, nlev)  ! local observations of the first class

    INTEGER :: k, mmean, mmdet, n, ic, icl, nclass, ngroup, npts
    INTEGER(c_int32_t), ALLOCATABLE, TARGET :: n2nc(:), n2n(:), group_start(:), group_member(:), vmode(:), elm_u(:), typ(:), &
                                               counts(:), status(:), mx(:), merge32(:, :)
    REAL(r_size), ALLOCATABLE, TARGET :: varloc(:), work3d(:, :, :)
    INTEGER(c_int32_t) :: nclass32, ngroup32
    INTEGER(c_int64_t) :: sm, sv
    INTEGER(c_size_t) :: nb_state
    TYPE(c_ptr) :: d_gues, d_anal, d_rig, d_rjg, d_hgt, d_beta, d_infl, d_ens, d_val, d_counts, &
                   d_status, d_varloc
    TYPE(c_ptr) :: d_gs, d_gm, d_vm, d_hl, d_vl, d_mx, d_gi, d_gj, d_si, d_sj, d_ei, d_ej, d_aco, d_ace, d_ri, d_rj, d_lev, &
                   d_dat, d_err
    TYPE(letkf_search_tables) :: t
    TYPE(letkf_beta_params) :: bp
    TYPE(letkf_das_args) :: a
    INTEGER(c_int32_t) :: mask

    k = nml%member
    mmean = k + 1
    mmdet = k + 2
    npts = nij1*nlev
    sm = INT(npts, c_int64_t)
    sv = sm*nens
    IF (nens /= k + 1 + MERGE(1, 0, nml%det_run)) CALL fail('nens must be MEMBER + 1 (+ 1 with DET_RUN)')
    IF (obs%nensobs < k + MERGE(1, 0, nml%det_run)) CALL fail('obsda_sort%ensval has too few rows')

    ! ---- set-up (letkf_tools.f90:130-203): host helpers of the C ABI
    ALLOCATE (n2nc(nv3d), n2n(nv3d))
    CALL chk(letkf_var_local_classes(INT(nv3d, c_int32_t), INT(nlt, c_int32_t), nml%var_local, n2nc, n2n, nclass32), &
             'letkf_var_local_classes')
    nclass = nclass32
    ALLOCATE (group_start(obs%nctype + 1), group_member(MAX(obs%nctype, 1)), elm_u(MAX(obs%nctype, 1)), typ(MAX(obs%nctype, 1)))
    elm_u(1:obs%nctype) = obs%elm_u_ctype(1:obs%nctype)
    typ(1:obs%nctype) = obs%typ_ctype(1:obs%nctype)
    ALLOCATE (merge32(SIZE(nml%ctype_merge, 1), SIZE(nml%ctype_merge, 2)))
    merge32 = nml%ctype_merge
    CALL chk(letkf_ctype_merge_groups(INT(obs%nctype, c_int32_t), elm_u, typ, INT(SIZE(merge32, 1), c_int32_t), &
                                      INT(SIZE(merge32, 2), c_int32_t), merge32, group_start, group_member, ngroup32), &
             'letkf_ctype_merge_groups')
    ngroup = ngroup32
    ALLOCATE (vmode(MAX(obs%nctype, 1)), mx(MAX(obs%nctype, 1)), varloc(MAX(obs%nctype, 1)))
    DO ic = 1, obs%nctype
      vmode(ic) = letkf_vmode(obs%elm_ctype(ic), obs%typ_ctype(ic))
      mx(ic) = obs%max_nobs_ctype(ic)
    END DO

    ! ---- upload: state, point coordinates, observation tables
    nb_state = 8_c_size_t*INT(npts, c_size_t)*nens*nv3d
    d_gues = up(c_loc(gues3d), nb_state)
    CALL chk(hipMalloc(d_anal, nb_state), 'hipMalloc anal3d')
    d_rig = up(c_loc(rig1), 8_c_size_t*nij1)
    d_rjg = up(c_loc(rjg1), 8_c_size_t*nij1)
    d_hgt = up(c_loc(hgt1), 8_c_size_t*npts)
    CALL chk(hipMalloc(d_beta, 8_c_size_t*npts), 'hipMalloc beta')
    d_ens = up(c_loc(obs%ensval), 8_c_size_t*obs%nensobs*MAX(obs%nobstotal, 1))
    d_val = up(c_loc(obs%val), 8_c_size_t*MAX(obs%nobstotal, 1))
    d_gs = up(c_loc(group_start), 4_c_size_t*(ngroup + 1))
    d_gm = up(c_loc(group_member), 4_c_size_t*MAX(obs%nctype, 1))
    d_vm = up(c_loc(vmode), 4_c_size_t*MAX(obs%nctype, 1))
    d_hl = up(c_loc(obs%hori_loc_ctype), 8_c_size_t*MAX(obs%nctype, 1))
    d_vl = up(c_loc(obs%vert_loc_ctype), 8_c_size_t*MAX(obs%nctype, 1))
    d_mx = up(c_loc(mx), 4_c_size_t*MAX(obs%nctype, 1))
    d_gi = up(c_loc(obs%ngrd_i), 4_c_size_t*MAX(obs%nctype, 1))
    d_gj = up(c_loc(obs%ngrd_j), 4_c_size_t*MAX(obs%nctype, 1))
    d_si = up(c_loc(obs%ngrdsch_i), 4_c_size_t*MAX(obs%nctype, 1))
    d_sj = up(c_loc(obs%ngrdsch_j), 4_c_size_t*MAX(obs%nctype, 1))
    d_ei = up(c_loc(obs%ngrdext_i), 4_c_size_t*MAX(obs%nctype, 1))
    d_ej = up(c_loc(obs%ngrdext_j), 4_c_size_t*MAX(obs%nctype, 1))
    d_aco = up(c_loc(obs%ac_off), 8_c_size_t*MAX(obs%nctype, 1))
    d_ace = up(c_loc(obs%ac_ext), 4_c_size_t*SIZE(obs%ac_ext))
    d_ri = up(c_loc(obs%ob_ri), 8_c_size_t*MAX(obs%nobstotal, 1))
    d_rj = up(c_loc(obs%ob_rj), 8_c_size_t*MAX(obs%nobstotal, 1))
    d_lev = up(c_loc(obs%ob_lev), 8_c_size_t*MAX(obs%nobstotal, 1))
    d_dat = up(c_loc(obs%ob_dat), 8_c_size_t*MAX(obs%nobstotal, 1))
    d_err = up(c_loc(obs%ob_err), 8_c_size_t*MAX(obs%nobstotal, 1))
    CALL chk(hipMalloc(d_varloc, 8_c_size_t*MAX(obs%nctype, 1)), 'hipMalloc varloc')
    CALL chk(hipMalloc(d_counts, 4_c_size_t*npts), 'hipMalloc counts')
    CALL chk(hipMalloc(d_status, 4_c_size_t*npts), 'hipMalloc status')
    ALLOCATE (counts(npts), status(npts))

    ! ---- members -> perturbations (:209-230); the mean already sits in slot mmean (write_ensmean, letkf.f90:176)
    CALL chk(letkf_ens_to_perturbations_dev(ctx, INT(k, c_int32_t), INT(nv3d, c_int32_t), sm, d_gues, 1_c_int64_t, sm, sv), &
             'letkf_ens_to_perturbations_dev')

    ! ---- multiplicative-inflation field (:237-267)
    ALLOCATE (work3d(nij1, nlev, nv3d))
    IF (nml%infl_mul > 0.0d0) THEN
      work3d = nml%infl_mul
    ELSE
      IF (.NOT. PRESENT(work3d_in)) CALL fail('INFL_MUL <= 0 needs work3d_in')
      work3d = work3d_in
    END IF
    d_infl = up(c_loc(work3d), 8_c_size_t*npts*nv3d)
    CALL chk(letkf_infl_init_dev(ctx, sm*nv3d, d_infl, nml%infl_mul, nml%infl_mul_min), 'letkf_infl_init_dev')

    ! ---- relax_beta (:1911-1948) for all points
    bp%radar_only = letkf_radar_only(INT(obs%nctype, c_int32_t), typ, INT(typ_pharad, c_int32_t))
    bp%ihalo = nml%ihalo; bp%jhalo = nml%jhalo; bp%nlong = nml%nlong; bp%nlatg = nml%nlatg; bp%reserved0 = 0
    bp%radar_zmax = nml%radar_zmax; bp%vert_local_radar = nml%vert_local_radar
    bp%boundary_buffer_width = nml%boundary_buffer_width; bp%dx = nml%dx; bp%dy = nml%dy
    CALL chk(letkf_relax_beta_dev(ctx, bp, INT(nij1, c_int64_t), INT(nlev, c_int32_t), d_rig, d_rjg, d_hgt, d_beta), &
             'letkf_relax_beta_dev')

    ! ---- the tables of obs_local
    t%nctype = obs%nctype; t%ngroup = ngroup; t%criterion = nml%max_nobs_per_grid_criterion
    t%nlon = nml%nlon; t%nlat = nml%nlat
    t%limit_hint = MERGE(2, 1, ANY(mx(1:obs%nctype) > 0))
    t%dx = nml%dx; t%dy = nml%dy; t%i_org = nml%i_org; t%j_org = nml%j_org; t%rain_base = nml%vert_local_rain_base
    t%group_start = d_gs; t%group_member = d_gm; t%vmode = d_vm; t%hori_loc = d_hl; t%vert_loc = d_vl; t%varloc = d_varloc
    t%max_nobs = d_mx; t%ngrd_i = d_gi; t%ngrd_j = d_gj; t%ngrdsch_i = d_si; t%ngrdsch_j = d_sj
    t%ngrdext_i = d_ei; t%ngrdext_j = d_ej; t%ac_off = d_aco; t%ac_ext = d_ace
    t%ob_ri = d_ri; t%ob_rj = d_rj; t%ob_lev = d_lev; t%ob_dat = d_dat; t%ob_err = d_err

    ! ---- main loop (:313-527), one pass per variable-localisation class: obs_local for every point, then the loop body
    a%k = k; a%nv = nv3d; a%det_run = MERGE(1, 0, nml%det_run); a%infl_adaptive = MERGE(1, 0, nml%infl_mul_adaptive)
    a%relax_to_inflated_prior = MERGE(1, 0, nml%relax_to_inflated_prior)
    a%iv_p = nml%iv3d_p - 1; a%iv_q_first = nml%iv3d_q - 1; a%iv_q_last = nml%iv3d_qlast - 1
    a%warm_stride = MERGE(nij1, 0, nlev > 1)        ! warm-start runs up the columns (vertical neighbours share their observations)
    a%relax_alpha = nml%relax_alpha; a%relax_alpha_spread = nml%relax_alpha_spread
    a%q_update_top = nml%q_update_top; a%q_sprd_max = nml%q_sprd_max
    a%npts = npts
    a%obs_off = c_null_ptr; a%obs_idx = c_null_ptr; a%rdiag_l = c_null_ptr; a%rloc_l = c_null_ptr   ! (the lists stay inside the library)
    a%ensval = d_ens; a%kld = obs%nensobs; a%dep = d_val; a%beta = d_beta; a%infl = d_infl
    a%gues = d_gues; a%anal = d_anal; a%sp = 1; a%sm = sm; a%sv = sv
    a%trans_out = c_null_ptr; a%transm_out = c_null_ptr; a%pa_out = c_null_ptr
    a%status = d_status; a%nsweep = c_null_ptr; a%rtps_infl_out = c_null_ptr; a%warm_run = 0; a%infl_sv = 0
    DO icl = 1, nclass
      mask = 0
      DO n = 1, nv3d
        IF (n2nc(n) == icl - 1) mask = IOR(mask, ISHFT(1_c_int32_t, n - 1))
      END DO
      ! var_local(first variable of the class, uid_obs_varlocal(elm)) per combined type (:1840)
      DO ic = 1, obs%nctype
        DO n = 1, nv3d
          IF (n2nc(n) == icl - 1) EXIT
        END DO
        varloc(ic) = nml%var_local(n, obs%uid_varlocal_ctype(ic))
      END DO
      CALL chk(hipMemcpy(d_varloc, c_loc(varloc), 8_c_size_t*MAX(obs%nctype, 1), hipMemcpyHostToDevice), 'upload varloc')
      ! obs_local for every point + the loop body, by slabs of levels whose lists fit the library's workspace: ONE call
      ! (letkf_das_columns_dev).  The pressure of a point is the ensemble mean of iv3d_p (:420).
      a%var_mask = MERGE(0_c_int32_t, mask, nclass == 1)
      CALL chk(letkf_das_columns_dev(ctx, a, t, INT(nij1, c_int64_t), INT(nlev, c_int32_t), d_rig, d_rjg, &
                                     off_ptr(d_gues, (mmean - 1)*sm + (nml%iv3d_p - 1)*sv), d_hgt, nml%list_bytes, &
                                     MERGE(d_counts, c_null_ptr, icl == 1)), 'letkf_das_columns_dev')
      CALL chk(letkf_ctx_synchronize(ctx), 'synchronize')
      CALL chk(hipMemcpy(c_loc(status), d_status, 4_c_size_t*npts, hipMemcpyDeviceToHost), 'download status')
      IF (ANY(status /= 0)) THEN                    ! the reference's behaviour: print and STOP 2 (common_mtx.f90:61-64)
        WRITE (6, '(A,I10,A)') '!!! ERROR (mtx_eigen): letkf_das_points_dev reports a non-zero status at', COUNT(status /= 0), ' points'
        STOP 2
      END IF
      IF (icl == 1 .AND. PRESENT(nobs_point)) THEN
        CALL chk(hipMemcpy(c_loc(counts), d_counts, 4_c_size_t*npts, hipMemcpyDeviceToHost), 'download counts')
        nobs_point = RESHAPE(counts, (/nij1, nlev/))
      END IF
    END DO

    ! ---- ensmean_grd on the analysis (letkf.f90:207), then everything back to the caller's arrays
    CALL chk(letkf_ens_mean_dev(ctx, INT(k, c_int32_t), INT(nv3d, c_int32_t), sm, d_anal, 1_c_int64_t, sm, sv), 'letkf_ens_mean_dev')
    CALL chk(letkf_ctx_synchronize(ctx), 'synchronize')
    CALL chk(hipMemcpy(c_loc(anal3d), d_anal, nb_state, hipMemcpyDeviceToHost), 'download anal3d')
    CALL chk(hipMemcpy(c_loc(gues3d), d_gues, nb_state, hipMemcpyDeviceToHost), 'download gues3d')
    IF (PRESENT(work3d_out)) CALL chk(hipMemcpy(c_loc(work3d_out), d_infl, 8_c_size_t*npts*nv3d, hipMemcpyDeviceToHost), &
                                      'download work3d')
    CALL free_all((/d_gues, d_anal, d_rig, d_rjg, d_hgt, d_beta, d_infl, d_ens, d_val, d_counts, d_status, d_varloc, &
                    d_gs, d_gm, d_vm, d_hl, d_vl, d_mx, d_gi, d_gj, d_si, d_sj, d_ei, d_ej, d_aco, d_ace, d_ri, d_rj, d_lev, &
                    d_dat, d_err/))
  END SUBROUTINE das_letkf_amd

  ! device pointer + an offset in doubles
  FUNCTION off_ptr(base, ndbl) RESULT(p)
    TYPE(c_ptr), INTENT(IN) :: base
    INTEGER(c_int64_t), INTENT(IN) :: ndbl
    TYPE(c_ptr) :: p
    p = TRANSFER(TRANSFER(base, 0_c_intptr_t) + 8_c_intptr_t*ndbl, p)
  END FUNCTION off_ptr

  FUNCTION up(host, nbytes) RESULT(d)
    TYPE(c_ptr), INTENT(IN) :: host
    INTEGER(c_size_t), INTENT(IN) :: nbytes
    TYPE(c_ptr) :: d
    CALL chk(hipMalloc(d, MAX(nbytes, 8_c_size_t)), 'hipMalloc')
    IF (nbytes > 0) CALL chk(hipMemcpy(d, host, nbytes, hipMemcpyHostToDevice), 'hipMemcpy H2D')
  END FUNCTION up

  SUBROUTINE free_all(ptrs)
    TYPE(c_ptr), INTENT(IN) :: ptrs(:)
    INTEGER :: i
    INTEGER(c_int) :: rc
    DO i = 1, SIZE(ptrs)
      rc = hipFree(ptrs(i))
    END DO
  END SUBROUTINE free_all

  SUBROUTINE chk(rc, what)
    INTEGER(c_int), INTENT(IN) :: rc
    CHARACTER(*), INTENT(IN) :: what
    IF (rc /= 0) THEN
      WRITE (6, '(A,I6,2A)') 'das_letkf_amd: error', rc, ' in ', what
      STOP 5
    END IF
  END SUBROUTINE chk

  SUBROUTINE fail(what)
    CHARACTER(*), INTENT(IN) :: what
    WRITE (6, '(2A)') 'das_letkf_amd: ', what
    STOP 6
  END SUBROUTINE fail

END MODULE letkf_tools_amd
