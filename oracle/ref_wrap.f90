!===============================================================================
! oracle/ref_wrap.f90 -- TEST INFRASTRUCTURE, NOT PRODUCT.
!
! bind(C) entry points that CALL the reference's own routines so that Python
! (ctypes) can run them as the parity oracle and as the "reference" CPU
! baseline.  Nothing here restates or replaces reference code: the reference
! modules (common, common_mtx, common_letkf, common_sort and the EISPACK /
! netlib BLAS files) are compiled from /root/reference where they lie by
! oracle/Makefile, and the result only ever lands in oracle/_ref/.
!
! Callee interfaces:
!   letkf_core        /root/reference/common/common_letkf.f90:52
!   mtx_eigen         /root/reference/common/common_mtx.f90:41
!   QUICKSELECT_arg   /root/reference/common/common_sort.f90:341
!===============================================================================
module ref_wrap
  use iso_c_binding
  use common, only: r_size
  use common_mtx, only: mtx_eigen
  use common_letkf, only: letkf_core
  use common_sort, only: QUICKSELECT_arg, QUICKSELECT_desc_arg
  implicit none
contains

  ! flags: bit0 transm present, bit1 pao present, bit2 rdiag_wloc present,
  !        bit3 rdiag_wloc value, bit4 infl_update present, bit5 infl_update value,
  !        bit6 depd present, bit7 transmd present
  subroutine ref_letkf_core(ne, nobs, nobsl, hdxb, rdiag, rloc, dep, parm_infl, &
                            trans, transm, pao, depd, transmd, flags) bind(C, name="ref_letkf_core")
    integer(c_int), value :: ne, nobs, nobsl, flags
    real(c_double), intent(in) :: hdxb(nobs, ne), rdiag(nobs), rloc(nobs), dep(nobs), depd(nobs)
    real(c_double), intent(inout) :: parm_infl
    real(c_double), intent(out) :: trans(ne, ne), transm(ne), pao(ne, ne), transmd(ne)
    logical :: has_tm, has_pa, has_wl, wl, has_iu, iu, has_dd, has_td

    has_tm = btest(flags, 0); has_pa = btest(flags, 1)
    has_wl = btest(flags, 2); wl = btest(flags, 3)
    has_iu = btest(flags, 4); iu = btest(flags, 5)
    has_dd = btest(flags, 6); has_td = btest(flags, 7)

    ! Fortran OPTIONAL arguments cannot be forwarded conditionally from
    ! non-optional dummies, so enumerate the combinations the tests use.
    if (.not. has_wl) wl = .false.
    if (.not. has_iu) iu = .false.
    if (has_tm .and. has_pa .and. has_dd .and. has_td) then
      call letkf_core(ne, nobs, nobsl, hdxb, rdiag, rloc, dep, parm_infl, trans, transm=transm, pao=pao, &
                      rdiag_wloc=wl, infl_update=iu, depd=depd, transmd=transmd)
    else if (has_tm .and. has_pa) then
      call letkf_core(ne, nobs, nobsl, hdxb, rdiag, rloc, dep, parm_infl, trans, transm=transm, pao=pao, &
                      rdiag_wloc=wl, infl_update=iu)
    else if (has_tm .and. has_dd .and. has_td) then
      call letkf_core(ne, nobs, nobsl, hdxb, rdiag, rloc, dep, parm_infl, trans, transm=transm, &
                      rdiag_wloc=wl, infl_update=iu, depd=depd, transmd=transmd)
    else if (has_tm) then
      call letkf_core(ne, nobs, nobsl, hdxb, rdiag, rloc, dep, parm_infl, trans, transm=transm, &
                      rdiag_wloc=wl, infl_update=iu)
    else if (has_pa) then
      call letkf_core(ne, nobs, nobsl, hdxb, rdiag, rloc, dep, parm_infl, trans, pao=pao, &
                      rdiag_wloc=wl, infl_update=iu)
    else if (has_wl .or. has_iu) then
      call letkf_core(ne, nobs, nobsl, hdxb, rdiag, rloc, dep, parm_infl, trans, &
                      rdiag_wloc=wl, infl_update=iu)
    else
      call letkf_core(ne, nobs, nobsl, hdxb, rdiag, rloc, dep, parm_infl, trans)
    end if
  end subroutine ref_letkf_core

  subroutine ref_mtx_eigen(n, a, eival, eivec, nrank_eff) bind(C, name="ref_mtx_eigen")
    integer(c_int), value :: n
    real(c_double), intent(in) :: a(n, n)
    real(c_double), intent(out) :: eival(n), eivec(n, n)
    integer(c_int), intent(out) :: nrank_eff
    integer :: nr
    call mtx_eigen(1, n, a, eival, eivec, nr)
    nrank_eff = nr
  end subroutine ref_mtx_eigen

  ! In-place K-select on an index array (1-based indices into a), ascending
  ! (desc=0) or descending (desc/=0) keys.
  subroutine ref_quickselect_arg(na, a, nx, x, left, right, k, desc) bind(C, name="ref_quickselect_arg")
    integer(c_int), value :: na, nx, left, right, k, desc
    real(c_double), intent(in) :: a(na)
    integer(c_int), intent(inout) :: x(nx)
    integer :: xx(nx)
    xx = x
    if (desc /= 0) then
      call QUICKSELECT_desc_arg(a, xx, left, right, k)
    else
      call QUICKSELECT_arg(a, xx, left, right, k)
    end if
    x = xx
  end subroutine ref_quickselect_arg

  ! Timed loop for the CPU baseline: nrep calls of the reference letkf_core
  ! on nprob independent problems laid out back to back (same shapes).
  subroutine ref_letkf_core_loop(nprob, ne, nobs, nobsl, hdxb, rdiag, rloc, dep, infl, &
                                 trans, transm, pao) bind(C, name="ref_letkf_core_loop")
    integer(c_int), value :: nprob, ne, nobs, nobsl
    real(c_double), intent(in) :: hdxb(nobs, ne, nprob), rdiag(nobs, nprob), rloc(nobs, nprob), dep(nobs, nprob)
    real(c_double), intent(inout) :: infl(nprob)
    real(c_double), intent(out) :: trans(ne, ne, nprob), transm(ne, nprob), pao(ne, ne, nprob)
    integer :: ip
    do ip = 1, nprob
      call letkf_core(ne, nobs, nobsl, hdxb(:, :, ip), rdiag(:, ip), rloc(:, ip), dep(:, ip), infl(ip), &
                      trans(:, :, ip), transm=transm(:, ip), pao=pao(:, :, ip), rdiag_wloc=.true.)
    end do
  end subroutine ref_letkf_core_loop

end module ref_wrap
