!===============================================================================
! common_letkf_amd.f90 -- drop-in replacement for the reference's
!   MODULE common_letkf  (common/common_letkf.f90)
! that keeps the letkf_core call signature (common/common_letkf.f90:52-68:
! explicit interface, OPTIONAL arguments passed by keyword at the 8 call sites
! in scale/letkf/letkf_tools.f90:418-436,573-591) and forwards to the
! MI355X-native C ABI  letkf_core_c  (include/letkf_amd.h) through
! ISO_C_BINDING.  An unmodified letkf_tools.f90 links against this module:
! the module name, the procedure name, argument names, kinds and INTENTs are
! the reference's.
!
! Differences to the reference, all deliberate (SURVEY.md section 8(b)):
!   * errors: the reference prints and STOPs inside mtx_eigen
!     (common/common_mtx.f90:61-64,75-78).  The C ABI returns a status; this
!     shim turns any non-zero status that the reference would have stopped on
!     into the same WRITE + STOP 2.  LETKF_ST_ILLCOND (3) is the case where the
!     reference silently zeroes modes and then divides by zero: it is reported
!     on unit 6 and the (finite, untruncated) result is returned.
!   * r_size: the path is IEEE double only (-DSINGLE is never set by any arch
!     file and cannot work in the reference either: dgemm/rs are double).
!
! Also provides the batched entry  letkf_core_batch  for a restructured driver
! loop (search phase -> one batched call per level slab) on host arrays.
!===============================================================================
MODULE common_letkf
  USE, INTRINSIC :: iso_c_binding
  IMPLICIT NONE
  PUBLIC

  INTEGER, PARAMETER :: r_size = c_double
  INTEGER, PARAMETER :: nbv = 20    ! kept for source compatibility (common/common_letkf.f90:28)

  INTERFACE
    SUBROUTINE letkf_core_c(ne, nobs, nobsl, hdxb, rdiag, rloc, dep, parm_infl, trans, transm, pao, &
                            rdiag_wloc, infl_update, depd, transmd, status) BIND(C, NAME='letkf_core_c')
      IMPORT :: c_int, c_ptr
      INTEGER(c_int), VALUE :: ne, nobs, nobsl
      TYPE(c_ptr), VALUE :: hdxb, rdiag, rloc, dep, parm_infl, trans   ! const double* / double*
      TYPE(c_ptr), VALUE :: transm, pao, rdiag_wloc, infl_update, depd, transmd, status   ! NULL == absent
    END SUBROUTINE letkf_core_c
  END INTERFACE

CONTAINS

SUBROUTINE letkf_core(ne, nobs, nobsl, hdxb, rdiag, rloc, dep, parm_infl, trans, transm, pao, &
                      rdiag_wloc, infl_update, depd, transmd)
  INTEGER, INTENT(IN) :: ne
  INTEGER, INTENT(IN) :: nobs
  INTEGER, INTENT(IN) :: nobsl
  REAL(r_size), INTENT(IN), TARGET :: hdxb(1:nobs, 1:ne)
  REAL(r_size), INTENT(IN), TARGET :: rdiag(1:nobs)
  REAL(r_size), INTENT(IN), TARGET :: rloc(1:nobs)
  REAL(r_size), INTENT(IN), TARGET :: dep(1:nobs)
  REAL(r_size), INTENT(INOUT), TARGET :: parm_infl
  REAL(r_size), INTENT(OUT), TARGET :: trans(ne, ne)
  REAL(r_size), INTENT(OUT), OPTIONAL, TARGET :: transm(ne)
  REAL(r_size), INTENT(OUT), OPTIONAL, TARGET :: pao(ne, ne)
  LOGICAL, INTENT(IN), OPTIONAL :: rdiag_wloc
  LOGICAL, INTENT(IN), OPTIONAL :: infl_update
  REAL(r_size), INTENT(IN), OPTIONAL, TARGET :: depd(1:nobs)
  REAL(r_size), INTENT(OUT), OPTIONAL, TARGET :: transmd(ne)

  INTEGER(c_int), TARGET :: wloc_c, iupd_c, status
  TYPE(c_ptr) :: p_transm, p_pao, p_wloc, p_iupd, p_depd, p_transmd

  p_transm = c_null_ptr; p_pao = c_null_ptr; p_wloc = c_null_ptr
  p_iupd = c_null_ptr; p_depd = c_null_ptr; p_transmd = c_null_ptr
  IF (PRESENT(transm)) p_transm = c_loc(transm)
  IF (PRESENT(pao)) p_pao = c_loc(pao)
  IF (PRESENT(rdiag_wloc)) THEN
    wloc_c = MERGE(1_c_int, 0_c_int, rdiag_wloc)
    p_wloc = c_loc(wloc_c)
  END IF
  IF (PRESENT(infl_update)) THEN
    iupd_c = MERGE(1_c_int, 0_c_int, infl_update)
    p_iupd = c_loc(iupd_c)
  END IF
  IF (PRESENT(depd)) p_depd = c_loc(depd)
  IF (PRESENT(transmd)) p_transmd = c_loc(transmd)

  status = 0
  CALL letkf_core_c(INT(ne, c_int), INT(nobs, c_int), INT(nobsl, c_int), c_loc(hdxb), c_loc(rdiag), c_loc(rloc), &
                    c_loc(dep), c_loc(parm_infl), c_loc(trans), p_transm, p_pao, p_wloc, p_iupd, p_depd, &
                    p_transmd, c_loc(status))

  SELECT CASE (status)
  CASE (0)
    CONTINUE
  CASE (1)   ! common/common_mtx.f90:61-64
    WRITE (6, '(A,I4)') '!!! ERROR (mtx_eigen): rs error code is ', status
    STOP 2
  CASE (2)   ! common/common_mtx.f90:75-78
    WRITE (6, '(A)') '!!! ERROR (mtx_eigen): All Eigenvalues are below 0'
    STOP 2
  CASE (3)
    WRITE (6, '(A)') '!!! WARNING (letkf_core): lambda_max/lambda_min > 1/sqrt(eps); the reference would truncate modes here'
  CASE DEFAULT
    WRITE (6, '(A,I6)') '!!! ERROR (letkf_core): libletkf_amd host error ', status
    STOP 2
  END SELECT
END SUBROUTINE letkf_core

END MODULE common_letkf
