!===============================================================================
! shim_driver.f90 -- exercises the Fortran -> C ABI -> HIP path exactly as an
! unmodified caller of letkf_core would (keyword OPTIONALs, hdxb with leading
! dimension nobs > nobsl, the call forms of scale/letkf/letkf_tools.f90:418-436).
! Reads one problem from a raw little-endian file, writes the outputs; driven by
! tests/test_fortran_shim.py, which compares them with the oracle.
!   file layout: int32 ne, nobs, nobsl, flags ; real64 parm_infl ;
!                hdxb(nobs,ne), rdiag(nobs), rloc(nobs), dep(nobs), depd(nobs)
!   flags: bit0 infl_update, bit1 DET_RUN form (depd/transmd), bit2 RTPS form (pao)
!===============================================================================
PROGRAM shim_driver
  USE common_letkf
  IMPLICIT NONE
  INTEGER :: ne, nobs, nobsl, flags, u, ios
  REAL(r_size) :: parm_infl
  REAL(r_size), ALLOCATABLE :: hdxb(:, :), rdiag(:), rloc(:), dep(:), depd(:)
  REAL(r_size), ALLOCATABLE :: trans(:, :), transm(:), pao(:, :), transmd(:)
  CHARACTER(len=512) :: fin, fout
  LOGICAL :: iupd

  CALL get_command_argument(1, fin)
  CALL get_command_argument(2, fout)
  OPEN (newunit=u, file=trim(fin), access='stream', form='unformatted', status='old', iostat=ios)
  IF (ios /= 0) STOP 3
  READ (u) ne, nobs, nobsl, flags
  READ (u) parm_infl
  ALLOCATE (hdxb(nobs, ne), rdiag(nobs), rloc(nobs), dep(nobs), depd(nobs))
  ALLOCATE (trans(ne, ne), transm(ne), pao(ne, ne), transmd(ne))
  READ (u) hdxb, rdiag, rloc, dep, depd
  CLOSE (u)
  iupd = btest(flags, 0)
  pao = 0.0d0
  transmd = 0.0d0

  IF (btest(flags, 2)) THEN
    IF (btest(flags, 1)) THEN      ! scale/letkf/letkf_tools.f90:418-421
      CALL letkf_core(ne, nobs, nobsl, hdxb, rdiag, rloc, dep, parm_infl, trans, transm=transm, pao=pao, &
                      rdiag_wloc=.true., infl_update=iupd, depd=depd, transmd=transmd)
    ELSE                           ! :423-425
      CALL letkf_core(ne, nobs, nobsl, hdxb, rdiag, rloc, dep, parm_infl, trans, transm=transm, pao=pao, &
                      rdiag_wloc=.true., infl_update=iupd)
    END IF
  ELSE
    IF (btest(flags, 1)) THEN      ! :429-432
      CALL letkf_core(ne, nobs, nobsl, hdxb, rdiag, rloc, dep, parm_infl, trans, transm=transm, &
                      rdiag_wloc=.true., infl_update=iupd, depd=depd, transmd=transmd)
    ELSE                           ! :434-436
      CALL letkf_core(ne, nobs, nobsl, hdxb, rdiag, rloc, dep, parm_infl, trans, transm=transm, &
                      rdiag_wloc=.true., infl_update=iupd)
    END IF
  END IF

  OPEN (newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
  WRITE (u) parm_infl
  WRITE (u) trans, transm, pao, transmd
  CLOSE (u)
END PROGRAM shim_driver
