!> Kind parameters of the drop-in API (same public names as the reference's numeric_kinds,
!> src/numeric_kinds.f90:3-11: everything on the solver path is `real(dp)` = IEEE binary64).
module numeric_kinds
  use, intrinsic :: iso_fortran_env, only: real32, real64, real128, int32, int16, int8
  implicit none
  private
  public :: sp, dp, qp, i4b, i2b, i1b
  integer, parameter :: sp = real32
  integer, parameter :: dp = real64
  integer, parameter :: qp = real128
  integer, parameter :: i4b = int32
  integer, parameter :: i2b = int16
  integer, parameter :: i1b = int8
end module numeric_kinds
