!> What a Fortran user of the reference's API experiences, timed in a FRESH process: host matrix in,
!> `call generalized_eigensolver(matrix, eigenvalues, eigenvectors, lowest, method, max_iterations, tolerance, iters)`
!> (src/davidson.f90:51-52) three times.  The first call carries what a process pays once (HIP runtime and code objects loaded,
!> first allocations), the others are the steady state of the drop-in call: engine created, matrix uploaded over PCIe, solved,
!> eigenvectors downloaded, everything released.  bench.py runs it as a child and puts the three figures in its `dropin` object.
!>   dropin_timing [n [lowest [asymmetric]]]      asymmetric = 1: one entry of the matrix is made asymmetric (probe fails: full upload)
program dropin_timing
  use, intrinsic :: iso_fortran_env, only: int64
  use numeric_kinds, only: dp
  use davidson, only: generalized_eigensolver
  use array_utils, only: generate_diagonal_dominant
  implicit none
  integer :: n, lowest, iters, k, asym
  character(len=32) :: arg
  real(dp), allocatable :: mtx(:, :), ev(:), x(:, :)
  real(dp) :: t0, t1, secs(3)
  integer(int64) :: cnt, rate

  n = 20000; lowest = 8; asym = 0
  if (command_argument_count() >= 1) then
     call get_command_argument(1, arg); read (arg, *) n
  end if
  if (command_argument_count() >= 2) then
     call get_command_argument(2, arg); read (arg, *) lowest
  end if
  if (command_argument_count() >= 3) then
     call get_command_argument(3, arg); read (arg, *) asym
  end if
  allocate(mtx(n, n), ev(lowest), x(n, lowest))
  mtx = generate_diagonal_dominant(n, 1.0e-3_dp)
  if (asym == 1) mtx(n, 1) = mtx(n, 1) + 1.0e-9_dp
  do k = 1, 3
     call system_clock(cnt, rate); t0 = real(cnt, dp) / real(rate, dp)
     call generalized_eigensolver(mtx, ev, x, lowest, "DPR", 1000, 1.0e-8_dp, iters)
     call system_clock(cnt, rate); t1 = real(cnt, dp) / real(rate, dp)
     secs(k) = t1 - t0
  end do
  print "(a, i0, a, i0, a, i0, a, 3(f10.5, 1x), a, 3(es23.16, 1x))", "DROPIN_TIMING n=", n, " lowest=", lowest, " iters=", iters, &
       " seconds=", secs, " eigenvalues=", ev(1:min(3, lowest))
end program dropin_timing
