!> The opt-in engine knobs from Fortran: symmetric-tiled storage, fp32 inner sweeps of the GJD correction
!> (engine_set_inner_precision), device-side Rayleigh-Ritz (engine_set_device_rr) - each against the answer of the drop-in
!> call with the reference's signature on the same matrices: same eigenvalues, same iteration counts, residuals below
!> the tolerance.
program prog_options
  use numeric_kinds, only: dp
  use davidson, only: generalized_eigensolver
  use davidson_device
  use array_utils, only: generate_diagonal_dominant, norm
  implicit none
  integer, parameter :: dim = 600, lowest = 3, max_dim = 12
  real(dp), allocatable :: mtx(:, :), stx(:, :)
  real(dp) :: ev0(lowest), x0(dim, lowest), evd(lowest), xd(dim, lowest), ev(lowest), x(dim, lowest), r(dim)
  type(davidson_engine) :: eng
  integer :: it0, itd, it, j, nfail

  nfail = 0
  allocate(mtx(dim, dim), stx(dim, dim))
  mtx = generate_diagonal_dominant(dim, 1d-2, seed=1)
  stx = generate_diagonal_dominant(dim, 1d-2, 1d0, 2)
  ! the drop-in calls (src/davidson.f90:51-52): full storage, host Rayleigh-Ritz, fp64 throughout
  call generalized_eigensolver(mtx, ev0, x0, lowest, "GJD", 100, 1d-8, it0, max_dim, stx)
  call generalized_eigensolver(mtx, evd, xd, lowest, "DPR", 200, 1d-8, itd, max_dim, stx)

  call engine_create(eng, dim, lowest, max_dim, gev=.true.)
  call engine_set_storage(eng, "symmetric")
  call engine_set_dense(eng, 1, mtx)
  call engine_set_dense(eng, 2, stx)

  call generalized_eigensolver(eng, ev, x, lowest, "GJD", 100, 1d-8, it, max_dim)
  call check("symmetric_storage_gjd_eigenvalues", maxval(abs(ev - ev0)) < 1d-10)
  call check("symmetric_storage_gjd_iterations", it == it0)

  call engine_set_inner_precision(eng, 32)
  call generalized_eigensolver(eng, ev, x, lowest, "GJD", 100, 1d-8, it, max_dim)
  call check("fp32_inner_sweeps_eigenvalues", maxval(abs(ev - ev0)) < 1d-9)
  call check("fp32_inner_sweeps_iterations", it == it0)
  do j = 1, lowest
     r = matmul(mtx, x(:, j)) - ev(j) * matmul(stx, x(:, j))
     call check("fp32_inner_sweeps_residual", norm(r) < 1d-8)
  end do
  call engine_set_inner_precision(eng, 64)

  call engine_set_device_rr(eng, .true.)
  call generalized_eigensolver(eng, ev, x, lowest, "DPR", 200, 1d-8, it, max_dim)
  call check("device_rr_dpr_eigenvalues", maxval(abs(ev - evd)) < 1d-10)
  call check("device_rr_dpr_iterations", it == itd)
  call generalized_eigensolver(eng, ev, x, lowest, "GJD", 100, 1d-8, it, max_dim)
  call check("device_rr_gjd_eigenvalues", maxval(abs(ev - ev0)) < 1d-10)
  call check("device_rr_gjd_iterations", it == it0)
  do j = 1, lowest
     r = matmul(mtx, x(:, j)) - ev(j) * matmul(stx, x(:, j))
     call check("device_rr_residual", norm(r) < 1d-8)
  end do
  call engine_destroy(eng)
  print "(a, 3i4)", "ITERS", it0, itd, it
  if (nfail > 0) error stop 2

contains
  subroutine check(name, ok)
    character(len=*), intent(in) :: name
    logical, intent(in) :: ok
    print "(a, a, 1x, l1)", "CHECK ", name, ok
    if (.not. ok) nfail = nfail + 1
  end subroutine check
end program prog_options
