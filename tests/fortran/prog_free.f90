!> Drop-in check, matrix-free path with Fortran callbacks (N=50, lowest=3, max_dim 20, tol 1e-8),
!> and the device-resident engine with the same operators evaluated on the GPU.
program prog_free
  use numeric_kinds, only: dp
  use davidson, only: generalized_eigensolver
  use davidson_device
  use array_utils, only: norm
  use harness_ops
  implicit none
  integer, parameter :: dim = 50, lowest = 3
  real(dp) :: ev(lowest), x(dim, lowest), ev_dev(lowest), x_dev(dim, lowest)
  real(dp) :: mtx(dim, dim), stx(dim, dim), r(dim)
  type(davidson_engine) :: eng
  integer :: iters, iters_dev, j, nfail

  nfail = 0
  do j = 1, dim
     mtx(:, j) = row_a(j, dim)
     stx(:, j) = row_b(j, dim)
  end do

  call generalized_eigensolver(apply_a, ev, x, lowest, "DPR", 1000, 1d-8, iters, 20, apply_b)
  do j = 1, lowest
     r = matmul(mtx, x(:, j)) - ev(j) * matmul(stx, x(:, j))
     call check("residual_free", norm(r) < 1d-8)
  end do

  call engine_create(eng, dim, lowest, 20, .true.)
  call engine_set_harness_operator(eng, 1)
  call engine_set_harness_operator(eng, 2)
  call generalized_eigensolver(eng, ev_dev, x_dev, lowest, "DPR", 1000, 1d-8, iters_dev, 20)
  call engine_destroy(eng)
  call check("device_operator_equals_callbacks", norm(ev - ev_dev) < 1d-8)

  print "(a, 2i4)", "ITERS", iters, iters_dev
  print "(a, 3es24.16)", "EVALS_FREE", ev
  if (nfail > 0) error stop 2

contains
  subroutine check(name, ok)
    character(len=*), intent(in) :: name
    logical, intent(in) :: ok
    print "(a, 1x, a, 1x, l1)", "CHECK", name, ok
    if (.not. ok) nfail = nfail + 1
  end subroutine check
end program prog_free
