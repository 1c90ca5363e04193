!> Drop-in check, dense path: what a user program of the reference looks like (config 1: N=50,
!> lowest=3, GJD and DPR, standard and generalized), linked against OUR modules.  Prints one
!> "CHECK name T|F" line per property and stops with a non-zero code on any F.
program prog_dense
  use numeric_kinds, only: dp
  use davidson, only: generalized_eigensolver
  use array_utils, only: generate_diagonal_dominant, norm, diagonal
  implicit none
  integer, parameter :: dim = 50, lowest = 3
  real(dp) :: mtx(dim, dim), stx(dim, dim)
  real(dp) :: ev_dpr(lowest), ev_gjd(lowest), ev_gen(lowest), ev_gen_gjd(lowest)
  real(dp) :: x_dpr(dim, lowest), x_gjd(dim, lowest), x_gen(dim, lowest), x_gen_gjd(dim, lowest)
  real(dp) :: r(dim)
  real(dp) :: big(dim + 7, dim + 3), ev_sec(lowest), x_sec(dim + 2, lowest)
  integer :: it_sec
  integer :: it_dpr, it_gjd, it_gen, it_gen_gjd, j, nfail

  nfail = 0
  mtx = generate_diagonal_dominant(dim, 1d-3)
  stx = generate_diagonal_dominant(dim, 1d-3, 1d0, 2)   ! seed 2 = the golden generalized case

  call generalized_eigensolver(mtx, ev_gjd, x_gjd, lowest, "GJD", 1000, 1d-8, it_gjd)
  call generalized_eigensolver(mtx, ev_dpr, x_dpr, lowest, "DPR", 1000, 1d-8, it_dpr)
  call generalized_eigensolver(mtx, ev_gen_gjd, x_gen_gjd, lowest, "GJD", 1000, 1d-8, it_gen_gjd, 10, stx)
  call generalized_eigensolver(mtx, ev_gen, x_gen, lowest, "DPR", 1000, 1d-8, it_gen, 10, stx)

  call check("gjd_equals_dpr", norm(ev_gjd - ev_dpr) < 1d-8)
  call check("gen_gjd_equals_dpr", norm(ev_gen_gjd - ev_gen) < 1d-8)
  do j = 1, lowest
     r = matmul(mtx, x_dpr(:, j)) - ev_dpr(j) * x_dpr(:, j)
     call check("residual_dpr", norm(r) < 1d-8)
     r = matmul(mtx, x_gjd(:, j)) - ev_gjd(j) * x_gjd(:, j)
     call check("residual_gjd", norm(r) < 1d-8)
     r = matmul(mtx, x_gen(:, j)) - ev_gen(j) * matmul(stx, x_gen(:, j))
     call check("residual_gen_dpr", norm(r) < 1d-8)
     r = matmul(mtx, x_gen_gjd(:, j)) - ev_gen_gjd(j) * matmul(stx, x_gen_gjd(:, j))
     call check("residual_gen_gjd", norm(r) < 1d-8)
     call check("unit_norm", abs(norm(x_dpr(:, j)) - 1d0) < 1d-10)
  end do
  call check("ascending", ev_dpr(1) < ev_dpr(2) .and. ev_dpr(2) < ev_dpr(3))

  ! non-contiguous array sections and keyword arguments, as any assumed-shape API must accept
  big = -1.0_dp
  big(4:dim + 3, 2:dim + 1) = mtx
  x_sec = 0.0_dp
  call generalized_eigensolver(big(4:dim + 3, 2:dim + 1), ev_sec, x_sec(2:dim + 1, :), lowest, "DPR", &
       max_iterations=1000, tolerance=1d-8, iters=it_sec, max_dim_sub=12)
  call check("section_input_same_eigenvalues", norm(ev_sec - ev_dpr) < 1d-8)
  call check("section_output_untouched_border", all(x_sec(1, :) == 0.0_dp) .and. all(x_sec(dim + 2, :) == 0.0_dp))
  r = matmul(mtx, x_sec(2:dim + 1, 1)) - ev_sec(1) * x_sec(2:dim + 1, 1)
  call check("section_residual", norm(r) < 1d-8)
  print "(a, 4i4)", "ITERS", it_dpr, it_gjd, it_gen, it_gen_gjd
  print "(a, 3es24.16)", "EVALS_DPR", ev_dpr
  print "(a, 3es24.16)", "EVALS_GEN", ev_gen
  if (nfail > 0) error stop 2

contains
  subroutine check(name, ok)
    character(len=*), intent(in) :: name
    logical, intent(in) :: ok
    print "(a, 1x, a, 1x, l1)", "CHECK", name, ok
    if (.not. ok) nfail = nfail + 1
  end subroutine check
end program prog_dense
