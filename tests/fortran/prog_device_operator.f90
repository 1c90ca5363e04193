!> The caller's own HIP kernel as operator, from Fortran: engine_set_device_operator with c_funloc of a bind(C) function that
!> lives in the caller's own library (tests/helpers/user_operator.hip -> lib/test/libuser_operator.so: a banded stencil), then
!> the generic generalized_eigensolver on the engine - against the drop-in dense call on the same matrix: same eigenvalues,
!> same iteration counts, residuals below the tolerance; davidson_free_buffers at the end.
program prog_device_operator
  use iso_c_binding
  use numeric_kinds, only: dp
  use davidson, only: generalized_eigensolver, davidson_free_buffers
  use davidson_device
  use array_utils, only: norm
  implicit none
  interface
     function user_op_create(d0, dstep, eps) bind(C, name="user_op_create") result(ctx)
       import :: c_double, c_ptr
       real(c_double), value :: d0, dstep, eps
       type(c_ptr) :: ctx
     end function user_op_create
     function user_op_apply(ctx, stream, n, row0, nloc, k, x, ldx, y, ldy) bind(C, name="user_op_apply") result(rc)
       import :: c_ptr, c_int64_t, c_int
       type(c_ptr), value :: ctx, stream, x, y
       integer(c_int64_t), value :: n, row0, nloc, ldx, ldy
       integer(c_int), value :: k
       integer(c_int) :: rc
     end function user_op_apply
  end interface
  integer, parameter :: dim = 2000, lowest = 4
  real(dp), allocatable :: mtx(:, :), diag(:)
  real(dp) :: ev0(lowest), x0(dim, lowest), ev(lowest), x(dim, lowest), r(dim)
  type(davidson_engine) :: eng
  integer :: it0, it, i, j, nfail, m
  character(len=3) :: methods(2) = ["DPR", "GJD"]

  nfail = 0
  allocate(mtx(dim, dim), diag(dim))
  mtx = 0.0_dp
  do i = 1, dim
     mtx(i, i) = real(i, dp)
     diag(i) = real(i, dp)
     if (i + 1 <= dim) then
        mtx(i, i + 1) = 0.3_dp
        mtx(i + 1, i) = 0.3_dp
     end if
     if (i + 2 <= dim) then
        mtx(i, i + 2) = 0.15_dp
        mtx(i + 2, i) = 0.15_dp
     end if
  end do
  call engine_create(eng, dim, lowest, 10 * lowest, gev=.false.)
  call engine_set_device_operator(eng, 1, c_funloc(user_op_apply), user_op_create(1.0_dp, 1.0_dp, 0.3_dp), diag)
  do m = 1, 2
     call generalized_eigensolver(mtx, ev0, x0, lowest, methods(m), 200, 1d-8, it0)      ! the drop-in dense call
     call generalized_eigensolver(eng, ev, x, lowest, methods(m), 200, 1d-8, it, 10 * lowest)
     call check(methods(m) // "_eigenvalues", maxval(abs(ev - ev0)) < 1d-10)
     call check(methods(m) // "_iterations", it == it0)
     do j = 1, lowest
        r = matmul(mtx, x(:, j)) - ev(j) * x(:, j)
        call check(methods(m) // "_residual", norm(r) < 1d-8)
     end do
  end do
  call engine_destroy(eng)
  call davidson_free_buffers()
  print "(a, 2i4)", "ITERS", it0, it
  if (nfail > 0) error stop 2

contains
  subroutine check(name, ok)
    character(len=*), intent(in) :: name
    logical, intent(in) :: ok
    print "(a, a, 1x, l1)", "CHECK ", name, ok
    if (.not. ok) nfail = nfail + 1
  end subroutine check
end program prog_device_operator
