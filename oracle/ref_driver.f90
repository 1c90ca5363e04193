!> TEST INFRASTRUCTURE ONLY - never linked into the product.
!>
!> bind(C) driver over the *unmodified reference modules* (compiled by oracle/build_ref.sh from
!> /root/reference/src where they lie).  It lets Python (ctypes) call the reference's own
!> `generalized_eigensolver` (davidson.f90:601-625) and a few of its helpers on arbitrary inputs, so
!> that (1) the numpy restatement in oracle/davidson_oracle.py can be pinned against the real
!> reference, (2) golden fixtures can be generated, (3) bench.py can time the reference's CPU+LAPACK
!> path on the GPU box's host cores (cpu_baseline.kind = "reference").
!>
!> Nothing in here restates reference code: it only *calls* it.

module ref_driver_callbacks
  use iso_c_binding
  use numeric_kinds, only: dp
  implicit none

  abstract interface
     subroutine c_apply(n, k, x, y) bind(C)
       import :: c_int, c_double
       integer(c_int), value :: n, k
       real(c_double), intent(in) :: x(n, k)
       real(c_double), intent(out) :: y(n, k)
     end subroutine c_apply
  end interface

  procedure(c_apply), pointer, save :: cb_a => null()
  procedure(c_apply), pointer, save :: cb_b => null()

contains

  function apply_a(input_vect) result(output_vect)
    real(dp), dimension(:, :), intent(in) :: input_vect
    real(dp), dimension(size(input_vect, 1), size(input_vect, 2)) :: output_vect
    real(dp), allocatable :: x(:, :)
    x = input_vect
    call cb_a(int(size(x, 1), c_int), int(size(x, 2), c_int), x, output_vect)
  end function apply_a

  function apply_b(input_vect) result(output_vect)
    real(dp), dimension(:, :), intent(in) :: input_vect
    real(dp), dimension(size(input_vect, 1), size(input_vect, 2)) :: output_vect
    real(dp), allocatable :: x(:, :)
    x = input_vect
    call cb_b(int(size(x, 1), c_int), int(size(x, 2), c_int), x, output_vect)
  end function apply_b

end module ref_driver_callbacks


!> The second operator of the reference's benchmark program (src/benchmark_free.f90:65-76, :24-34): the identity, applied the way
!> that program applies it - through the reference's free_matmul with a row function that returns e_i.
module ref_driver_benchmark
  use numeric_kinds, only: dp
  use davidson_free, only: free_matmul
  implicit none
contains
  function unit_row(i, dim) result(vector)
    integer, intent(in) :: i, dim
    real(dp), dimension(dim) :: vector
    vector = 0.0_dp
    vector(i) = 1.0_dp
  end function unit_row

  function apply_unit_rows(input_vect) result(output_vect)
    real(dp), dimension(:, :), intent(in) :: input_vect
    real(dp), dimension(size(input_vect, 1), size(input_vect, 2)) :: output_vect
    output_vect = free_matmul(unit_row, input_vect)
  end function apply_unit_rows
end module ref_driver_benchmark


!> dense solve: matrix (n,n) column-major, optional second matrix, method 0=DPR 1=GJD,
!> max_dim < 0 means "argument absent" (davidson.f90:115-119).
subroutine ref_dense_solve(n, a, has_b, b, lowest, method, max_it, tol, max_dim, evals, evecs, iters) bind(C)
  use iso_c_binding
  use numeric_kinds, only: dp
  use davidson, only: generalized_eigensolver
  implicit none
  integer(c_int), value :: n, has_b, lowest, method, max_it, max_dim
  real(c_double), value :: tol
  real(c_double), intent(in) :: a(n, n), b(n, *)
  real(c_double), intent(out) :: evals(lowest), evecs(n, lowest)
  integer(c_int), intent(out) :: iters
  character(len=3) :: meth
  integer :: it

  meth = "DPR"
  if (method == 1) meth = "GJD"
  it = -1
  if (has_b /= 0) then
     if (max_dim >= 0) then
        call generalized_eigensolver(a, evals, evecs, lowest, meth, max_it, tol, it, max_dim, b(:, 1:n))
     else
        call generalized_eigensolver(a, evals, evecs, lowest, meth, max_it, tol, it, second_matrix=b(:, 1:n))
     end if
  else
     if (max_dim >= 0) then
        call generalized_eigensolver(a, evals, evecs, lowest, meth, max_it, tol, it, max_dim)
     else
        call generalized_eigensolver(a, evals, evecs, lowest, meth, max_it, tol, it)
     end if
  end if
  iters = it
end subroutine ref_dense_solve


!> matrix-free solve with the harness operators of the reference's own tests
!> (tests/test_utils.f90: apply_mtx_to_vect / apply_stx_to_vect).
subroutine ref_free_solve_harness(n, lowest, max_it, tol, max_dim, evals, evecs, iters) bind(C)
  use iso_c_binding
  use numeric_kinds, only: dp
  use davidson, only: generalized_eigensolver
  use test_utils, only: apply_mtx_to_vect, apply_stx_to_vect
  implicit none
  integer(c_int), value :: n, lowest, max_it, max_dim
  real(c_double), value :: tol
  real(c_double), intent(out) :: evals(lowest), evecs(n, lowest)
  integer(c_int), intent(out) :: iters
  integer :: it
  it = -1
  call generalized_eigensolver(apply_mtx_to_vect, evals, evecs, lowest, "DPR", max_it, tol, it, max_dim, &
       apply_stx_to_vect)
  iters = it
end subroutine ref_free_solve_harness


!> the reference's benchmark program as a call (src/benchmark_free.f90:80-111): A = the cos row generator of the tests
!> (test_utils: apply_mtx_to_vect - the same function as that program's mtx_gemv), B = I through free_matmul, DPR.
subroutine ref_free_solve_benchmark(n, lowest, max_it, tol, max_dim, evals, evecs, iters) bind(C)
  use iso_c_binding
  use numeric_kinds, only: dp
  use davidson, only: generalized_eigensolver
  use test_utils, only: apply_mtx_to_vect
  use ref_driver_benchmark, only: apply_unit_rows
  implicit none
  integer(c_int), value :: n, lowest, max_it, max_dim
  real(c_double), value :: tol
  real(c_double), intent(out) :: evals(lowest), evecs(n, lowest)
  integer(c_int), intent(out) :: iters
  integer :: it
  it = -1
  call generalized_eigensolver(apply_mtx_to_vect, evals, evecs, lowest, "DPR", max_it, tol, it, max_dim, apply_unit_rows)
  iters = it
end subroutine ref_free_solve_benchmark


!> matrix-free solve with C callbacks y = A x, y = B x (x, y are (n,k) column-major).
subroutine ref_free_solve_cb(n, fa, fb, lowest, max_it, tol, max_dim, evals, evecs, iters) bind(C)
  use iso_c_binding
  use numeric_kinds, only: dp
  use davidson, only: generalized_eigensolver
  use ref_driver_callbacks
  implicit none
  integer(c_int), value :: n, lowest, max_it, max_dim
  type(c_funptr), value :: fa, fb
  real(c_double), value :: tol
  real(c_double), intent(out) :: evals(lowest), evecs(n, lowest)
  integer(c_int), intent(out) :: iters
  integer :: it
  call c_f_procpointer(fa, cb_a)
  call c_f_procpointer(fb, cb_b)
  it = -1
  call generalized_eigensolver(apply_a, evals, evecs, lowest, "DPR", max_it, tol, it, max_dim, apply_b)
  iters = it
end subroutine ref_free_solve_cb


!> the harness matrices as test_free_numpy.f90:18-21 builds them (column j = row function(j)).
subroutine ref_harness_matrices(n, mtx, stx) bind(C)
  use iso_c_binding
  use test_utils, only: compute_matrix_on_the_fly, compute_stx_on_the_fly
  implicit none
  integer(c_int), value :: n
  real(c_double), intent(out) :: mtx(n, n), stx(n, n)
  integer :: j
  do j = 1, n
     mtx(:, j) = compute_matrix_on_the_fly(j, n)
     stx(:, j) = compute_stx_on_the_fly(j, n)
  end do
end subroutine ref_harness_matrices


!> lapack_wrapper.f90:176-236
subroutine ref_lapack_qr(m, n, basis) bind(C)
  use iso_c_binding
  use lapack_wrapper, only: lapack_qr
  implicit none
  integer(c_int), value :: m, n
  real(c_double), intent(inout) :: basis(m, n)
  call lapack_qr(basis)
end subroutine ref_lapack_qr


!> lapack_wrapper.f90:14-91
subroutine ref_lapack_eigensolver(n, mtx, has_stx, stx, evals, evecs) bind(C)
  use iso_c_binding
  use lapack_wrapper, only: lapack_generalized_eigensolver
  implicit none
  integer(c_int), value :: n, has_stx
  real(c_double), intent(in) :: mtx(n, n), stx(n, *)
  real(c_double), intent(out) :: evals(n), evecs(n, n)
  if (has_stx /= 0) then
     call lapack_generalized_eigensolver(mtx, evals, evecs, stx(:, 1:n))
  else
     call lapack_generalized_eigensolver(mtx, evals, evecs)
  end if
end subroutine ref_lapack_eigensolver


!> array_utils.f90:136-160 (sorts a copy of diag)
subroutine ref_generate_preconditioner(n, diag, dim_sub, precond) bind(C)
  use iso_c_binding
  use numeric_kinds, only: dp
  use array_utils, only: generate_preconditioner
  implicit none
  integer(c_int), value :: n, dim_sub
  real(c_double), intent(in) :: diag(n)
  real(c_double), intent(out) :: precond(n, dim_sub)
  real(dp) :: d(n)
  d = diag
  precond = generate_preconditioner(d, dim_sub)
end subroutine ref_generate_preconditioner


!> lapack_wrapper.f90:238-277 (nrhs = 1)
subroutine ref_lapack_solver(n, arr, brr) bind(C)
  use iso_c_binding
  use lapack_wrapper, only: lapack_solver
  implicit none
  integer(c_int), value :: n
  real(c_double), intent(inout) :: arr(n, n), brr(n, 1)
  call lapack_solver(arr, brr)
end subroutine ref_lapack_solver


!> Input for the CPU baseline: OUR counter-based generate_diagonal_dominant (semantics of array_utils.f90:86-113; the splitmix64
!> stream of oracle/davidson_oracle.py: uniform01 - bit-identical to the device generator) written into a caller-provided,
!> still untouched n x n array under OpenMP, column blocks per thread: every page is first touched by a thread of the team that
!> later reads it, so the matrix is spread over the NUMA nodes of the host the way a parallel producer leaves it (a single-threaded
!> copy would put all 3.2 GB of the N=20000 matrix on one node and halve what MKL's DGEMV gets).  Not reference code.
subroutine ref_generate(n, sparsity, seed, has_diag, diag_val, a) bind(C)
  use iso_c_binding
  use iso_fortran_env, only: int64
  implicit none
  integer(c_int), value :: n, seed, has_diag
  real(c_double), value :: sparsity, diag_val
  real(c_double), intent(out) :: a(n, n)
  integer(int64) :: golden, m1, m2, key, z
  integer :: i, j
  golden = ior(shiftl(int(z'9E3779B9', int64), 32), int(z'7F4A7C15', int64))
  m1 = ior(shiftl(int(z'BF58476D', int64), 32), int(z'1CE4E5B9', int64))
  m2 = ior(shiftl(int(z'94D049BB', int64), 32), int(z'133111EB', int64))
  !$omp parallel do schedule(static) private(i, key, z)
  do j = 1, n
     do i = 1, n
        if (i == j) then
           if (has_diag /= 0) then
              a(i, j) = diag_val
           else
              a(i, j) = real(j, c_double)
           end if
        else
           key = shiftl(int(min(i, j) - 1, int64), 32) + int(max(i, j) - 1, int64) + int(seed, int64) * golden
           z = key + golden
           z = ieor(z, shiftr(z, 30)) * m1
           z = ieor(z, shiftr(z, 27)) * m2
           z = ieor(z, shiftr(z, 31))
           a(i, j) = real(shiftr(z, 11), c_double) * (1.0_c_double / 9007199254740992.0_c_double) * sparsity
        end if
     end do
  end do
  !$omp end parallel do
end subroutine ref_generate


!> What the host memory system delivers to a plain parallel read of the same matrix (sum of all entries under OpenMP, the
!> static schedule that first touched it): the yardstick next to MKL's DGEMV rate in bench.py's cpu_baseline.  Not reference code.
subroutine ref_stream_sum(n, a, total) bind(C)
  use iso_c_binding
  implicit none
  integer(c_int), value :: n
  real(c_double), intent(in) :: a(n, n)
  real(c_double), intent(out) :: total
  real(c_double) :: s
  integer :: i, j
  s = 0.0_c_double
  !$omp parallel do schedule(static) private(i) reduction(+:s)
  do j = 1, n
     do i = 1, n
        s = s + a(i, j)
     end do
  end do
  !$omp end parallel do
  total = s
end subroutine ref_stream_sum


!> y = A x for a column-major n x n matrix under OpenMP (every thread owns a contiguous range of rows and walks all columns):
!> what a sweep of A costs on these cores when the BLAS does not get in the way - the second yardstick next to MKL's DGEMV in
!> bench.py's cpu_baseline (on the AMD hosts of the GPU boxes MKL's DGEMV runs at a seventh of the rate a plain parallel read
!> of the same matrix reaches).  Not reference code.
subroutine ref_gemv_omp(n, a, x, y) bind(C)
  use iso_c_binding
  implicit none
  integer(c_int), value :: n
  real(c_double), intent(in) :: a(n, n), x(n)
  real(c_double), intent(out) :: y(n)
  integer :: i, j, i0, i1, t, nt
  integer, external :: omp_get_thread_num, omp_get_num_threads
  !$omp parallel private(i, j, i0, i1, t, nt)
  t = omp_get_thread_num()
  nt = omp_get_num_threads()
  i0 = int(int(n, 8) * t / nt) + 1
  i1 = int(int(n, 8) * (t + 1) / nt)
  y(i0:i1) = 0.0_c_double
  do j = 1, n
     do i = i0, i1
        y(i) = y(i) + a(i, j) * x(j)
     end do
  end do
  !$omp end parallel
end subroutine ref_gemv_omp
