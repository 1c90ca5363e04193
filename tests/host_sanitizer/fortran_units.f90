!> Host-only Fortran units of the drop-in library (array_utils, lapack_wrapper) under AddressSanitizer: the publics the
!> reference's callers use (src/array_utils.f90:11-12, src/lapack_wrapper.f90:9-10), on small inputs with known answers.
program fortran_units
  use numeric_kinds, only: dp
  use array_utils, only: eye, norm, concatenate, diagonal, generate_preconditioner, generate_diagonal_dominant
  use lapack_wrapper, only: lapack_generalized_eigensolver, lapack_matmul, lapack_matrix_vector, lapack_qr, lapack_solver, &
       lapack_sort
  implicit none
  integer, parameter :: n = 37
  real(dp), allocatable :: a(:, :), b(:, :), q(:, :), vals(:), vecs(:, :), d(:), v(:, :), x(:, :), r(:, :)
  integer, allocatable :: keys(:)
  integer :: i

  a = generate_diagonal_dominant(n, 1.0e-2_dp)
  b = generate_diagonal_dominant(n, 1.0e-2_dp, 1.0_dp)
  if (maxval(abs(a - transpose(a))) /= 0.0_dp) error stop "generate_diagonal_dominant: not symmetric"
  d = diagonal(a)
  if (abs(d(n) - real(n, dp)) > 0.0_dp) error stop "diagonal"
  v = generate_preconditioner(d, 6)
  if (any(shape(v) /= [n, 6]) .or. abs(sum(v) - 6.0_dp) > 0.0_dp) error stop "generate_preconditioner"
  allocate(vals(n), vecs(n, n))
  call lapack_generalized_eigensolver(a, vals, vecs)
  r = lapack_matmul("N", "N", a, vecs)
  do i = 1, n
     if (norm(r(:, i) - vals(i) * vecs(:, i)) > 1.0e-10_dp) error stop "standard eigenpairs"
  end do
  call lapack_generalized_eigensolver(a, vals, vecs, b)
  r = lapack_matmul("N", "N", a, vecs) - lapack_matmul("N", "N", b, vecs) * spread(vals, 1, n)
  if (maxval(abs(r)) > 1.0e-9_dp) error stop "generalized eigenpairs"
  q = a(:, 1:9)
  call lapack_qr(q)
  if (maxval(abs(lapack_matmul("T", "N", q, q) - eye(9, 9))) > 1.0e-12_dp) error stop "lapack_qr"
  call concatenate(q, a(:, 10:12))
  if (any(shape(q) /= [n, 12])) error stop "concatenate"
  x = reshape(a(:, 3), [n, 1])
  r = x
  q = a                                   ! the solver factors its matrix argument in place
  call lapack_solver(q, r)
  if (norm(lapack_matrix_vector("N", a, r(:, 1)) - x(:, 1)) > 1.0e-9_dp) error stop "lapack_solver"
  d = [(real(mod(i * 7, n), dp), i = 1, n)]
  keys = lapack_sort("I", d)
  if (any(d(2:n) < d(1:n - 1)) .or. size(keys) /= n) error stop "lapack_sort"
  print *, "fortran units under the sanitizer: ok"
end program fortran_units
