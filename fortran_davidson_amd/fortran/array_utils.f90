!> Small host utilities with the public surface of the reference's array_utils
!> (src/array_utils.f90:11-12).  On the solver path their N-long work happens on the GPU
!> (dav_init_basis, dav_set_dense_generated, fused norms); these host versions serve user code and
!> the API-faithful matrix-free path.
module array_utils
  use numeric_kinds, only: dp
  use lapack_wrapper, only: lapack_sort
  use, intrinsic :: iso_fortran_env, only: int64
  implicit none
  private
  public :: concatenate, diagonal, eye, generate_diagonal_dominant, norm, generate_preconditioner

contains

  !> m x n matrix with alpha (default 1) on the main diagonal.
  pure function eye(m, n, alpha)
    integer, intent(in) :: n, m
    real(dp), intent(in), optional :: alpha
    real(dp), dimension(m, n) :: eye
    integer :: i
    real(dp) :: x
    x = 1.0_dp
    if (present(alpha)) x = alpha
    eye = 0.0_dp
    do i = 1, min(m, n)
       eye(i, i) = x
    end do
  end function eye

  !> Euclidean norm, unscaled sum of squares (as src/array_utils.f90:46-53).
  pure function norm(vector)
    real(dp), dimension(:), intent(in) :: vector
    real(dp) :: norm
    norm = sqrt(sum(vector * vector))
  end function norm

  !> arr <- [arr, brr]
  subroutine concatenate(arr, brr)
    real(dp), dimension(:, :), intent(inout), allocatable :: arr
    real(dp), dimension(:, :), intent(in) :: brr
    real(dp), dimension(:, :), allocatable :: wide
    integer :: nc
    nc = size(arr, 2)
    allocate(wide(size(arr, 1), nc + size(brr, 2)))
    wide(:, :nc) = arr
    wide(:, nc + 1:) = brr
    call move_alloc(wide, arr)
  end subroutine concatenate

  !> Symmetric test matrix: off-diagonal entries uniform in [0, sparsity), diagonal = row index or
  !> diag_val (semantics of src/array_utils.f90:86-113).  The uniform stream is the counter-based
  !> splitmix64 hash shared bit for bit with the device generator (csrc/common.h) and the oracle, so
  !> host and GPU build the same matrix.  `seed` selects the stream; without it successive calls use
  !> streams 1, 2, 3, ... - like the reference, whose `random_number` stream advances, two calls in
  !> one program give two different matrices, but here reproducibly.
  function generate_diagonal_dominant(m, sparsity, diag_val, seed) result(arr)
    integer, intent(in) :: m
    real(dp) :: sparsity
    real(dp), optional :: diag_val
    integer, intent(in), optional :: seed
    real(dp), dimension(m, m) :: arr
    integer(int64) :: s, key, lo, hi
    integer :: i, j
    integer, save :: calls_without_seed = 0
    if (present(seed)) then
       s = int(seed, int64)
    else
       calls_without_seed = calls_without_seed + 1
       s = int(calls_without_seed, int64)
    end if
    do j = 1, m
       do i = 1, j - 1
          lo = int(i - 1, int64)
          hi = int(j - 1, int64)
          key = shiftl(lo, 32) + hi + s * golden()
          arr(i, j) = real(shiftr(splitmix64(key), 11), dp) * (1.0_dp / 9007199254740992.0_dp) * sparsity
          arr(j, i) = arr(i, j)
       end do
       if (present(diag_val)) then
          arr(j, j) = diag_val
       else
          arr(j, j) = real(j, dp)
       end if
    end do
  end function generate_diagonal_dominant

  pure function golden() result(g)
    integer(int64) :: g
    g = ior(shiftl(int(z'9E3779B9', int64), 32), int(z'7F4A7C15', int64))
  end function golden

  !> splitmix64 finaliser.  int64 arithmetic wraps modulo 2^64 (two's complement), which is exactly the
  !> unsigned arithmetic of the C/HIP and numpy versions; shiftr is a logical shift.
  pure function splitmix64(z0) result(z)
    integer(int64), intent(in) :: z0
    integer(int64) :: z
    z = z0 + golden()
    z = ieor(z, shiftr(z, 30)) * ior(shiftl(int(z'BF58476D', int64), 32), int(z'1CE4E5B9', int64))
    z = ieor(z, shiftr(z, 27)) * ior(shiftl(int(z'94D049BB', int64), 32), int(z'133111EB', int64))
    z = ieor(z, shiftr(z, 31))
  end function splitmix64

  !> Main diagonal of a square matrix.
  function diagonal(matrix)
    real(dp), dimension(:, :), intent(in) :: matrix
    real(dp), dimension(size(matrix, 1)) :: diagonal
    integer :: i
    do i = 1, size(matrix, 1)
       diagonal(i) = matrix(i, i)
    end do
  end function diagonal

  !> Initial basis: column i is the unit vector at the i-th smallest entry of diag (which is sorted
  !> in place, as in src/array_utils.f90:136-160).
  function generate_preconditioner(diag, dim_sub) result(precond)
    real(dp), dimension(:), intent(inout) :: diag
    integer, intent(in) :: dim_sub
    real(dp), dimension(size(diag), dim_sub) :: precond
    integer, dimension(size(diag)) :: keys
    integer :: i
    keys = lapack_sort('I', diag)
    precond = 0.0_dp
    do i = 1, size(diag)
       if (keys(i) <= dim_sub) precond(i, keys(i)) = 1.0_dp
    end do
  end function generate_preconditioner

end module array_utils
