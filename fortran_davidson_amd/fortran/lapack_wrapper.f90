!> Host-side LAPACK/BLAS helpers with the public surface of the reference's lapack_wrapper
!> (src/lapack_wrapper.f90:9-10) so user code that `use`s them keeps compiling.
!>
!> In this engine only lapack_generalized_eigensolver is on the solver path (the m x m
!> Rayleigh-Ritz problem stays on the host, BASELINE north_star); the N-long products the reference
!> routes through lapack_matmul / lapack_matrix_vector / lapack_qr / lapack_solver run on the GPU
!> (include/davidson_hip.h).  The remaining wrappers are kept as plain host utilities.
module lapack_wrapper
  use numeric_kinds, only: dp
  implicit none
  private
  public :: lapack_generalized_eigensolver, lapack_generalized_eigensolver_lowest, &
       lapack_matmul, lapack_matrix_vector, lapack_qr, lapack_solver, lapack_sort, &
       lapack_cholesky_inverse, lapack_rayleigh_ritz

  interface
     subroutine dsyev(jobz, uplo, n, a, lda, w, work, lwork, info)
       import :: dp
       character :: jobz, uplo
       integer :: n, lda, lwork, info
       real(dp) :: a(lda, *), w(*), work(*)
     end subroutine
     subroutine dsygv(itype, jobz, uplo, n, a, lda, b, ldb, w, work, lwork, info)
       import :: dp
       character :: jobz, uplo
       integer :: itype, n, lda, ldb, lwork, info
       real(dp) :: a(lda, *), b(ldb, *), w(*), work(*)
     end subroutine
     subroutine dsygvx(itype, jobz, range, uplo, n, a, lda, b, ldb, vl, vu, il, iu, abstol, m, w, z, ldz, &
          work, lwork, iwork, ifail, info)
       import :: dp
       character :: jobz, range, uplo
       integer :: itype, n, lda, ldb, il, iu, m, ldz, lwork, info, iwork(*), ifail(*)
       real(dp) :: a(lda, *), b(ldb, *), vl, vu, abstol, w(*), z(ldz, *), work(*)
     end subroutine
     subroutine dsyevd(jobz, uplo, n, a, lda, w, work, lwork, iwork, liwork, info)
       import :: dp
       character :: jobz, uplo
       integer :: n, lda, lwork, liwork, info, iwork(*)
       real(dp) :: a(lda, *), w(*), work(*)
     end subroutine
     subroutine dsygvd(itype, jobz, uplo, n, a, lda, b, ldb, w, work, lwork, iwork, liwork, info)
       import :: dp
       character :: jobz, uplo
       integer :: itype, n, lda, ldb, lwork, liwork, info, iwork(*)
       real(dp) :: a(lda, *), b(ldb, *), w(*), work(*)
     end subroutine
     subroutine dsyevr(jobz, range, uplo, n, a, lda, vl, vu, il, iu, abstol, m, w, z, ldz, isuppz, work, lwork, &
          iwork, liwork, info)
       import :: dp
       character :: jobz, range, uplo
       integer :: n, lda, il, iu, m, ldz, lwork, liwork, info, isuppz(*), iwork(*)
       real(dp) :: a(lda, *), vl, vu, abstol, w(*), z(ldz, *), work(*)
     end subroutine
     subroutine dsygst(itype, uplo, n, a, lda, b, ldb, info)
       import :: dp
       character :: uplo
       integer :: itype, n, lda, ldb, info
       real(dp) :: a(lda, *), b(ldb, *)
     end subroutine
     subroutine dtrsm(side, uplo, transa, diag, m, n, alpha, a, lda, b, ldb)
       import :: dp
       character :: side, uplo, transa, diag
       integer :: m, n, lda, ldb
       real(dp) :: alpha, a(lda, *), b(ldb, *)
     end subroutine
     subroutine dgeqrf(m, n, a, lda, tau, work, lwork, info)
       import :: dp
       integer :: m, n, lda, lwork, info
       real(dp) :: a(lda, *), tau(*), work(*)
     end subroutine
     subroutine dorgqr(m, n, k, a, lda, tau, work, lwork, info)
       import :: dp
       integer :: m, n, k, lda, lwork, info
       real(dp) :: a(lda, *), tau(*), work(*)
     end subroutine
     subroutine dsysv(uplo, n, nrhs, a, lda, ipiv, b, ldb, work, lwork, info)
       import :: dp
       character :: uplo
       integer :: n, nrhs, lda, ldb, lwork, info, ipiv(*)
       real(dp) :: a(lda, *), b(ldb, *), work(*)
     end subroutine
     subroutine dpotrf(uplo, n, a, lda, info)
       import :: dp
       character :: uplo
       integer :: n, lda, info
       real(dp) :: a(lda, *)
     end subroutine
     subroutine dtrtri(uplo, diag, n, a, lda, info)
       import :: dp
       character :: uplo, diag
       integer :: n, lda, info
       real(dp) :: a(lda, *)
     end subroutine
     subroutine dgemm(transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc)
       import :: dp
       character :: transa, transb
       integer :: m, n, k, lda, ldb, ldc
       real(dp) :: alpha, beta, a(lda, *), b(ldb, *), c(ldc, *)
     end subroutine
     subroutine dgemv(trans, m, n, alpha, a, lda, x, incx, beta, y, incy)
       import :: dp
       character :: trans
       integer :: m, n, lda, incx, incy
       real(dp) :: alpha, beta, a(lda, *), x(*), y(*)
     end subroutine
  end interface

contains

  !> All eigenpairs of the symmetric problem mtx*y = w*y, or mtx*y = w*stx*y when stx is present
  !> (DSYEV / DSYGV itype=1, upper triangle, ascending; contract of src/lapack_wrapper.f90:14-91).
  subroutine lapack_generalized_eigensolver(mtx, eigenvalues, eigenvectors, stx)
    real(dp), dimension(:, :), intent(in) :: mtx
    real(dp), dimension(:, :), intent(in), optional :: stx
    real(dp), dimension(size(mtx, 1)), intent(inout) :: eigenvalues
    real(dp), dimension(size(mtx, 1), size(mtx, 2)), intent(inout) :: eigenvectors
    real(dp), allocatable :: a(:, :), b(:, :), work(:)
    real(dp) :: query(1)
    integer :: n, info, lwork

    n = size(mtx, 1)
    allocate(a(n, n))
    a = mtx
    if (present(stx)) then
       allocate(b(n, n))
       b = stx
       call dsygv(1, "V", "U", n, a, n, b, n, eigenvalues, query, -1, info)
       call check_lapack_call(info, "DSYGV")
       lwork = max(1, int(query(1)))
       allocate(work(lwork))
       call dsygv(1, "V", "U", n, a, n, b, n, eigenvalues, work, lwork, info)
       call check_lapack_call(info, "DSYGV")
    else
       call dsyev("V", "U", n, a, n, eigenvalues, query, -1, info)
       call check_lapack_call(info, "DSYEV")
       lwork = max(1, int(query(1)))
       allocate(work(lwork))
       call dsyev("V", "U", n, a, n, eigenvalues, work, lwork, info)
       call check_lapack_call(info, "DSYEV")
    end if
    eigenvectors = a
  end subroutine lapack_generalized_eigensolver

  !> Rayleigh-Ritz step of the outer loop: the lowest `nvec` eigenpairs of the projected problem (all of
  !> them when nvec = n).  Small problems (n < 48) take exactly the reference's route (DSYEV / DSYGV,
  !> lapack_generalized_eigensolver above).  Above that the host eigensolver is the largest non-device cost of
  !> an iteration (sequential MKL on the MI355X host: n = 64: 202 us DSYEV / 152 us DSYEVD, n = 256: 6.2 / 3.8 ms;
  !> n = 800: 170 ms DSYEV against a 30 ms sweep of a 160 GB matrix), so: all pairs -> divide and conquer
  !> (DSYEVD / DSYGVD); a SMALL leading subset (nvec <= n/8) -> MRRR on that subset only (DSYEVR; generalized:
  !> Cholesky reduction DPOTRF + DSYGST, back-transformation DTRSM).  Where the line is, measured on the MI355X
  !> host (profiles/tools/rr_time2.py, sequential MKL): a quarter of the pairs costs DSYEVR MORE than DSYEVD
  !> costs for all of them (n = 128: 1108 against 754 us, n = 400: 13.3 against 10.7 ms), an eighth less
  !> (n = 128: 646 us, n = 256: 3.0 against 3.9 ms, n = 800: 46 against 64 ms) - until round 4 the line was at
  !> n/2, and the restart-size problem of a configs[2] solve (32 of 128 pairs) ran the slower of the two.
  !> Eigenvalues ascending; eigenvectors normalised as DSYEV / DSYGV itype=1 do (y^T stx y = I).
  subroutine lapack_rayleigh_ritz(mtx, eigenvalues, eigenvectors, nvec, stx)
    real(dp), dimension(:, :), intent(in) :: mtx
    real(dp), dimension(:, :), intent(in), optional :: stx
    integer, intent(in) :: nvec
    real(dp), dimension(size(mtx, 1)), intent(inout) :: eigenvalues
    real(dp), dimension(size(mtx, 1), size(mtx, 2)), intent(inout) :: eigenvectors
    integer, parameter :: switch_order = 48
    real(dp), allocatable :: a(:, :), b(:, :), z(:, :), w(:), work(:)
    integer, allocatable :: iwork(:), isuppz(:)
    real(dp) :: query(1)
    integer :: n, info, lwork, liwork, found, iquery(1)

    n = size(mtx, 1)
    if (n < switch_order) then
       call lapack_generalized_eigensolver(mtx, eigenvalues, eigenvectors, stx)
       return
    end if
    allocate(a(n, n))
    a = mtx
    if (present(stx)) then
       allocate(b(n, n))
       b = stx
    end if
    if (8 * nvec > n) then
       if (present(stx)) then
          call dsygvd(1, "V", "U", n, a, n, b, n, eigenvalues, query, -1, iquery, -1, info)
          call check_lapack_call(info, "DSYGVD")
          lwork = max(1, int(query(1)))
          liwork = max(1, iquery(1))
          allocate(work(lwork), iwork(liwork))
          call dsygvd(1, "V", "U", n, a, n, b, n, eigenvalues, work, lwork, iwork, liwork, info)
          call check_lapack_call(info, "DSYGVD")
       else
          call dsyevd("V", "U", n, a, n, eigenvalues, query, -1, iquery, -1, info)
          call check_lapack_call(info, "DSYEVD")
          lwork = max(1, int(query(1)))
          liwork = max(1, iquery(1))
          allocate(work(lwork), iwork(liwork))
          call dsyevd("V", "U", n, a, n, eigenvalues, work, lwork, iwork, liwork, info)
          call check_lapack_call(info, "DSYEVD")
       end if
       eigenvectors = a
       return
    end if
    ! leading subset
    if (present(stx)) then
       call dpotrf("U", n, b, n, info)                       ! stx = U^T U
       call check_lapack_call(info, "DPOTRF")
       call dsygst(1, "U", n, a, n, b, n, info)              ! a <- U^-T a U^-1
       call check_lapack_call(info, "DSYGST")
    end if
    allocate(z(n, nvec), w(n), isuppz(2 * n))
    call dsyevr("V", "I", "U", n, a, n, 0.0_dp, 0.0_dp, 1, nvec, 0.0_dp, found, w, z, n, isuppz, query, -1, &
         iquery, -1, info)
    call check_lapack_call(info, "DSYEVR")
    lwork = max(1, int(query(1)))
    liwork = max(1, iquery(1))
    allocate(work(lwork), iwork(liwork))
    call dsyevr("V", "I", "U", n, a, n, 0.0_dp, 0.0_dp, 1, nvec, 0.0_dp, found, w, z, n, isuppz, work, lwork, &
         iwork, liwork, info)
    call check_lapack_call(info, "DSYEVR")
    if (found /= nvec) call check_lapack_call(-99, "DSYEVR (eigenpairs found)")
    if (present(stx)) call dtrsm("L", "U", "N", "N", n, nvec, 1.0_dp, b, n, z, n)     ! y = U^-1 z
    eigenvalues = huge(1.0_dp)
    eigenvalues(1:nvec) = w(1:nvec)
    eigenvectors = 0.0_dp
    eigenvectors(:, 1:nvec) = z
  end subroutine lapack_rayleigh_ritz

  !> Lowest `lowest` eigenpairs through DSYGVX.  Dead code in the reference (no caller, abstol
  !> uninitialised, src/lapack_wrapper.f90:93-174); kept signature-compatible and made well defined.
  subroutine lapack_generalized_eigensolver_lowest(mtx, stx, eigenvalues, eigenvectors, lowest)
    real(dp), dimension(:, :), intent(in) :: mtx, stx
    integer, intent(in) :: lowest
    real(dp), dimension(lowest), intent(inout) :: eigenvalues
    real(dp), dimension(size(mtx, 1), lowest), intent(inout) :: eigenvectors
    real(dp), allocatable :: a(:, :), b(:, :), w(:), z(:, :), work(:)
    integer, allocatable :: iwork(:), ifail(:)
    real(dp) :: query(1)
    integer :: n, info, lwork, found

    n = size(mtx, 1)
    allocate(a(n, n), b(n, n), w(n), z(n, lowest), iwork(5 * n), ifail(n))
    a = mtx
    b = stx
    call dsygvx(1, "V", "I", "U", n, a, n, b, n, 0.0_dp, 0.0_dp, 1, lowest, 0.0_dp, found, w, z, n, &
         query, -1, iwork, ifail, info)
    call check_lapack_call(info, "DSYGVX")
    lwork = max(1, int(query(1)))
    allocate(work(lwork))
    call dsygvx(1, "V", "I", "U", n, a, n, b, n, 0.0_dp, 0.0_dp, 1, lowest, 0.0_dp, found, w, z, n, &
         work, lwork, iwork, ifail, info)
    call check_lapack_call(info, "DSYGVX")
    eigenvalues = w(1:lowest)
    eigenvectors = z
  end subroutine lapack_generalized_eigensolver_lowest

  !> Replace `basis` by an orthonormal basis of its column space (thin Q of Householder QR).
  subroutine lapack_qr(basis)
    real(dp), dimension(:, :), intent(inout) :: basis
    real(dp), allocatable :: q(:, :), tau(:), work(:)
    real(dp) :: query(1)
    integer :: m, n, info, lwork

    m = size(basis, 1)
    n = size(basis, 2)
    allocate(q(m, n), tau(max(1, min(m, n))))
    q = basis
    call dgeqrf(m, n, q, m, tau, query, -1, info)
    call check_lapack_call(info, "DGEQRF")
    lwork = max(1, int(query(1)))
    allocate(work(lwork))
    call dgeqrf(m, n, q, m, tau, work, lwork, info)
    call check_lapack_call(info, "DGEQRF")
    deallocate(work)
    call dorgqr(m, n, min(m, n), q, m, tau, query, -1, info)
    call check_lapack_call(info, "DORGQR")
    lwork = max(1, int(query(1)))
    allocate(work(lwork))
    call dorgqr(m, n, min(m, n), q, m, tau, work, lwork, info)
    call check_lapack_call(info, "DORGQR")
    basis = q
  end subroutine lapack_qr

  !> Solve arr * x = brr for a symmetric arr (DSYSV, upper), x overwrites brr.  A singular pivot is
  !> replaced by tiny() and the solve repeated once (behaviour of src/lapack_wrapper.f90:238-277).
  subroutine lapack_solver(arr, brr)
    real(dp), dimension(:, :), intent(inout) :: arr, brr
    real(dp), allocatable :: work(:)
    integer, allocatable :: ipiv(:)
    real(dp) :: query(1)
    integer :: n, nrhs, info, lwork

    n = size(arr, 1)
    nrhs = size(brr, 2)
    allocate(ipiv(n))
    call dsysv("U", n, nrhs, arr, n, ipiv, brr, n, query, -1, info)
    call check_lapack_call(info, "DSYSV")
    lwork = max(1, int(query(1)))
    allocate(work(lwork))
    call dsysv("U", n, nrhs, arr, n, ipiv, brr, n, work, lwork, info)
    if (info > 0) then
       arr(info, info) = tiny(1.0_dp)
       call dsysv("U", n, nrhs, arr, n, ipiv, brr, n, work, lwork, info)
    end if
    call check_lapack_call(info, "DSYSV")
  end subroutine lapack_solver

  !> alpha * op(arr) * op(brr) through DGEMM.
  function lapack_matmul(transA, transB, arr, brr, alpha) result(mtx)
    character(len=1), intent(in) :: transA, transB
    real(dp), dimension(:, :), intent(in) :: arr, brr
    real(dp), optional, intent(in) :: alpha
    real(dp), dimension(:, :), allocatable :: mtx
    real(dp) :: scale
    integer :: m, n, k

    scale = 1.0_dp
    if (present(alpha)) scale = alpha
    m = merge(size(arr, 2), size(arr, 1), transA == 'T')
    k = merge(size(arr, 1), size(arr, 2), transA == 'T')
    n = merge(size(brr, 1), size(brr, 2), transB == 'T')
    allocate(mtx(m, n))
    mtx = 0.0_dp
    call dgemm(transA, transB, m, n, k, scale, arr, size(arr, 1), brr, size(brr, 1), 0.0_dp, mtx, m)
  end function lapack_matmul

  !> alpha * op(mtx) * vector through DGEMV.
  function lapack_matrix_vector(transA, mtx, vector, alpha) result(rs)
    character(len=1), intent(in) :: transA
    real(dp), dimension(:, :), intent(in) :: mtx
    real(dp), dimension(:), intent(in) :: vector
    real(dp), optional, intent(in) :: alpha
    real(dp), dimension(:), allocatable :: rs
    real(dp) :: scale

    scale = 1.0_dp
    if (present(alpha)) scale = alpha
    allocate(rs(merge(size(mtx, 2), size(mtx, 1), transA == 'T')))
    rs = 0.0_dp
    call dgemv(transA, size(mtx, 1), size(mtx, 2), scale, mtx, size(mtx, 1), vector, 1, 0.0_dp, rs, 1)
  end function lapack_matrix_vector

  !> Sort `vector` in place ('I' increasing / 'D' decreasing) and return, for every ORIGINAL
  !> position i, the position keys(i) its value takes in the sorted vector.  Stable merge sort in
  !> O(n log n); equal values keep their original order (the reference's O(n^2) key recovery,
  !> src/lapack_wrapper.f90:384-390, is undefined for duplicates - SURVEY Appendix B).
  function lapack_sort(id, vector) result(keys)
    real(dp), dimension(:), intent(inout) :: vector
    character(len=1), intent(in) :: id
    integer, dimension(size(vector)) :: keys
    integer, allocatable :: perm(:), tmp(:)
    real(dp), allocatable :: sorted(:)
    integer :: n, i, width, lo, mid, hi, a, b, k
    logical :: take_left

    n = size(vector)
    allocate(perm(n), tmp(n), sorted(n))
    perm = [(i, i = 1, n)]
    width = 1
    do while (width < n)
       lo = 1
       do while (lo <= n)
          mid = min(lo + width, n + 1)
          hi = min(lo + 2 * width, n + 1)
          a = lo
          b = mid
          do k = lo, hi - 1
             if (a < mid .and. b < hi) then
                if (id == 'D') then
                   take_left = vector(perm(a)) >= vector(perm(b))
                else
                   take_left = vector(perm(a)) <= vector(perm(b))
                end if
             else
                take_left = a < mid
             end if
             if (take_left) then
                tmp(k) = perm(a)
                a = a + 1
             else
                tmp(k) = perm(b)
                b = b + 1
             end if
          end do
          lo = hi
       end do
       perm = tmp
       width = 2 * width
    end do
    do i = 1, n
       sorted(i) = vector(perm(i))
       keys(perm(i)) = i
    end do
    vector = sorted
  end function lapack_sort

  !> Upper-triangular rinv with rinv^T * g * rinv = I (g = R^T R by DPOTRF, rinv = R^-1 by DTRTRI).
  !> info /= 0 (g not numerically positive definite) is returned to the caller, not fatal: the block
  !> orthonormalisation then falls back to the eigen-decomposition route.  Not part of the reference's
  !> wrapper set; used by the device driver for the k x k basis transforms.
  subroutine lapack_cholesky_inverse(g, rinv, info)
    real(dp), dimension(:, :), intent(in) :: g
    real(dp), dimension(size(g, 1), size(g, 2)), intent(out) :: rinv
    integer, intent(out) :: info
    integer :: n, i, j
    n = size(g, 1)
    rinv = g
    call dpotrf("U", n, rinv, n, info)
    if (info /= 0) return
    call dtrtri("U", "N", n, rinv, n, info)
    if (info /= 0) return
    do j = 1, n
       do i = j + 1, n
          rinv(i, j) = 0.0_dp
       end do
    end do
  end subroutine lapack_cholesky_inverse

  subroutine check_lapack_call(info, name)
    integer, intent(in) :: info
    character(len=*), intent(in) :: name
    if (info /= 0) then
       print *, "call to subroutine: ", name, " has failed!"
       print *, "info: ", info
       error stop
    end if
  end subroutine check_lapack_call

end module lapack_wrapper
