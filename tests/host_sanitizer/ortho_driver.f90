!> Host-only driver of the rank decisions of the block orthonormalisation (fortran_davidson_amd/fortran/davidson_ortho.f90:
!> block_orthonormalise / ortho_pass_transform / dependent_columns / restart_transform) - the code that replaced the reference's
!> concatenate + Householder QR of the whole basis (src/davidson.f90:210-213, src/lapack_wrapper.f90:176-236) - on HOST arrays:
!> the N-long side (Gram products, block updates, replacement columns) is the `host_ortho` backend below, everything else is the
!> product's code.  No GPU, no HIP library: runs under AddressSanitizer in tests/test_host_sanitizer.py.
!>
!> For every case the driver builds a basis V (orthonormal, n x m) and a correction block T (n x kt), runs the product's passes and
!> checks
!>   (1) [V T_final] is orthonormal to 1e-11;
!>   (2) every original column of T that Householder QR of [V T] (DGEQRF, column order preserved: src/lapack_wrapper.f90:205-230) finds
!>       independent at the level the test fixes (|r_jj| >= keep_level * |t_j|) lies in span [V T_final] to 1e-9 of its norm;
!>   (3) no column QR keeps at that level is ever declared dependent, and the set of columns the product DECLARED dependent (and
!>       replaced by unit vectors of the start order) equals the set QR declares dependent, |r_jj| < drop_level * |t_j| - for inputs
!>       built so that no column falls between the two levels - in this sense:
!>         MATCH_ALWAYS  structural dependence (banded blocks, zero columns, columns inside the basis): equal under DAV_ORTHO_EARLY = 0
!>                       and 1 - what is left of such a column after a pass falls back into the span again, and the second pass
!>                       sees that ("twice is enough");
!>         MATCH_EARLY   dependence among generic columns (duplicates, sums, near-duplicates below drop_level): equal under the default
!>                       DAV_ORTHO_EARLY = 1, whose first pass looks at the pivots of the Gram block; with 0 the rounding noise left of
!>                       such a column is a generic direction, the pass normalises it and keeps it as the completion vector without
!>                       declaring anything - Householder QR completes with the direction of ITS rounding noise just the same - and
!>                       (1) + (2) are what is checked.
!> Between the levels (a column dependent to 1e-7 .. 1e-11 of its norm) keeping the direction or replacing it is a judgement call on
!> which DAV_ORTHO_EARLY = 0 / 1 differ; such columns are only checked through (1) and (2) (MATCH_NONE).
!> The program prints one line per case and ends with "ortho driver: ok"; any failure is an `error stop`.
module host_ortho_backend
  use numeric_kinds, only: dp
  use davidson_ortho, only: ortho_backend
  implicit none
  private
  public :: host_ortho

  type, extends(ortho_backend) :: host_ortho
     real(dp), allocatable :: v(:, :), t(:, :)
     integer, allocatable :: order(:)          !< start order: entry k (0-based) = row index of the unit vector
     integer :: ngram = 0, napply = 0
   contains
     procedure :: gram => host_gram
     procedure :: apply => host_apply
     procedure :: unit_column => host_unit_column
     procedure :: put_column => host_put_column
  end type host_ortho

contains

  subroutine host_gram(be, m, kt, c, g)
    class(host_ortho), intent(inout) :: be
    integer, intent(in) :: m, kt
    real(dp), intent(out) :: c(:, :), g(:, :)
    if (m > 0) c(1:m, 1:kt) = matmul(transpose(be%v(:, 1:m)), be%t(:, 1:kt))
    g(1:kt, 1:kt) = matmul(transpose(be%t(:, 1:kt)), be%t(:, 1:kt))
    be%ngram = be%ngram + 1
  end subroutine host_gram

  subroutine host_apply(be, m, kt, c, mm)
    class(host_ortho), intent(inout) :: be
    integer, intent(in) :: m, kt
    real(dp), intent(in) :: c(:, :), mm(:, :)
    real(dp), allocatable :: w(:, :)
    w = be%t(:, 1:kt)
    if (m > 0) w = w - matmul(be%v(:, 1:m), c(1:m, 1:kt))
    be%t(:, 1:kt) = matmul(w, mm(1:kt, 1:kt))
    be%napply = be%napply + 1
  end subroutine host_apply

  function host_unit_column(be, m, j, entry) result(ok)
    class(host_ortho), intent(inout) :: be
    integer, intent(in) :: m, j, entry
    logical :: ok
    ok = entry >= 0 .and. entry < size(be%order)
    if (.not. ok) return
    be%t(:, j) = 0.0_dp
    be%t(be%order(entry + 1), j) = 1.0_dp
  end function host_unit_column

  subroutine host_put_column(be, m, j, vec)
    class(host_ortho), intent(inout) :: be
    integer, intent(in) :: m, j
    real(dp), intent(in) :: vec(:)
    be%t(:, j) = vec
  end subroutine host_put_column

end module host_ortho_backend


program ortho_driver
  use numeric_kinds, only: dp
  use davidson_ortho
  use host_ortho_backend
  implicit none
  interface
     subroutine dgeqrf(m, n, a, lda, tau, work, lwork, info)
       import :: dp
       integer :: m, n, lda, lwork, info
       real(dp) :: a(lda, *), tau(*), work(*)
     end subroutine dgeqrf
  end interface
  real(dp), parameter :: keep_level = 1.0e-6_dp, drop_level = 1.0e-11_dp
  integer, parameter :: MATCH_NONE = 0, MATCH_ALWAYS = 1, MATCH_EARLY = 2
  integer, parameter :: n = 96
  integer :: ncases, nfail
  ncases = 0
  nfail = 0

  call banded_block()
  call block_diagonal_duplicates()
  call duplicated_and_zero_columns()
  call near_dependent_columns()
  call columns_inside_the_basis()
  call generic_full_rank()
  call restart_transform_cases()
  call dependent_columns_unit()
  if (nfail > 0) then
     print *, "ortho driver: ", nfail, " of ", ncases, " cases FAILED"
     error stop 1
  end if
  print "(a, i0, a, l1)", "ortho driver: ok (", ncases, " cases), DAV_ORTHO_EARLY on: ", ortho_early()

contains

  !> deterministic numbers in (-0.5, 0.5)
  function noise(seed, rows, cols) result(x)
    integer, intent(in) :: seed, rows, cols
    real(dp) :: x(rows, cols)
    integer :: j
    do j = 1, cols
       call pseudo_random_vector(x(:, j), seed + 131 * j)
    end do
  end function noise

  !> m unit vectors e_1..e_m (the reference's start basis, src/array_utils.f90:136-160) rotated among themselves: an orthonormal basis of
  !> the same span with dense coefficients
  function rotated_unit_basis(m, seed) result(v)
    integer, intent(in) :: m, seed
    real(dp) :: v(n, m)
    real(dp) :: q(m, m), r(m, m)
    integer :: i, j
    q = noise(seed, m, m)
    ! Gram-Schmidt twice (small, well conditioned)
    do j = 1, m
       do i = 1, 2
          if (j > 1) q(:, j) = q(:, j) - matmul(q(:, 1:j - 1), matmul(transpose(q(:, 1:j - 1)), q(:, j)))
       end do
       q(:, j) = q(:, j) / sqrt(sum(q(:, j)**2))
    end do
    r = q
    v = 0.0_dp
    v(1:m, 1:m) = r
  end function rotated_unit_basis

  !> run the product's passes on (v, t) and check (1)-(3); match: MATCH_NONE / MATCH_ALWAYS / MATCH_EARLY (see the header)
  subroutine run_case(name, v, t, match)
    character(len=*), intent(in) :: name
    real(dp), intent(in) :: v(:, :), t(:, :)
    integer, intent(in) :: match
    type(host_ortho) :: be
    integer :: m, kt, j, i, info, lwork
    logical, allocatable :: replaced(:), qr_dep(:), qr_keep(:)
    real(dp), allocatable :: a(:, :), tau(:), work(:), basis(:, :), gram(:, :), res(:)
    real(dp) :: orth, tn, worst_span
    logical :: ok
    m = size(v, 2)
    kt = size(t, 2)
    be%v = v
    be%t = t
    allocate(be%order(n))
    do i = 1, n
       be%order(i) = i                      ! the diagonal ascends with the index: entry k of the start order is row k + 1
    end do
    allocate(replaced(kt), qr_dep(kt), qr_keep(kt))
    call block_orthonormalise(be, n, m, kt, replaced=replaced)
    ! (1) orthonormality of the result
    allocate(basis(n, m + kt))
    basis(:, 1:m) = v
    basis(:, m + 1:) = be%t
    gram = matmul(transpose(basis), basis)
    orth = 0.0_dp
    do j = 1, m + kt
       do i = 1, m + kt
          orth = max(orth, abs(gram(i, j) - merge(1.0_dp, 0.0_dp, i == j)))
       end do
    end do
    ! Householder QR of [V T], column order preserved
    allocate(a(n, m + kt), tau(m + kt), work(1))
    a(:, 1:m) = v
    a(:, m + 1:) = t
    call dgeqrf(n, m + kt, a, n, tau, work, -1, info)
    lwork = max(1, int(work(1)))
    deallocate(work)
    allocate(work(lwork))
    call dgeqrf(n, m + kt, a, n, tau, work, lwork, info)
    if (info /= 0) error stop "dgeqrf"
    worst_span = 0.0_dp
    do j = 1, kt
       tn = sqrt(sum(t(:, j)**2))
       qr_dep(j) = abs(a(m + j, m + j)) < drop_level * tn .or. tn == 0.0_dp
       qr_keep(j) = abs(a(m + j, m + j)) >= keep_level * tn .and. tn > 0.0_dp
       if (qr_keep(j)) then
          res = t(:, j) - matmul(basis, matmul(transpose(basis), t(:, j)))     ! (2)
          worst_span = max(worst_span, sqrt(sum(res**2)) / tn)
       end if
    end do
    ok = orth < 1.0e-11_dp .and. worst_span < 1.0e-9_dp
    ok = ok .and. .not. any(replaced .and. qr_keep)
    if (match == MATCH_ALWAYS .or. (match == MATCH_EARLY .and. ortho_early())) ok = ok .and. all(replaced .eqv. qr_dep)
    ncases = ncases + 1
    print "(a, a, a, l1, a, es9.2, a, es9.2, a, i0, a, i0, a, i0, a, i0)", "case ", name, ": ok=", ok, " orth=", orth, " span=", worst_span, &
         " replaced=", count(replaced), " qr_dependent=", count(qr_dep), " grams=", be%ngram, " applies=", be%napply
    if (.not. ok) then
       nfail = nfail + 1
       print *, "   replaced:     ", replaced
       print *, "   qr dependent: ", qr_dep
       print *, "   qr keeps:     ", qr_keep
    end if
  end subroutine run_case

  !> The case that diverged until round 5: start vectors e_1..e_m of a matrix of bandwidth 2; the DPR corrections t = r / (theta - d)
  !> have the support of their residuals, rows 1..m+2 - after projection against span(e_1..e_m) the block has rank 2
  subroutine banded_block()
    integer, parameter :: m = 8
    real(dp) :: v(n, m), t(n, m)
    integer :: j
    v = 0.0_dp
    do j = 1, m
       v(j, j) = 1.0_dp
    end do
    t = 0.0_dp
    t(1:m + 2, :) = noise(3, m + 2, m)
    call run_case("banded: corrections confined to rows 1..m+2", v, t, MATCH_ALWAYS)
    v = rotated_unit_basis(m, 5)
    call run_case("banded, rotated basis", v, t, MATCH_ALWAYS)
    ! bandwidth 4: rank 4 beyond the basis
    t = 0.0_dp
    t(1:m + 4, :) = noise(7, m + 4, m)
    call run_case("banded, bandwidth 4 (rank 4 beyond the basis)", v, t, MATCH_ALWAYS)
  end subroutine banded_block

  !> two identical decoupled blocks and a basis that treats them alike: every correction comes twice
  subroutine block_diagonal_duplicates()
    integer, parameter :: m = 6, kt = 6
    real(dp) :: v(n, m), t(n, kt), half(n, kt / 2)
    v = rotated_unit_basis(m, 11)
    half = noise(13, n, kt / 2)
    t(:, 1:kt / 2) = half
    t(:, kt / 2 + 1:kt) = half
    call run_case("block diagonal: every correction twice", v, t, MATCH_EARLY)
    ! the copies scaled (the corrections of two equal blocks at different Ritz values are parallel, not equal)
    t(:, kt / 2 + 1:kt) = -3.5_dp * half
    call run_case("block diagonal: parallel corrections", v, t, MATCH_EARLY)
  end subroutine block_diagonal_duplicates

  subroutine duplicated_and_zero_columns()
    integer, parameter :: m = 10, kt = 7
    real(dp) :: v(n, m), t(n, kt)
    v = rotated_unit_basis(m, 17)
    t = noise(19, n, kt)
    t(:, 4) = t(:, 2)
    call run_case("one duplicated column", v, t, MATCH_EARLY)
    t = noise(23, n, kt)
    t(:, 3) = 0.0_dp
    call run_case("one zero column", v, t, MATCH_ALWAYS)
    t(:, 6) = 0.0_dp
    t(:, 7) = t(:, 1) + t(:, 2)
    call run_case("two zero columns and a sum of two others", v, t, MATCH_EARLY)
    t = 0.0_dp
    call run_case("a block of zeros", v, t, MATCH_ALWAYS)
  end subroutine duplicated_and_zero_columns

  !> t_3 = t_1 + delta * w with w a unit vector orthogonal to everything else: dependent to delta of its norm
  subroutine near_dependent_columns()
    integer, parameter :: m = 10, kt = 5
    real(dp), parameter :: deltas(6) = [1.0e-3_dp, 1.0e-5_dp, 1.0e-8_dp, 1.0e-10_dp, 1.0e-12_dp, 1.0e-14_dp]
    real(dp) :: v(n, m), t(n, kt), w(n)
    character(len=64) :: name
    integer :: k
    v = rotated_unit_basis(m, 29)
    do k = 1, size(deltas)
       t = noise(31, n, kt)
       t(1:m, :) = 0.0_dp                              ! the block orthogonal to the basis: only the mutual dependence is on trial
       w = 0.0_dp
       w(n) = 1.0_dp
       t(n, :) = 0.0_dp
       t(:, 3) = t(:, 1) + deltas(k) * sqrt(sum(t(:, 1)**2)) * w
       write (name, "(a, es8.1)") "near-dependent column, delta =", deltas(k)
       ! between drop_level and keep_level both answers are legitimate (see the header): no set comparison there
       call run_case(trim(name), v, t, merge(MATCH_EARLY, MATCH_NONE, deltas(k) >= keep_level .or. deltas(k) < drop_level))
    end do
  end subroutine near_dependent_columns

  !> corrections that lie in span(V) up to delta: after the projection only delta of them is left
  subroutine columns_inside_the_basis()
    integer, parameter :: m = 12, kt = 4
    real(dp), parameter :: deltas(4) = [1.0e-2_dp, 1.0e-5_dp, 1.0e-12_dp, 0.0_dp]
    real(dp) :: v(n, m), t(n, kt), coef(m, 1), extra(n, 1)
    character(len=64) :: name
    integer :: k
    v = rotated_unit_basis(m, 37)
    do k = 1, size(deltas)
       t = noise(41, n, kt)
       coef = noise(43, m, 1)
       extra = noise(47, n, 1)
       extra(1:m, 1) = 0.0_dp
       t(:, 2) = matmul(v, coef(:, 1)) + deltas(k) * extra(:, 1)
       write (name, "(a, es8.1)") "column inside span(V) up to", deltas(k)
       call run_case(trim(name), v, t, merge(MATCH_ALWAYS, MATCH_NONE, deltas(k) >= keep_level .or. deltas(k) < drop_level))
    end do
  end subroutine columns_inside_the_basis

  subroutine generic_full_rank()
    integer, parameter :: m = 16, kt = 16
    real(dp) :: v(n, m), t(n, kt)
    integer :: j
    v = rotated_unit_basis(m, 53)
    t = noise(59, n, kt)
    call run_case("generic block of full rank", v, t, MATCH_ALWAYS)
    do j = 1, kt
       t(:, j) = t(:, j) * 10.0_dp**(3 * j - 24)         ! column norms over 45 decades
    end do
    call run_case("full rank, column norms over 45 decades", v, t, MATCH_ALWAYS)
    ! no basis at all (m = 0): the very first block
    call run_case("no basis (m = 0)", v(:, 1:0), noise(61, n, 5), MATCH_ALWAYS)
  end subroutine generic_full_rank

  !> restart_transform: Y (m x kt, S-orthonormal in the driver) -> Y M with (Y M)^T (Y M) = I
  subroutine restart_transform_cases()
    integer, parameter :: m = 40, kt = 12
    real(dp) :: y(m, kt), g(kt, kt)
    integer :: i, j, k
    real(dp) :: dev
    do k = 1, 2
       y = noise(67 + k, m, kt)
       if (k == 2) then
          do j = 1, kt
             y(:, j) = y(:, j) * 10.0_dp**(j - 6)      ! badly scaled columns
          end do
          y(:, 5) = y(:, 4) * 2.0_dp + 1.0e-6_dp * y(:, 5)          ! and a nearly dependent one: the eigenvalue route
       end if
       call restart_transform(y, m, kt)
       g = matmul(transpose(y), y)
       dev = 0.0_dp
       do j = 1, kt
          do i = 1, kt
             dev = max(dev, abs(g(i, j) - merge(1.0_dp, 0.0_dp, i == j)))
          end do
       end do
       ncases = ncases + 1
       print "(a, i0, a, es9.2)", "case restart_transform ", k, ": deviation from orthonormal ", dev
       if (.not. dev < 1.0e-9_dp) nfail = nfail + 1
    end do
  end subroutine restart_transform_cases

  !> dependent_columns on Gram blocks with known answers: the left-to-right scan flags the LATER of two dependent columns
  subroutine dependent_columns_unit()
    integer, parameter :: kt = 6
    real(dp) :: x(n, kt), gs(kt, kt), nrm(kt)
    logical :: dep(kt)
    integer :: ndep, i, j
    x = noise(71, n, kt)
    x(:, 5) = x(:, 2)                                     ! column 5 repeats column 2
    x(:, 6) = x(:, 1) - x(:, 3)                           ! column 6 depends on 1 and 3
    do j = 1, kt
       nrm(j) = sqrt(sum(x(:, j)**2))
    end do
    gs = matmul(transpose(x), x)
    do j = 1, kt
       do i = 1, kt
          gs(i, j) = gs(i, j) / (nrm(i) * nrm(j))
       end do
    end do
    call dependent_columns(gs, kt, 1.0e-10_dp, dep, ndep)
    ncases = ncases + 1
    print *, "case dependent_columns: flagged ", dep
    if (ndep /= 2 .or. .not. (dep(5) .and. dep(6)) .or. any(dep(1:4))) nfail = nfail + 1
  end subroutine dependent_columns_unit

end program ortho_driver
