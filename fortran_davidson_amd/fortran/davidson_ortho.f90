!> Block orthonormalisation of the Davidson driver: everything that decides, from k x k Gram blocks, how a correction block
!> T = V(:, m+1:m+kt) is made orthonormal to the basis V(:, 1:m) and to itself - the transform of one block Gram-Schmidt pass, the
!> rank decisions (which columns are numerically dependent and get replaced), the transform of a generalized restart - in place of
!> the reference's concatenate + Householder QR of the whole basis (src/davidson.f90:210-213, src/lapack_wrapper.f90:176-236).
!> Host only: the N-long work (Gram products, block updates, replacement columns) goes through the abstract `ortho_backend`, which
!> the driver implements on the device (davidson.f90: device_ortho) and tests/host_sanitizer/ortho_driver.f90 on host arrays - the
!> same decision code runs under AddressSanitizer without a GPU, against the column choices of Householder QR.
module davidson_ortho
  use numeric_kinds, only: dp
  use iso_c_binding, only: c_int64_t
  use lapack_wrapper, only: lapack_rayleigh_ritz, lapack_cholesky_inverse, lapack_matmul
  use davidson_knobs, only: trace_iterations
  implicit none
  private
  public :: ortho_backend, block_orthonormalise, ortho_pass_transform, dependent_columns, restart_transform, transform_projected, &
       ortho_early, pseudo_random_vector

  !> The N-long side of a pass.  Columns are numbered within the block (1..kt); the block stands behind m basis columns.
  type, abstract :: ortho_backend
   contains
     !> c(1:m, 1:kt) = V^T T, g(1:kt, 1:kt) = T^T T of the block as it stands
     procedure(gram_interface), deferred :: gram
     !> T <- (T - V c) mm
     procedure(apply_interface), deferred :: apply
     !> T(:, j) <- the unit vector at entry `entry` (0-based) of the start order (the order of the smallest diagonal entries of A);
     !> .false. - and the column untouched - when there is no such entry
     procedure(unit_interface), deferred :: unit_column
     !> T(:, j) <- vec (all n rows)
     procedure(put_interface), deferred :: put_column
  end type ortho_backend

  abstract interface
     subroutine gram_interface(be, m, kt, c, g)
       import :: ortho_backend, dp
       class(ortho_backend), intent(inout) :: be
       integer, intent(in) :: m, kt
       real(dp), intent(out) :: c(:, :), g(:, :)
     end subroutine gram_interface
     subroutine apply_interface(be, m, kt, c, mm)
       import :: ortho_backend, dp
       class(ortho_backend), intent(inout) :: be
       integer, intent(in) :: m, kt
       real(dp), intent(in) :: c(:, :), mm(:, :)
     end subroutine apply_interface
     function unit_interface(be, m, j, entry) result(ok)
       import :: ortho_backend
       class(ortho_backend), intent(inout) :: be
       integer, intent(in) :: m, j, entry
       logical :: ok
     end function unit_interface
     subroutine put_interface(be, m, j, vec)
       import :: ortho_backend, dp
       class(ortho_backend), intent(inout) :: be
       integer, intent(in) :: m, j
       real(dp), intent(in) :: vec(:)
     end subroutine put_interface
  end interface

contains

  !> Orthonormalise the kt columns T = V(:, m+1:m+kt) against V(:, 1:m) and among themselves
  !> (replaces concatenate + lapack_qr of the whole basis, src/davidson.f90:210-213).
  !> Each pass: one device Gram [V T]^T T, a kt x kt factorisation on the host (ortho_pass_transform:
  !> T <- (T - V C) M), one device block update.  Two passes give orthonormality to
  !> rounding; a direction that is numerically dependent (e.g. the correction of an already converged
  !> pair) is replaced by a deterministic pseudo-random vector, as Householder QR would complete the
  !> basis with an arbitrary direction.
  !> only_first = .true.: return after the first pass that applied a transform (its number in last_pass) - the driver then
  !> sweeps the block and runs the last pass together with the projection (project_with_last_pass); first_pass: number of
  !> the first pass made here (a continuation).
  subroutine block_orthonormalise(be, n, m, kt, c_first, g_first, only_first, last_pass, first_pass, replaced)
    class(ortho_backend), intent(inout) :: be
    integer, intent(in) :: n, m, kt
    !> Gram blocks V^T T and T^T T of the block as it stands (first pass), when the caller already has them
    real(dp), intent(in), optional :: c_first(:, :), g_first(:, :)
    logical, intent(in), optional :: only_first
    integer, intent(out), optional :: last_pass
    integer, intent(in), optional :: first_pass
    !> columns of the block that were declared numerically dependent and replaced (any round)
    logical, intent(out), optional :: replaced(kt)
    integer, parameter :: max_pass = 8
    real(dp), allocatable :: c(:, :), g(:, :), mm(:, :), vec(:)
    logical, allocatable :: null_cols(:)
    integer, allocatable :: queue(:)
    integer :: pass, j, nnull, pass0, nqueue, qpos, rounds
    integer, parameter :: max_rounds = 12
    logical :: first_gram
    logical :: got_unit
    real(dp) :: wmax, wmin
    logical :: clean, stop_early

    allocate(c(max(m, 1), kt), g(kt, kt), mm(kt, kt), null_cols(kt))
    if (present(replaced)) replaced = .false.
    clean = .false.
    stop_early = .false.
    if (present(only_first)) stop_early = only_first
    pass0 = 1
    if (present(first_pass)) pass0 = first_pass
    if (present(last_pass)) last_pass = pass0
    pass = pass0
    rounds = 0
    first_gram = .true.
    do
       if (first_gram .and. present(c_first)) then
          if (m > 0) c(1:m, :) = c_first
          g = g_first
       else
          call be%gram(m, kt, c, g)
       end if
       first_gram = .false.
       call ortho_pass_transform(pass, m, kt, c, g, mm, wmin, wmax, null_cols, nnull, detect=rounds < max_rounds)
       if (nnull > 0) then
          ! replace numerically null columns and repeat the pass (a replacement round is not a pass: however many rounds a block
          ! needs, its passes are still to come; from the seventh round on every replacement is pseudo-random - generic vectors
          ! cannot come back null as long as the basis is narrower than the space).  A block that twelve rounds have not settled
          ! is a block the relative tests misjudge: they are switched off for it (detect) and the passes go on as they did
          ! before those tests existed - rescaling, with the eigenvalue floor
          rounds = rounds + 1
          if (rounds > max_rounds + 4) then
             print *, "generalized_eigensolver: a correction block keeps columns that are exactly zero"
             error stop
          end if
          ! Round 1: the unit vector at the column's own entry of the start order (the (m + j)-th smallest diagonal entry: the
          ! direction the initial guess would have taken next, and what the reference's Householder QR leaves in such a column when
          ! the diagonal ascends with the index).  Rounds 2-6, for a column whose unit vector came back null (it lay in the span
          ! of the healthy columns): the entries of THEIR slots, then the entries behind the block.  After that pseudo-random vectors.
          if (.not. allocated(queue)) then
             allocate(queue(2 * kt))
             nqueue = 0
             do j = 1, kt
                if (.not. null_cols(j)) then
                   nqueue = nqueue + 1
                   queue(nqueue) = m + j - 1
                end if
             end do
             do j = 1, kt
                nqueue = nqueue + 1
                queue(nqueue) = m + kt + j - 1
             end do
             qpos = 0
          end if
          do j = 1, kt
             if (null_cols(j)) then
                if (present(replaced)) replaced(j) = .true.
                got_unit = .false.
                if (rounds == 1) then
                   got_unit = be%unit_column(m, j, m + j - 1)
                else if (rounds <= 6 .and. qpos < nqueue) then
                   qpos = qpos + 1
                   got_unit = be%unit_column(m, j, queue(qpos))
                end if
                if (trace_iterations()) print "(a, i0, a, i0, a, i0, a, i0, a, l1)", "davidson trace: block at m=", m, " pass ", pass, " round ", rounds, &
                     ": column ", j, " is numerically dependent; replaced by a unit vector: ", got_unit
                if (.not. got_unit) then
                   allocate(vec(n))
                   call pseudo_random_vector(vec, m + j + 7919 * (pass + 31 * rounds))
                   call be%put_column(m, j, vec)
                   deallocate(vec)
                end if
             end if
          end do
          cycle
       end if
       call be%apply(m, kt, c, mm)
       if (present(last_pass)) last_pass = pass
       if (stop_early) return
       ! a pass that started from a nearly orthonormal block (all scaled Gram eigenvalues close to 1
       ! and negligible overlap with V) leaves it orthonormal to rounding
       if (pass >= 2 .and. wmin > 0.5_dp .and. wmax < 2.0_dp) then
          clean = .true.
          exit
       end if
       if (pass >= max(max_pass, pass0 + 2)) exit
       pass = pass + 1
    end do
    if (.not. clean) then
       print *, "Warning: block orthonormalisation did not settle in ", max_pass, " passes"
    end if
  end subroutine block_orthonormalise

  !> raw = [V T']^T (Op T') ((m + kt) x kt) -> the same blocks for T'' = (T' - V C) M, whose image Op T'' = (Op T' - (Op V) C) M
  !> follows it: rows 1:m  V^T Op T'' = (raw_V - P C) M,  rows m+1:  T''^T Op T'' = M^T (raw_T - C^T raw_V - raw_V^T C + C^T P C) M,
  !> with P = pm(1:m, 1:m) the projected matrix of the basis so far (Op symmetric, as everywhere).
  subroutine transform_projected(pm, raw, c2, mm, m, kt)
    real(dp), intent(in) :: pm(:, :), c2(:, :), mm(:, :)
    real(dp), intent(inout) :: raw(:, :)
    integer, intent(in) :: m, kt
    real(dp), allocatable :: pc(:, :), newv(:, :), tt(:, :)
    integer :: p
    p = m + kt
    pc = lapack_matmul("N", "N", pm(1:m, 1:m), c2(1:m, 1:kt))
    newv = lapack_matmul("N", "N", raw(1:m, 1:kt) - pc, mm(1:kt, 1:kt))
    tt = raw(m + 1:p, 1:kt) - lapack_matmul("T", "N", c2(1:m, 1:kt), raw(1:m, 1:kt)) &
         - lapack_matmul("T", "N", raw(1:m, 1:kt), c2(1:m, 1:kt)) + lapack_matmul("T", "N", c2(1:m, 1:kt), pc)
    raw(1:m, 1:kt) = newv
    raw(m + 1:p, 1:kt) = lapack_matmul("T", "N", mm(1:kt, 1:kt), lapack_matmul("N", "N", tt, mm(1:kt, 1:kt)))
  end subroutine transform_projected

  !> The transform of ONE block Gram-Schmidt pass from its Gram blocks C = V^T T (m x kt) and G = T^T T (kt x kt):
  !> T <- (T - V C) M.  wmin / wmax: conditioning of the scaled Gram block G' = D (G - C^T C) D the pass started from (a pass
  !> numbered >= 2 with wmin > 0.5 and wmax < 2 leaves the block orthonormal to rounding).  nnull > 0: the columns flagged in
  !> null_cols are numerically null - no transform is made, the caller replaces them and repeats the pass.
  subroutine ortho_pass_transform(pass, m, kt, c, g, mm, wmin, wmax, null_cols, nnull, detect)
    integer, intent(in) :: pass, m, kt
    !> .false.: only the absolute test for null columns (a zero correction) - the relative tests of dependence are skipped
    logical, intent(in), optional :: detect
    real(dp), intent(in) :: c(:, :), g(:, :)
    real(dp), intent(out) :: mm(kt, kt), wmin, wmax
    logical, intent(out) :: null_cols(kt)
    integer, intent(out) :: nnull
    real(dp), parameter :: floor_rel = 1.0e-14_dp, again_rel = 1.0e-10_dp, first_rel = 1.0e-13_dp
    real(dp), allocatable :: gp(:, :), d(:), w(:), u(:, :)
    integer :: j, l, info
    logical :: chol_ok, relative_tests
    real(dp) :: dev

    relative_tests = .true.
    if (present(detect)) relative_tests = detect
    allocate(gp(kt, kt), d(kt), w(kt), u(kt, kt))
    wmin = 0.0_dp
    wmax = huge(1.0_dp)
    gp = g(1:kt, 1:kt)
    if (m > 0) then
       if (m * kt >= 4096) then
          gp = gp - lapack_matmul("T", "N", c(1:m, 1:kt), c(1:m, 1:kt))      ! DGEMM: the intrinsic is O(100 ms) at m = kt = 400
       else
          gp = gp - matmul(transpose(c(1:m, 1:kt)), c(1:m, 1:kt))
       end if
    end if
    nnull = 0
    do j = 1, kt
       null_cols(j) = .not. (gp(j, j) > tiny(1.0_dp) * 1.0e16_dp)
       ! "twice is enough": a column that a pass has already orthogonalised and normalised, and that loses five digits of its norm
       ! to V AGAIN, lies in span(V) to working precision - what is left of it is rounding noise.  (Corrections confined to the span
       ! of the basis and a few more rows - banded or block-structured operators: t = r / (theta - d) has the support of r - never
       ! leave it however often they are projected and rescaled; the reference's Householder QR completes the basis with arbitrary
       ! orthonormal columns there, src/davidson.f90:197-215, this driver with pseudo-random ones.)
       if (relative_tests .and. pass >= 2 .and. .not. null_cols(j)) null_cols(j) = gp(j, j) < again_rel * g(j, j)
       if (null_cols(j)) nnull = nnull + 1
    end do
    if (nnull > 0) return
    do j = 1, kt
       d(j) = 1.0_dp / sqrt(gp(j, j))
    end do
    do j = 1, kt
       do l = 1, kt
          gp(l, j) = gp(l, j) * d(l) * d(j)
       end do
    end do
    ! deviation of the scaled Gram block from the identity
    dev = 0.0_dp
    do j = 1, kt
       do l = 1, kt
          if (l == j) then
             dev = max(dev, abs(gp(l, j) - 1.0_dp))
          else
             dev = max(dev, abs(gp(l, j)))
          end if
       end do
    end do
    if (pass >= 2 .and. dev * real(kt, dp) < 1.0e-7_dp) then
       ! already orthonormal to ~1e-7: G^(-1/2) = I - E/2 + O(E^2) is exact to rounding, no
       ! eigen-decomposition needed (the usual state of the second pass)
       do j = 1, kt
          do l = 1, kt
             mm(l, j) = -0.5_dp * gp(l, j) * d(l)
          end do
          mm(j, j) = (1.5_dp - 0.5_dp * gp(j, j)) * d(j)
       end do
       wmin = 1.0_dp - dev * real(kt, dp)
       wmax = 1.0_dp + dev * real(kt, dp)
    else
       ! Cholesky route first (CholQR: M = D R^-1 with D G' D = R^T R): a k x k DPOTRF + DTRTRI costs a
       ! fraction of a symmetric eigen-decomposition.  It is accepted only when the factor is well
       ! conditioned (diagonal ratio); otherwise - rank deficiency, clustered corrections - the
       ! eigen-decomposition route (SVQB) with its eigenvalue floor takes over.
       call lapack_cholesky_inverse(gp, u, info)
       chol_ok = .false.
       if (info == 0) then
          wmin = huge(1.0_dp)
          wmax = 0.0_dp
          do j = 1, kt
             wmin = min(wmin, abs(u(j, j)))
             wmax = max(wmax, abs(u(j, j)))
          end do
          chol_ok = wmax < 1.0e4_dp * wmin          ! cond(R) estimate below 1e4 => cond(G') below 1e8
       end if
       if (chol_ok) then
          do j = 1, kt
             do l = 1, kt
                mm(l, j) = d(l) * u(l, j)
             end do
          end do
          ! report the conditioning in the same terms as the eigenvalue route (1/r_jj^2 ~ eigenvalues)
          wmin = 1.0_dp / (wmax * wmax)
          wmax = wmin * 1.0e8_dp
       else
          if (relative_tests .and. pass >= 2) then
             ! ... and the same for columns that depend on EACH OTHER after a pass has already orthonormalised the block: those the
             ! left-to-right factorisation cannot reach (remaining pivot below again_rel) are replaced, not rescaled
             call dependent_columns(gp, kt, again_rel, null_cols, nnull)
             if (nnull > 0) return
          else if (relative_tests .and. ortho_early()) then
             ! the FIRST pass already sees dependence that is exact up to rounding (remaining pivot at the noise level of the Gram
             ! product: the corrections of a banded matrix, docs/history/DESIGN_rounds_1_to_5.md section 0): replaced before the block is swept, instead of a
             ! sweep of noise columns, a second pass that finds them, and a second sweep.  Conservative (dependent_columns: noise)
             do j = 1, kt
                w(j) = 64.0_dp * epsilon(1.0_dp) * g(j, j) * d(j) * d(j)          ! d(j)**2 = 1 / gp(j, j) before the scaling
             end do
             call dependent_columns(gp, kt, first_rel, null_cols, nnull, w)
             if (nnull > 0) return
          end if
          call lapack_rayleigh_ritz(gp, w, u, kt)
          wmax = maxval(w)
          wmin = minval(w)
          do j = 1, kt
             w(j) = max(w(j), floor_rel * wmax)
          end do
          do j = 1, kt
             do l = 1, kt
                mm(l, j) = d(l) * u(l, j) / sqrt(w(j))
             end do
          end do
       end if
    end if
  end subroutine ortho_pass_transform

  !> Structural rank deficiency of a correction block is looked for at the FIRST pass already (it saves a banded matrix the sweep
  !> of its noise columns); DAV_ORTHO_EARLY=0 turns that off (A/B knob).  The first-pass test is conservative (dependent_columns:
  !> noise): before the first pass the columns are not orthogonal to the basis, the projected Gram block G - C^T C carries the
  !> rounding of G at the scale of the UNPROJECTED columns, and a threshold of 1e-13 of the projected norms sits below that noise
  !> for corrections that lie mostly in the span of the basis (generalized problems with a second operator far from the identity:
  !> an unconditional test accepted noise pivots and rejected every column behind them, round after round) - such columns, and
  !> everything behind an ill-conditioned accepted column, are left to the second pass.
  function ortho_early() result(on)
    logical :: on
    integer :: stat, length
    character(len=8) :: buf
    integer, save :: cached = -1
    if (cached < 0) then
       cached = 1
       call get_environment_variable("DAV_ORTHO_EARLY", buf, length, stat)
       if (stat == 0 .and. length > 0) then
          if (buf(1:1) == "0") cached = 0
       end if
    end if
    on = cached == 1
  end function ortho_early

  !> Left-to-right Cholesky of a Gram block with unit diagonal: a column whose remaining pivot - once the accepted columns to its
  !> left are eliminated - falls below thr depends on them to working precision and is skipped (dep, ndep).  Left to right, not
  !> pivoted, because that is the order in which the reference's Householder QR finds its dependent columns: the completion
  !> vectors then land in the same slots.
  subroutine dependent_columns(gs, kt, thr, dep, ndep, noise)
    integer, intent(in) :: kt
    real(dp), intent(in) :: gs(kt, kt), thr
    logical, intent(out) :: dep(kt)
    integer, intent(out) :: ndep
    !> first pass only: noise(j) = rounding level of column j's entries of gs (the projected Gram block carries the rounding of the
    !> unprojected one: eps g_jj / gp_jj).  With it the test is conservative: a column is only called dependent where the block
    !> can tell - its own entries are accurate to thr, and every column accepted before it was accepted with a pivot far above
    !> the noise (>= 1e-6) - everything else is left to the second pass
    real(dp), intent(in), optional :: noise(kt)
    real(dp) :: l(kt, kt), rem
    integer :: i, j, nacc, acc(kt)
    logical :: can_tell
    l = 0.0_dp
    dep = .false.
    ndep = 0
    nacc = 0
    can_tell = .true.
    do j = 1, kt
       ! row j of the factor against the accepted columns
       do i = 1, nacc
          l(j, i) = (gs(j, acc(i)) - dot_product(l(j, 1:i - 1), l(acc(i), 1:i - 1))) / l(acc(i), i)
       end do
       rem = gs(j, j) - dot_product(l(j, 1:nacc), l(j, 1:nacc))
       if (rem >= thr) then
          nacc = nacc + 1
          acc(nacc) = j
          l(j, nacc) = sqrt(rem)
          if (present(noise) .and. rem < 1.0e-6_dp) can_tell = .false.
       else if (.not. present(noise)) then
          dep(j) = .true.
          ndep = ndep + 1
       else if (can_tell .and. noise(j) < thr) then
          dep(j) = .true.
          ndep = ndep + 1
       else
          ! cannot tell: keep the column (with a pivot at the threshold, so that the factor stays finite) and stop judging
          nacc = nacc + 1
          acc(nacc) = j
          l(j, nacc) = sqrt(thr)
          can_tell = .false.
       end if
    end do
  end subroutine dependent_columns

  !> yk (m x kt) <- yk * M with M = G^(-1/2)-like (Cholesky R^-1, or the eigen-decomposition route with an eigenvalue
  !> floor when the factor is ill-conditioned), G = yk^T yk: the columns of the result are Euclidean-orthonormal.
  subroutine restart_transform(yk, m, kt)
    integer, intent(in) :: m, kt
    real(dp), intent(inout) :: yk(m, kt)
    real(dp), allocatable :: g(:, :), u(:, :), w(:), mm(:, :), d(:)
    integer :: j, l, info, pass
    real(dp) :: wmin, wmax
    allocate(g(kt, kt), u(kt, kt), w(kt), mm(kt, kt), d(kt))
    do pass = 1, 2                      ! the second pass sees G = I + O(eps cond): it removes what the first left
       g = lapack_matmul("T", "N", yk, yk)
       do j = 1, kt
          d(j) = 1.0_dp / sqrt(g(j, j))
       end do
       do j = 1, kt
          do l = 1, kt
             g(l, j) = g(l, j) * d(l) * d(j)
          end do
       end do
       call lapack_cholesky_inverse(g, u, info)
       wmin = huge(1.0_dp)
       wmax = 0.0_dp
       if (info == 0) then
          do j = 1, kt
             wmin = min(wmin, abs(u(j, j)))
             wmax = max(wmax, abs(u(j, j)))
          end do
       end if
       if (info == 0 .and. wmax < 1.0e4_dp * wmin) then
          do j = 1, kt
             do l = 1, kt
                mm(l, j) = d(l) * u(l, j)
             end do
          end do
       else
          call lapack_rayleigh_ritz(g, w, u, kt)
          wmax = maxval(w)
          do j = 1, kt
             w(j) = max(w(j), 1.0e-14_dp * wmax)
          end do
          do j = 1, kt
             do l = 1, kt
                mm(l, j) = d(l) * u(l, j) / sqrt(w(j))
             end do
          end do
       end if
       yk = lapack_matmul("N", "N", yk, mm)
    end do
  end subroutine restart_transform

  !> Deterministic filler direction (xorshift), entries in (-0.5, 0.5).
  subroutine pseudo_random_vector(vec, salt)
    real(dp), intent(out) :: vec(:)
    integer, intent(in) :: salt
    integer(c_int64_t) :: s
    integer :: i
    s = 88172645463325252_c_int64_t + int(salt, c_int64_t) * 2654435761_c_int64_t
    do i = 1, size(vec)
       s = ieor(s, shiftl(s, 13))
       s = ieor(s, shiftr(s, 7))
       s = ieor(s, shiftl(s, 17))
       vec(i) = real(shiftr(s, 11), dp) / 9007199254740992.0_dp - 0.5_dp
    end do
  end subroutine pseudo_random_vector

end module davidson_ortho
