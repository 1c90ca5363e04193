!> Host-side helpers of the Davidson driver that touch neither the device nor LAPACK: environment knobs (read where the driver
!> asks, never inside a kernel path), the symmetry probe of the dense front end, the basis-width policy of the reference
!> (src/davidson.f90:195-213), the inner tolerances of the GJD correction solves.
module davidson_knobs
  use, intrinsic :: iso_c_binding
  use numeric_kinds, only: dp
  implicit none
  private
  public :: env_device, env_storage, env_storage_symmetric, symmetry_probe, basis_capacity, gjd_tol_unwanted, gjd_tol_wanted, &
       gjd_adaptive_factor, restart_refresh_interval, ascending_order, tick, trace_iterations, verbose

contains

  !> Device index from the environment (DAVIDSON_DEVICE, default 0): an engine knob that does not
  !> touch the reference's argument lists (dense and matrix-free front ends alike).
  function env_device() result(dev)
    integer :: dev, stat, length
    character(len=16) :: buf
    dev = 0
    call get_environment_variable("DAVIDSON_DEVICE", buf, length, stat)
    if (stat == 0 .and. length > 0) read (buf(1:length), *, iostat=stat) dev
    if (stat /= 0) dev = 0
  end function env_device

  !> DAVIDSON_STORAGE for the dense front end: 1 = "symmetric", 0 = "full", -1 = not set (the front end decides by symmetry_probe)
  function env_storage() result(mode)
    integer :: mode, stat, length
    character(len=16) :: buf
    mode = -1
    call get_environment_variable("DAVIDSON_STORAGE", buf, length, stat)
    if (stat == 0 .and. length >= 3) then
       if (buf(1:3) == "sym") mode = 1
       if (buf(1:3) == "ful") mode = 0
    end if
  end function env_storage

  !> DAVIDSON_STORAGE=symmetric selects symmetric-tiled storage for the dense front end (engines: engine_set_storage)
  function env_storage_symmetric() result(sym)
    logical :: sym
    integer :: stat, length
    character(len=16) :: buf
    call get_environment_variable("DAVIDSON_STORAGE", buf, length, stat)
    sym = (stat == 0 .and. length >= 3)
    if (sym) sym = buf(1:3) == "sym"
  end function env_storage_symmetric

  !> Is the matrix symmetric where it is looked at?  Exact comparison of 8 sampled rows (the first, the last, six spread over the
  !> order) with their columns at up to 512 positions each (the row's neighbourhood of the diagonal excluded: stride over the whole
  !> order) - ~4000 pairs, a fraction of a millisecond whatever the order (a walk along a ROW of a column-major matrix is one cache
  !> miss per entry: whole rows cost 3.5 ms at N=20000, a tenth of the upload they are meant to halve).  A matrix with a single
  !> asymmetric entry outside the sample passes - as it passes the reference, which never looks; what the probe guards against is an
  !> input that is not meant to be symmetric at all.
  function symmetry_probe(matrix) result(symmetric)
    real(dp), dimension(:, :), intent(in) :: matrix
    logical :: symmetric
    integer :: n, k, i, j, nsample, step
    n = size(matrix, 1)
    symmetric = size(matrix, 2) == n
    if (.not. symmetric) return
    nsample = min(n, 8)
    step = max(1, n / 512)
    do k = 0, nsample - 1
       i = 1 + int(int(k, c_int64_t) * int(n - 1, c_int64_t) / int(max(nsample - 1, 1), c_int64_t))
       do j = 1 + mod(k, step), n, step
          if (matrix(i, j) /= matrix(j, i)) then
             symmetric = .false.
             return
          end if
       end do
       ! the two corners of the row: the entries a one-sided (triangular) input would leave different
       if (matrix(i, 1) /= matrix(1, i) .or. matrix(i, n) /= matrix(n, i)) then
          symmetric = .false.
          return
       end if
    end do
  end function symmetry_probe

  !> Widest basis the reference's policy can reach: m starts at 2*lowest and doubles while
  !> m <= max_dim (src/davidson.f90:195-213), so it may overshoot max_dim once.
  pure function basis_capacity(lowest, max_dim) result(cap)
    integer, intent(in) :: lowest, max_dim
    integer :: cap
    cap = 2 * lowest
    do while (cap <= max_dim)
       cap = 2 * cap
    end do
  end function basis_capacity

  !> Inner tolerance of the GJD solves for the Ritz pairs beyond `lowest`: 1e-2 relative (DAV_GJD_TOL_UNWANTED
  !> overrides; 1e-4 until round 4).  Their corrections only enrich the search space; with 1e-6, 1e-4, 1e-2 and 1e-1
  !> every golden GJD case keeps the reference's outer iteration count while the block sweeps of A drop by a third
  !> to a half (N=40000 generalized: 62 -> 45 / 39 / 33 sweeps), and over a grid of 108 problems against the oracle's
  !> exact solves (tests/gjd_policy_sweep.py: orders 150-500, lowest 2-8, sparsity 1e-3 - 5e-2, standard
  !> and generalized) 1e-2 gives the iteration counts of 1e-4 in every case.  These pairs sit in the interior of the
  !> projected spectrum, where MINRES on A - theta B converges slowest: at configs[3] they kept the inner solve
  !> going for 13 of 18 steps after the wanted pairs had finished.  Hence the sign: a NEGATIVE tolerance (the default,
  !> -1e-2) makes these pairs followers (dav_gjd_correction_n) - they stop at |t| or when every wanted pair has
  !> stopped, whichever comes first: they get the inner steps the wanted pairs need, not a solve of their own
  !> (144-problem sweep against the oracle: never more outer iterations than the reference's exact solves).
  function gjd_tol_unwanted() result(t)
    real(dp) :: t
    integer :: stat, length
    character(len=32) :: buf
    t = -1.0e-2_dp
    call get_environment_variable("DAV_GJD_TOL_UNWANTED", buf, length, stat)
    if (stat == 0 .and. length > 0) read (buf(1:length), *, iostat=stat) t
    if (stat /= 0 .or. t == 0.0_dp) t = -1.0e-2_dp
  end function gjd_tol_unwanted

  !> Inner tolerance of the GJD solve for a WANTED pair whose residual norm is `err`: the correction equation is solved
  !> only as far as the outer iteration can use it.  An exact solve (the reference's DSYSV) takes the residual from
  !> err to ~err**2; an inexact one with relative tolerance tau to ~max(err**2, tau*err).  tau = c * tolerance / err
  !> therefore leaves c * tolerance on top of what the exact solve reaches: where the reference converges (err**2 below
  !> the tolerance) so does this, where it does not, the next residual is the reference's to within c * tolerance.
  !> c = 0.01 (gjd_adaptive_factor), tau clipped to [1e-10, 1e-2].
  function gjd_tol_wanted(err, tolerance, c) result(t)
    real(dp), intent(in) :: err, tolerance, c
    real(dp) :: t
    t = 1.0e-10_dp
    if (c > 0.0_dp .and. err > 0.0_dp) t = min(1.0e-2_dp, max(1.0e-10_dp, c * tolerance / err))
    ! (A forcing term on top - no more accurate than c2 * err, because an exact solve "only" leaves ~err**2 - was measured and
    ! dropped: these matrices converge faster than that estimate, and with c2 = 1e-3 already 11 of 144 problems need an outer
    ! iteration more than the reference; profiles/experiments/r04_gjd_policy_sweep3.log.)
  end function gjd_tol_wanted

  !> c of gjd_tol_wanted: 0.01; DAV_GJD_ADAPTIVE overrides (0 = the fixed 1e-10 of round 3).  Read once per solve, before the loop.
  function gjd_adaptive_factor() result(c)
    real(dp) :: c
    integer :: stat, length
    character(len=32) :: buf
    c = 0.01_dp
    call get_environment_variable("DAV_GJD_ADAPTIVE", buf, length, stat)
    if (stat == 0 .and. length > 0) then
       read (buf(1:length), *, iostat=stat) c
       if (stat /= 0) c = 0.01_dp
    end if
  end function gjd_adaptive_factor

  !> After how many collapse restarts W = A*V (and B*V) of the kept block are recomputed instead of contracted (see the restart
  !> branch of the loop): 8; DAV_REFRESH_EVERY overrides (1 = after every restart, as the reference does).
  function restart_refresh_interval() result(k)
    integer :: k, stat, length
    character(len=16) :: buf
    k = 8
    call get_environment_variable("DAV_REFRESH_EVERY", buf, length, stat)
    if (stat == 0 .and. length > 0) then
       read (buf(1:length), *, iostat=stat) k
       if (stat /= 0 .or. k < 1) k = 8
    end if
  end function restart_refresh_interval

  !> order(k) = index of the k-th smallest entry (stable insertion sort: a handful of eigenvalues)
  subroutine ascending_order(x, order)
    real(dp), intent(in) :: x(:)
    integer, intent(out) :: order(size(x))
    integer :: a, b, t
    do a = 1, size(x)
       order(a) = a
    end do
    do a = 2, size(x)
       t = order(a)
       b = a - 1
       do while (b >= 1)
          if (x(order(b)) <= x(t)) exit
          order(b + 1) = order(b)
          b = b - 1
       end do
       order(b + 1) = t
    end do
  end subroutine ascending_order

  function tick() result(t)
    real(dp) :: t
    integer(c_int64_t) :: count, rate
    call system_clock(count, rate)
    t = real(count, dp) / real(rate, dp)
  end function tick

  !> DAVIDSON_VERBOSE=2 (or more): one line per outer iteration (basis width, residual norms of the wanted pairs)
  function trace_iterations() result(on)
    logical :: on
    integer :: stat, length, level
    character(len=8) :: buf
    on = .false.
    call get_environment_variable("DAVIDSON_VERBOSE", buf, length, stat)
    if (stat == 0 .and. length > 0) then
       read (buf(1:length), *, iostat=stat) level
       on = (stat == 0 .and. level >= 2)
    end if
  end function trace_iterations

  function verbose() result(on)
    logical :: on
    integer :: stat, length
    character(len=8) :: buf
    call get_environment_variable("DAVIDSON_VERBOSE", buf, length, stat)
    on = (stat == 0 .and. length > 0)
  end function verbose

end module davidson_knobs
