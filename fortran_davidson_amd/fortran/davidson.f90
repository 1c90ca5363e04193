!> MI355X-native block Davidson eigensolver behind the Fortran API of NLESC-JCER/Fortran_Davidson.
!>
!> Public surface (drop-in): module `davidson` with the generic `generalized_eigensolver`
!> (reference: src/davidson.f90:599-625) whose dense and matrix-free specifics keep the reference's
!> argument lists (src/davidson.f90:51-52, :277-278), plus `free_matmul` (src/davidson.f90:526).
!> A third specific takes a `davidson_engine` handle so that a matrix already resident in HBM (or a
!> built-in device operator) can be solved repeatedly without re-uploading.
!>
!> Design: the host keeps the control flow of the reference's outer loop (src/davidson.f90:138-229)
!> and the m x m Rayleigh-Ritz problem (lapack_wrapper); everything N-long lives in HBM behind the
!> C ABI of include/davidson_hip.h.  Per iteration the device does ONE block sweep of A (the new
!> basis columns only) instead of the reference's (m+1) sweeps, residues come from the cached A*V
!> panel, and Householder QR of the whole basis is replaced by a block Gram-Schmidt of the new
!> columns (same span, hence the same Ritz values, residual norms and iteration counts).

module davidson_device
  use, intrinsic :: iso_c_binding
  use numeric_kinds, only: dp
  use davidson_hip_c
  use lapack_wrapper, only: lapack_rayleigh_ritz, lapack_matmul
  use davidson_knobs
  use davidson_ortho
  use davidson_engine_setup
  implicit none
  private
  public :: POLICY_ALL, POLICY_UNCONVERGED, POLICY_LOCKING
  public :: davidson_engine, engine_create, engine_destroy, engine_set_dense, engine_set_storage, env_device, env_storage_symmetric, env_storage, symmetry_probe, fits_as_full_rows, engine_set_device_rr, engine_set_inner_precision, &
       engine_read_matrix, engine_dense_begin, engine_dense_put_rows, engine_dense_end, &
       engine_set_correction_policy, &
       engine_generate_diagonal_dominant, engine_set_hashed_operator, engine_set_harness_operator, &
       engine_set_identity, engine_set_device_operator, engine_comm_unique_id, engine_comm_init, &
       generalized_eigensolver_device, davidson_device_loop, basis_capacity, davidson_free_buffers

  !> The N-long side of the block orthonormalisation (davidson_ortho: ortho_backend) on the engine's panels: K2 Gram launches, K3
  !> block updates, replacement columns written into the basis panel
  type, extends(ortho_backend) :: device_ortho
     type(c_ptr) :: h = c_null_ptr
     integer :: n = 0
   contains
     procedure :: gram => device_ortho_gram
     procedure :: apply => device_ortho_apply
     procedure :: unit_column => device_ortho_unit_column
     procedure :: put_column => device_ortho_put_column
  end type device_ortho

  abstract interface
     function block_operator(input_vect) result(output_vect)
       import :: dp
       real(dp), dimension(:, :), intent(in) :: input_vect
       real(dp), dimension(size(input_vect, 1), size(input_vect, 2)) :: output_vect
     end function block_operator
  end interface

contains

  !> Solve with the operators already resident behind `eng` (third specific of the generic).
  !> Argument meaning as generalized_eigensolver_dense; `eigenvectors` is optional so that a
  !> benchmark can leave the Ritz vectors on the device.
  subroutine generalized_eigensolver_device(eng, eigenvalues, eigenvectors, lowest, method, max_iterations, &
       tolerance, iters, max_dim_sub)
    type(davidson_engine), intent(inout) :: eng
    integer, intent(in) :: lowest
    real(dp), dimension(lowest), intent(out) :: eigenvalues
    real(dp), dimension(:, :), intent(out), optional :: eigenvectors
    character(len=*), intent(in) :: method
    integer, intent(in) :: max_iterations
    real(dp), intent(in) :: tolerance
    integer, intent(out) :: iters
    integer, intent(in), optional :: max_dim_sub
    integer :: max_dim
    max_dim = 10 * lowest
    if (present(max_dim_sub)) max_dim = max_dim_sub
    if (basis_capacity(lowest, max_dim) > eng%max_cols) then
       print *, "generalized_eigensolver: engine created for a narrower basis than lowest/max_dim_sub need"
       error stop
    end if
    ! stored matrix: the dense driver's sticky convergence flags (src/davidson.f90:176); matrix-free operator A:
    ! the matrix-free driver's all-at-once test (:416)
    call davidson_device_loop(eng%h, eng%n, lowest, method, max_iterations, tolerance, iters, max_dim, &
         eng%gev, .not. eng%free_semantics, eigenvalues, policy=eng%policy, device_rr=eng%device_rr, &
         phase_seconds=eng%phase_seconds)
    if (present(eigenvectors)) then
       call check_dav(dav_panel_get(eng%h, DAV_PANEL_X, 0_c_int, int(lowest, c_int), eigenvectors, &
            int(size(eigenvectors, 1), c_int64_t)), "dav_panel_get")
    end if
  end subroutine generalized_eigensolver_device

  !> The outer loop (control flow of src/davidson.f90:138-229 / :375-441) on a prepared engine.
  !> sticky = .true. reproduces the dense path's sticky convergence flags (:176), .false. the
  !> matrix-free path's all-at-once test (:416).  fun_a/fun_b present = operators applied by the
  !> host through callbacks (API-faithful matrix-free path).
  subroutine davidson_device_loop(h, n, lowest, method, max_iterations, tolerance, iters, max_dim, gev, &
       sticky, eigenvalues, fun_a, fun_b, policy, device_rr, phase_seconds)
    type(c_ptr), intent(in) :: h
    integer, intent(in) :: n, lowest, max_iterations, max_dim
    character(len=*), intent(in) :: method
    real(dp), intent(in) :: tolerance
    integer, intent(out) :: iters
    logical, intent(in) :: gev, sticky
    real(dp), dimension(lowest), intent(out) :: eigenvalues
    procedure(block_operator), optional :: fun_a, fun_b
    !> POLICY_ALL (default) = the reference: one correction per basis vector, the basis doubles.
    !> POLICY_UNCONVERGED (opt-in, changes iters): corrections only for those of the `lowest` wanted pairs
    !> whose residual is still above the tolerance; all-at-once convergence test.
    integer, intent(in), optional :: policy
    !> Rayleigh-Ritz on the device (engine_set_device_rr); bases wider than 128 columns use the host
    logical, intent(in), optional :: device_rr
    !> wall time of this solve by phase (see davidson_engine%phase_seconds)
    real(dp), intent(out), optional :: phase_seconds(8)

    integer :: m, kt, i, j, cap, initial_dimension, meth, inner, phase, pol, ncorr, nvec, nrestart, opass
    integer :: refresh_every
    integer(c_int), allocatable :: sel(:)
    integer(c_int) :: sweeps
    logical :: drr
    real(dp), allocatable :: theta_sel(:), tols(:)
    real(dp), allocatable, target :: c_pre(:, :), g_pre(:, :)
    logical :: have_pre
    logical :: expand_now
    integer(c_int64_t) :: ld
    integer(c_int64_t), allocatable :: idx(:)
    real(dp), allocatable :: hm(:, :), sm(:, :), theta(:), y(:, :), errors(:)
    logical, allocatable :: has_converged(:)
    logical :: host_ops, done, lazy_x, have_all_pairs
    real(dp) :: t0, phase_s(8), tol_unwanted, adaptive_c
    type(device_ortho) :: dev

    phase_s = 0.0_dp
    dev%h = h
    dev%n = n
    host_ops = present(fun_a)
    if (host_ops) then
       meth = DAV_METHOD_DPR       ! the matrix-free driver never looks at `method`: always DPR (src/davidson.f90:428)
    else
       select case (trim(method))
       case ("DPR")
          meth = DAV_METHOD_DPR
       case ("GJD")
          meth = DAV_METHOD_GJD
       case default
          ! the reference leaves the correction undefined here (src/davidson.f90:656-669)
          print *, "generalized_eigensolver: unknown correction method '", trim(method), "' (DPR or GJD)"
          error stop
       end select
    end if
    pol = POLICY_ALL
    if (present(policy)) pol = policy

    initial_dimension = 2 * lowest
    cap = basis_capacity(lowest, max_dim)
    ld = int(cap, c_int64_t)
    allocate(hm(cap, cap), sm(cap, cap), idx(initial_dimension), errors(lowest), has_converged(lowest))
    hm = 0.0_dp
    sm = 0.0_dp
    has_converged = .false.
    refresh_every = restart_refresh_interval()
    tol_unwanted = gjd_tol_unwanted()          ! environment knobs of the GJD inner solves: read once, here
    adaptive_c = gjd_adaptive_factor()
    nrestart = 0
    drr = .false.
    if (present(device_rr)) drr = device_rr .and. cap <= 128
    call check_dav(dav_rr_enable(h, merge(1_c_int, 0_c_int, drr)), "dav_rr_enable")
    ! The Ritz vectors are what the caller gets back at the end (src/davidson.f90:186-187); the DPR correction never reads them:
    ! they are made once, when the loop stops (finish_ritz_vectors), not in every iteration.  GJD needs x_k for its correction
    ! solves and the device-side Rayleigh-Ritz keeps Y on the device: both compute X in the Ritz phase as before.
    lazy_x = (.not. drr) .and. meth /= DAV_METHOD_GJD
    call check_dav(dav_set_lazy_ritz_vectors(h, merge(1_c_int, 0_c_int, lazy_x)), "dav_set_lazy_ritz_vectors")

    ! Several ranks: everything the loop decides follows from these inputs and from all-reduced (hence bitwise identical) numbers.
    ! The inputs are verified in a collective whose size does not depend on them (round-5 advisor: the per-iteration words of
    ! dav_agree_next ride on all-reduces whose element count they determine - ranks that differ in `lowest` or `max_dim` would enter
    ! those with different counts) whenever they differ from what the engine verified last: the first solve always, repeated solves
    ! with the same inputs never (dav_agree_inputs); a single rank: a no-op
    call check_dav(dav_agree_inputs(h, [real(n, dp), real(lowest, dp), real(max_dim, dp), real(max_iterations, dp), tolerance, &
         real(pol, dp), real(meth, dp), merge(1.0_dp, 0.0_dp, sticky), merge(1.0_dp, 0.0_dp, gev)], 9_c_int), "dav_agree_inputs")

    ! 1. initial basis: unit vectors at the lowest diagonal entries; W0 = A*V0
    m = initial_dimension
    t0 = tick()
    call check_dav(dav_init_basis(h, int(m, c_int), idx), "dav_init_basis")
    if (host_ops) call apply_host_block(h, n, 0, m, fun_a, fun_b)
    ! 2. projected matrices
    if (drr) then
       call check_dav(dav_project_dev(h, 0_c_int, int(m, c_int)), "dav_project")
    else
       call check_dav(dav_project(h, 0_c_int, int(m, c_int), hm, ld, sm, ld), "dav_project")
    end if
    call lap(phase_s, t0, 1)

    iters = max_iterations + 1
    done = .false.
    if (pol == POLICY_LOCKING) then
       if (host_ops .or. drr) then
          print *, "generalized_eigensolver: the 'locking' policy serves device operators with the Rayleigh-Ritz problem on the host"
          error stop
       end if
       call davidson_locking_loop(h, n, lowest, meth, pol, max_iterations, tolerance, max_dim, cap, gev, hm, sm, ld, m, dev, refresh_every, &
            adaptive_c, eigenvalues, iters, done, phase_s, t0)
    end if
    outer_loop: do i = 1, merge(0, max_iterations, pol == POLICY_LOCKING)
       ! 3. Rayleigh-Ritz on the host (the only LAPACK call on the path)
       if (allocated(theta)) deallocate(theta, y)
       allocate(theta(m), y(m, m))
       ! Does this iteration grow the basis or end in a collapse restart?  (known before the Ritz problem is
       ! solved: it only depends on the width.)  The reference's policy corrects every Ritz pair, so a growing
       ! iteration needs all m eigenpairs of the projected problem; a restarting one only the 2*lowest it
       ! keeps, and the opt-in policy never more than that.
       if (pol == POLICY_ALL) then
          expand_now = m <= max_dim
          ncorr = merge(m, lowest, expand_now)
       else
          ! grow while the widest possible block still fits; a basis that has just been (re)started always grows -
          ! as long as the projected matrices (cap x cap) can take one more column
          expand_now = ((m + lowest <= max_dim) .or. (m <= initial_dimension)) .and. m < cap
          ncorr = lowest
       end if
       ! several ranks: the decisions of this iteration must be the same everywhere.  They are functions of these words and of
       ! the all-reduced (bitwise identical) residual norms; the words ride on the all-reduce of the Ritz phase below (no
       ! collective of their own), and a rank that differs stops with a message instead of hanging its peers
       call check_dav(dav_agree_next(h, [real(i, dp), real(m, dp), merge(1.0_dp, 0.0_dp, expand_now), real(ncorr, dp), &
            tolerance, real(count(has_converged), dp), real(pol, dp), real(meth, dp)], 8_c_int), "dav_agree_next")
       ! Convergence usually arrives at the widest basis, exactly where the full Ritz problem is dearest
       ! (order 64: 150-200 us on the host) although a converged iteration only needs the `lowest` wanted pairs.
       ! When the previous residues say convergence is near, solve for those pairs first (MRRR on a subset) and
       ! test them; only if the test fails is the full problem solved.  Same Ritz pairs, same iteration count.
       have_all_pairs = .false.
       if (.not. drr .and. pol == POLICY_ALL .and. expand_now .and. i > 1 .and. m >= 48 .and. 2 * lowest <= m) then
          if (maxval(errors) < sqrt(tolerance)) then
             ! (lapack_rayleigh_ritz solves for ALL pairs when the wanted ones are more than an eighth of the order: theta and y are
             ! then complete, and the full problem below is not solved a second time - round-4 advisor)
             have_all_pairs = 8 * lowest > m
             if (gev) then
                call lapack_rayleigh_ritz(hm(1:m, 1:m), theta, y, lowest, sm(1:m, 1:m))
             else
                call lapack_rayleigh_ritz(hm(1:m, 1:m), theta, y, lowest)
             end if
             call lap(phase_s, t0, 2)
             call check_dav(dav_ritz_residual_correction_n(h, int(m, c_int), int(lowest, c_int), int(lowest, c_int), y, &
                  int(m, c_int64_t), theta, DAV_METHOD_NONE, errors), "dav_ritz_residual_correction")
             call lap(phase_s, t0, 3)
             if (sticky) then
                do j = 1, lowest
                   if (errors(j) < tolerance) has_converged(j) = .true.
                end do
                done = all(has_converged)
             else
                done = all(errors < tolerance)
             end if
             if (done) then
                eigenvalues = theta(1:lowest)
                iters = i
                call finish_ritz_vectors()
                exit outer_loop
             end if
          end if
       end if
       nvec = min(m, initial_dimension)
       if (pol == POLICY_ALL .and. expand_now) nvec = m
       if (.not. drr .and. .not. have_all_pairs) then
          if (gev) then
             call lapack_rayleigh_ritz(hm(1:m, 1:m), theta, y, nvec, sm(1:m, 1:m))
          else
             call lapack_rayleigh_ritz(hm(1:m, 1:m), theta, y, nvec)
          end if
       end if
       call lap(phase_s, t0, 2)

       ! 4. Ritz vectors, residues, their norms and the DPR correction - one fused device phase
       phase = meth
       if (.not. expand_now) phase = DAV_METHOD_NONE      ! this iteration ends in a restart: no correction block
       have_pre = .false.
       if (drr) then
          ! 3 + 4 on the device: eigenpairs of the device-resident projected matrices, then the same fused phase from
          ! where they lie; one synchronisation brings back the Ritz values, the residual norms and the Gram blocks
          if (phase == DAV_METHOD_DPR) then
             if (allocated(c_pre)) deallocate(c_pre, g_pre)
             allocate(c_pre(m, ncorr), g_pre(ncorr, ncorr))
             call check_dav(dav_rr_ritz(h, int(m, c_int), int(ncorr, c_int), int(lowest, c_int), int(phase, c_int), theta, &
                  errors, c_loc(c_pre), int(m, c_int64_t), c_loc(g_pre), int(ncorr, c_int64_t), sweeps), "dav_rr_ritz")
             have_pre = .true.
          else
             call check_dav(dav_rr_ritz(h, int(m, c_int), int(ncorr, c_int), int(lowest, c_int), int(phase, c_int), theta, &
                  errors, c_null_ptr, 0_c_int64_t, c_null_ptr, 0_c_int64_t, sweeps), "dav_rr_ritz")
          end if
       else if (phase == DAV_METHOD_DPR) then
          ! the Gram blocks of the first orthonormalisation pass come back with the residual norms
          if (allocated(c_pre)) deallocate(c_pre, g_pre)
          allocate(c_pre(m, ncorr), g_pre(ncorr, ncorr))
          call check_dav(dav_ritz_residual_correction_g(h, int(m, c_int), int(ncorr, c_int), int(lowest, c_int), y, &
               int(m, c_int64_t), theta, errors, c_pre, int(m, c_int64_t), g_pre, int(ncorr, c_int64_t)), &
               "dav_ritz_residual_correction")
          have_pre = .true.
       else
          call check_dav(dav_ritz_residual_correction_n(h, int(m, c_int), int(ncorr, c_int), int(lowest, c_int), y, &
               int(m, c_int64_t), theta, int(phase, c_int), errors), "dav_ritz_residual_correction")
       end if
       call lap(phase_s, t0, 3)
       eigenvalues = theta(1:lowest)
       if (trace_iterations()) print "(a, i0, a, i0, a, es10.3, a, es10.3, a, i0)", "davidson trace: iteration ", i, " m=", m, &
            " max residual ", maxval(errors(1:lowest)), " min ", minval(errors(1:lowest)), " below tolerance ", &
            count(errors(1:lowest) < tolerance)
       if (sticky .and. pol == POLICY_ALL) then
          do j = 1, lowest
             if (errors(j) < tolerance) has_converged(j) = .true.
          end do
          done = all(has_converged)
       else
          done = all(errors < tolerance)
       end if
       if (done .or. i == max_iterations) call finish_ritz_vectors()     ! the last Ritz pairs are what a non-converged solve returns
       if (done) then
          iters = i
          exit outer_loop
       end if

       if (expand_now) then
          ! 5. correction block T -> orthonormalise against V and itself -> new basis columns
          if (pol == POLICY_ALL) then
             ! one correction per basis vector (the reference) - but never more columns than the space has
             ! left: with m + kt > n the reference stops in DORGQR (src/lapack_wrapper.f90:176-236); here the
             ! leading n - m corrections complete the basis and the next Ritz problem is exact
             kt = min(m, n - m)
             if (kt <= 0) then                           ! V already spans everything (unreachable: exact Ritz pairs converge)
                call finish_ritz_vectors()
                exit outer_loop
             end if
             if (meth == DAV_METHOD_GJD) then
                ! the wanted pairs are solved to 1e-10 (what makes the outer iteration count equal to the
                ! reference's exact solves); the other m - lowest corrections only enrich the basis
                if (allocated(tols)) deallocate(tols)
                allocate(tols(kt))
                tols = tol_unwanted
                do j = 1, min(lowest, kt)
                   tols(j) = gjd_tol_wanted(errors(j), tolerance, adaptive_c)
                end do
                call check_dav(dav_gjd_correction_n(h, int(m, c_int), int(kt, c_int), theta, 300_c_int, 1.0e-10_dp, &
                     tols, inner), "dav_gjd_correction")
                call lap(phase_s, t0, 8)
             end if
          else
             ! only the wanted pairs that have not converged: keep their columns, drop the others
             kt = min(count(errors >= tolerance), n - m, cap - m)   ! cap - m: max_dim_sub < 2*lowest leaves cap = 2*lowest
             if (kt <= 0) then
                call finish_ritz_vectors()
                exit outer_loop
             end if
             if (allocated(sel)) deallocate(sel, theta_sel)
             allocate(sel(kt), theta_sel(kt))
             kt = 0
             do j = 1, lowest
                if (errors(j) >= tolerance .and. kt < size(sel)) then
                   kt = kt + 1
                   sel(kt) = int(j - 1, c_int)
                   theta_sel(kt) = theta(j)
                end if
             end do
             if (meth == DAV_METHOD_GJD) then
                call check_dav(dav_panel_select(h, DAV_PANEL_X, 0_c_int, int(kt, c_int), sel), "dav_panel_select")
                call check_dav(dav_panel_select(h, DAV_PANEL_R, 0_c_int, int(kt, c_int), sel), "dav_panel_select")
                if (allocated(tols)) deallocate(tols)
                allocate(tols(kt))
                do j = 1, kt
                   tols(j) = gjd_tol_wanted(errors(sel(j) + 1), tolerance, adaptive_c)
                end do
                call check_dav(dav_gjd_correction_n(h, int(m, c_int), int(kt, c_int), theta_sel, 300_c_int, &
                     1.0e-10_dp, tols, inner), "dav_gjd_correction")
                call lap(phase_s, t0, 8)
             else
                call check_dav(dav_panel_select(h, DAV_PANEL_V, int(m, c_int), int(kt, c_int), sel), "dav_panel_select")
                if (have_pre) then                       ! keep the same columns of the prefetched Gram blocks
                   c_pre(:, 1:kt) = c_pre(:, sel(1:kt) + 1)
                   g_pre(1:kt, 1:kt) = g_pre(sel(1:kt) + 1, sel(1:kt) + 1)
                end if
             end if
          end if
          if (drr) then
             ! device-resident projected matrices: all passes first, then the sweep and the projection on the device
             if (have_pre) then
                call block_orthonormalise(dev, n, m, kt, c_pre(:, 1:kt), g_pre(1:kt, 1:kt))
             else
                call block_orthonormalise(dev, n, m, kt)
             end if
             call lap(phase_s, t0, 4)
             call check_dav(dav_expand(h, int(m, c_int), int(kt, c_int)), "dav_expand")
             call lap(phase_s, t0, 5)
             call check_dav(dav_project_dev(h, int(m, c_int), int(kt, c_int)), "dav_project")
             call lap(phase_s, t0, 6)
          else
             ! first Gram-Schmidt pass only (the block is then orthonormal to ~1e-8) ...
             if (have_pre) then
                call block_orthonormalise(dev, n, m, kt, c_pre(:, 1:kt), g_pre(1:kt, 1:kt), only_first=.true., last_pass=opass)
             else
                call block_orthonormalise(dev, n, m, kt, only_first=.true., last_pass=opass)
             end if
             call lap(phase_s, t0, 4)
             ! 6. ... one block sweep of A over the new columns as that pass left them ...
             call check_dav(dav_expand(h, int(m, c_int), int(kt, c_int)), "dav_expand")
             if (host_ops) call apply_host_block(h, n, m, kt, fun_a, fun_b)
             call lap(phase_s, t0, 5)
             ! ... and the last pass together with the projection: one reduction, one round trip (the pass is linear, so the
             ! images A*T and B*T follow T, and the new rows / columns of H (and S) follow on the host)
             call project_with_last_pass(h, n, m, kt, gev, hm, sm, ld, dev, opass + 1, fun_a, fun_b)
             call lap(phase_s, t0, 6)
          end if
          m = m + kt
       else
          ! collapse restart: V <- V*Y(:, 1:2L) (src/davidson.f90:218).  The reference then re-applies A (and B) to the
          ! whole basis (:223-226); here W = A*V and B*V are contracted with the same columns instead - W*Y = A*(V*Y)
          ! to rounding - so a restart costs three panel products and no sweep of the operators.
          kt = initial_dimension
          if (drr .and. .not. gev) then
             call check_dav(dav_rr_restart(h, int(m, c_int), int(kt, c_int)), "dav_rr_restart")
          else
             if (drr) call check_dav(dav_rr_get(h, int(m, c_int), int(m, c_int), theta, y, int(m, c_int64_t)), "dav_rr_get")
             if (gev) then
                ! Y is S-orthonormal in the generalized case: V*Y(:, 1:kt) has the Gram matrix G = Y^T Y (V itself is
                ! Euclidean-orthonormal), so Y*M with M^T G M = I makes the restarted basis orthonormal again - what the
                ! reference gets from its next QR of the whole basis - and carries W and B*V along
                call restart_transform(y(1:m, 1:kt), m, kt)
             end if
             call check_dav(dav_restart(h, int(m, c_int), int(kt, c_int), y, int(m, c_int64_t)), "dav_restart")
          end if
          ! W = A*V and B*V were contracted, not recomputed: each restart adds ~eps*|A|*sqrt(m) to W - A*V.  Every
          ! `refresh_every`-th restart re-applies the operators to the kept block, as the reference does after every
          ! restart (src/davidson.f90:223-226), so the drift of the residuals the convergence test reads stays bounded
          ! over long runs (max_iterations = 1000)
          nrestart = nrestart + 1
          if (mod(nrestart, refresh_every) == 0) then
             call check_dav(dav_expand(h, 0_c_int, int(kt, c_int)), "dav_expand")
             if (host_ops) call apply_host_block(h, n, 0, kt, fun_a, fun_b)
          end if
          call lap(phase_s, t0, 7)
          hm = 0.0_dp
          sm = 0.0_dp
          if (drr) then
             call check_dav(dav_project_dev(h, 0_c_int, int(kt, c_int)), "dav_project")
          else
             call check_dav(dav_project(h, 0_c_int, int(kt, c_int), hm, ld, sm, ld), "dav_project")
          end if
          call lap(phase_s, t0, 6)
          m = kt
       end if
    end do outer_loop

    if (.not. done) then
       iters = max_iterations + 1
       print *, "Warning: Algorithm did not converge!!"
    end if
    if (verbose()) then
       print "(a, i0, a, i0, a, 8f9.3)", "davidson: n=", n, " iters=", iters, &
            " ms[setup rr ritz ortho apply project restart gjd]=", phase_s * 1.0e3_dp
    end if
    if (present(phase_seconds)) phase_seconds = phase_s
    call check_dav(dav_set_lazy_ritz_vectors(h, 0_c_int), "dav_set_lazy_ritz_vectors")

  contains



    !> X(:, 1:lowest) = V(:, 1:m) * Y(:, 1:lowest) for the Ritz pairs of this iteration (see lazy_x above)
    subroutine finish_ritz_vectors()
      if (lazy_x) call check_dav(dav_ritz_vectors(h, int(m, c_int), int(lowest, c_int), y, int(size(y, 1), c_int64_t)), &
           "dav_ritz_vectors")
    end subroutine finish_ritz_vectors


  end subroutine davidson_device_loop

  !> wall time since the last lap goes to phase `slot` (davidson_engine%phase_seconds)
  subroutine lap(phase_s, t0, slot)
    real(dp), intent(inout) :: phase_s(8), t0
    integer, intent(in) :: slot
    real(dp) :: t1
    t1 = tick()
    phase_s(slot) = phase_s(slot) + (t1 - t0)
    t0 = t1
  end subroutine lap

  !> The last orthonormalisation pass of the block V(:, m+1:m+kt) (already swept: W, B*V hold its images) fused with the
  !> projection: dav_project_ortho returns [V T]^T (A T), [V T]^T (B T), C = V^T T and G = T^T T in one fetch; the pass
  !> T <- (T - V C) M is applied to T and its images (dav_ortho_apply_all) and to the projected blocks here:
  !>   V^T A T'' = (Hv - H C) M,   T''^T A T'' = M^T (Ht - C^T Hv - Hv^T C + C^T H C) M   (A symmetric, as everywhere).
  !> A pass that does not leave the block clean (rank-deficient corrections: rare) falls back to the separate passes, a second
  !> sweep of the block and dav_project.
  subroutine project_with_last_pass(h, n, m, kt, gev, hm, sm, ld, dev, first_pass, fun_a, fun_b)
    type(c_ptr), intent(in) :: h
    integer, intent(in) :: n, m, kt
    logical, intent(in) :: gev
    real(dp), intent(inout), contiguous :: hm(:, :), sm(:, :)   !< the projected matrices (cap x cap): their new rows / columns are filled here
    integer(c_int64_t), intent(in) :: ld
    type(device_ortho), intent(inout) :: dev
    integer, intent(in) :: first_pass
    procedure(block_operator), optional :: fun_a, fun_b     !< host callbacks (the matrix-free front end): the fallback path re-applies them
    logical :: host_ops
    integer, parameter :: max_pass = 8
    real(dp), allocatable, target :: hraw(:, :), sraw(:, :)
    real(dp), allocatable :: c2(:, :), g2(:, :), mm(:, :)
    logical, allocatable :: null_cols(:)
    real(dp) :: wmin, wmax
    integer :: nnull, p, pass
    host_ops = present(fun_a)
    p = m + kt
    allocate(hraw(p, kt), c2(max(m, 1), kt), g2(kt, kt), mm(kt, kt), null_cols(kt))
    if (gev) then
       allocate(sraw(p, kt))
       call check_dav(dav_project_ortho(h, int(m, c_int), int(kt, c_int), hraw, int(p, c_int64_t), c_loc(sraw), &
            int(p, c_int64_t), c2, int(max(m, 1), c_int64_t), g2, int(kt, c_int64_t)), "dav_project_ortho")
    else
       call check_dav(dav_project_ortho(h, int(m, c_int), int(kt, c_int), hraw, int(p, c_int64_t), c_null_ptr, &
            0_c_int64_t, c2, int(max(m, 1), c_int64_t), g2, int(kt, c_int64_t)), "dav_project_ortho")
    end if
    pass = first_pass
    do
       call ortho_pass_transform(pass, m, kt, c2, g2, mm, wmin, wmax, null_cols, nnull)
       if (nnull > 0) then
          ! numerically null columns (rank-deficient corrections: rare): they are replaced by fresh directions, which have no
          ! images yet - separate passes, a second sweep of the block and dav_project
          call block_orthonormalise(dev, n, m, kt, first_pass=pass)
          call check_dav(dav_expand(h, int(m, c_int), int(kt, c_int)), "dav_expand")
          if (host_ops) call apply_host_block(h, n, m, kt, fun_a, fun_b)
          call check_dav(dav_project(h, int(m, c_int), int(kt, c_int), hm, ld, sm, ld), "dav_project")
          return
       end if
       call check_dav(dav_ortho_apply_all(h, int(m, c_int), int(kt, c_int), c2, int(max(m, 1), c_int64_t), mm, &
            int(kt, c_int64_t)), "dav_ortho_apply_all")
       call transform_projected(hm, hraw, c2, mm, m, kt)
       if (gev) call transform_projected(sm, sraw, c2, mm, m, kt)
       if (pass >= 2 .and. wmin > 0.5_dp .and. wmax < 2.0_dp) exit          ! orthonormal to rounding
       if (pass >= max_pass) then
          print *, "Warning: block orthonormalisation did not settle in ", max_pass, " passes"
          exit
       end if
       ! one more pass (a block whose second pass still found it ill-conditioned): its Gram blocks; the images and the projected
       ! blocks keep following the block - no second sweep
       pass = pass + 1
       call check_dav(dav_ortho_gram(h, int(m, c_int), int(kt, c_int), c2, int(max(m, 1), c_int64_t), g2, &
            int(kt, c_int64_t)), "dav_ortho_gram")
    end do
    hm(1:m, m + 1:p) = hraw(1:m, :)
    hm(m + 1:p, 1:m) = transpose(hraw(1:m, :))
    hm(m + 1:p, m + 1:p) = hraw(m + 1:p, :)
    if (gev) then
       sm(1:m, m + 1:p) = sraw(1:m, :)
       sm(m + 1:p, 1:m) = transpose(sraw(1:m, :))
       sm(m + 1:p, m + 1:p) = sraw(m + 1:p, :)
    end if
  end subroutine project_with_last_pass

  !> Opt-in policy "locking" (see engine_set_correction_policy; oracle/davidson_oracle.py: generalized_eigensolver_dense_locking
  !> states the same loop in the reference's building blocks).  Basis columns 1..nlock are the locked Ritz vectors (W carries
  !> their images), nlock+1..m the active basis; hm is the projected matrix of all m columns, of which only the active block is
  !> ever solved.  A contraction V <- V Z, W <- W Z (dav_restart with the m x keep matrix Z) does the locking rotation and the
  !> collapse restart in one product.
  subroutine davidson_locking_loop(h, n, lowest, meth, pol, max_iterations, tolerance, max_dim, cap, gev, hm, sm, ld, m, dev, refresh_every, &
       adaptive_c, eigenvalues, iters, done, phase_s, t0)
    type(c_ptr), intent(in) :: h
    integer, intent(in) :: n, lowest, meth, pol, max_iterations, max_dim, cap, refresh_every
    real(dp), intent(in) :: tolerance, adaptive_c
    logical, intent(in) :: gev
    real(dp), intent(inout), contiguous :: hm(:, :), sm(:, :)   !< projected matrices of all m columns (only the active block is ever solved)
    integer(c_int64_t), intent(in) :: ld
    integer, intent(inout) :: m                            !< basis width: locked (guard) columns first, then the active basis
    type(device_ortho), intent(inout) :: dev
    real(dp), intent(out) :: eigenvalues(lowest)
    integer, intent(inout) :: iters
    logical, intent(inout) :: done
    real(dp), intent(inout) :: phase_s(8), t0              !< wall time by phase (davidson_engine%phase_seconds) and the last lap's clock
    integer :: kt, inner, opass, nrestart
    integer(c_int), allocatable :: sel(:)
    real(dp), allocatable :: tols(:)
    integer :: nlock, want, ma, nconv, want_new, m_rest, keep_a, keep, jj, col, it, lockbase
    real(dp), allocatable :: th(:), ya(:, :), yfull(:, :), err(:), theta_lock(:), z(:, :), lam(:), ysel(:, :), th_sel(:), yconv(:, :), za(:, :), &
         eye_k(:, :), perm(:, :)
    integer, allocatable :: rest(:), order(:)
    logical, allocatable :: conv(:)
    logical :: grow
    allocate(theta_lock(lowest), lam(lowest))
    nlock = 0
    nrestart = 0
    ! generalized problems: basis columns 1..nlock are GUARD vectors (an orthonormal basis of span(B X_locked), see the header of this
    ! procedure's caller); the locked eigenvectors themselves wait in the Ritz-vector panel behind its first `lowest` columns
    lockbase = lowest
    do it = 1, max_iterations
       want = lowest - nlock
       ma = m - nlock
       call check_dav(dav_agree_next(h, [real(it, dp), real(m, dp), real(nlock, dp), real(want, dp), tolerance, real(pol, dp), &
            real(meth, dp)], 7_c_int), "dav_agree_next")
       if (allocated(th)) deallocate(th, ya, yfull, err, conv)
       allocate(th(ma), ya(ma, ma), yfull(m, want), err(want), conv(want))
       if (gev) then
          call lapack_rayleigh_ritz(hm(nlock + 1:m, nlock + 1:m), th, ya, ma, sm(nlock + 1:m, nlock + 1:m))
       else
          call lapack_rayleigh_ritz(hm(nlock + 1:m, nlock + 1:m), th, ya, ma)
       end if
       call lap(phase_s, t0, 2)
       yfull = 0.0_dp
       yfull(nlock + 1:m, :) = ya(:, 1:want)
       call check_dav(dav_ritz_residual_correction_n(h, int(m, c_int), int(want, c_int), int(want, c_int), yfull, &
            int(m, c_int64_t), th, int(meth, c_int), err), "dav_ritz_residual_correction")
       call lap(phase_s, t0, 3)
       conv = err < tolerance
       nconv = count(conv)
       if (trace_iterations()) print "(a, i0, a, i0, a, i0, a, es10.3, a, i0)", "davidson trace (locking): iteration ", it, " m=", m, &
            " locked=", nlock, " max residual ", maxval(err), " newly below tolerance ", nconv
       want_new = want - nconv
       m_rest = ma - nconv
       ! Ritz vectors that stay active, in ascending order of their values
       if (allocated(rest)) deallocate(rest)
       allocate(rest(m_rest))
       col = 0
       do jj = 1, ma
          if (jj <= want) then
             if (conv(jj)) cycle
          end if
          col = col + 1
          rest(col) = jj
       end do
       ! the Ritz vectors that converge now (ma x nconv), in the order they are locked in
       if (allocated(yconv)) deallocate(yconv)
       allocate(yconv(ma, max(nconv, 1)))
       col = 0
       do jj = 1, want
          if (conv(jj)) then
             col = col + 1
             yconv(:, col) = ya(:, jj)
             theta_lock(nlock + col) = th(jj)
          end if
       end do
       if (want_new == 0) then
          ! every wanted pair is locked or has just converged: Ritz vectors in ascending order of the eigenvalues
          lam(1:lowest) = theta_lock(1:lowest)
          allocate(order(lowest))
          call ascending_order(lam, order)
          eigenvalues = lam(order)
          if (gev) then
             ! the vectors locked earlier wait in the Ritz-vector panel; those of this iteration join them, then one permutation
             call check_dav(dav_panel_transform(h, DAV_PANEL_V, int(nlock, c_int), int(ma, c_int), yconv, int(ma, c_int64_t), int(nconv, c_int), &
                  DAV_PANEL_X, int(lockbase + nlock, c_int)), "dav_panel_transform")
             allocate(perm(lowest, lowest))
             perm = 0.0_dp
             do jj = 1, lowest
                perm(order(jj), jj) = 1.0_dp
             end do
             call check_dav(dav_panel_transform(h, DAV_PANEL_X, int(lockbase, c_int), int(lowest, c_int), perm, int(lowest, c_int64_t), &
                  int(lowest, c_int), DAV_PANEL_X, 0_c_int), "dav_panel_transform")
          else
             allocate(ysel(m, lowest))
             ysel = 0.0_dp
             do jj = 1, lowest
                if (order(jj) <= nlock) then
                   ysel(order(jj), jj) = 1.0_dp
                else
                   ysel(nlock + 1:m, jj) = yconv(:, order(jj) - nlock)
                end if
             end do
             call check_dav(dav_ritz_vectors(h, int(m, c_int), int(lowest, c_int), ysel, int(m, c_int64_t)), "dav_ritz_vectors")
          end if
          iters = it
          done = .true.
          return
       end if
       kt = min(want_new, n - m)
       grow = ((m_rest + want_new <= max_dim) .or. (m_rest <= 2 * want_new)) .and. (m + kt <= cap) .and. kt > 0
       if (grow) then
          ! the corrections of the pairs that stay wanted, compacted to the front of the block behind the basis
          if (allocated(sel)) deallocate(sel, th_sel)
          allocate(sel(kt), th_sel(kt))
          col = 0
          do jj = 1, want
             if (.not. conv(jj) .and. col < kt) then
                col = col + 1
                sel(col) = int(jj - 1, c_int)
                th_sel(col) = th(jj)
             end if
          end do
          if (meth == DAV_METHOD_GJD) then
             call check_dav(dav_panel_select(h, DAV_PANEL_X, 0_c_int, int(kt, c_int), sel), "dav_panel_select")
             call check_dav(dav_panel_select(h, DAV_PANEL_R, 0_c_int, int(kt, c_int), sel), "dav_panel_select")
             if (allocated(tols)) deallocate(tols)
             allocate(tols(kt))
             do jj = 1, kt
                tols(jj) = gjd_tol_wanted(err(sel(jj) + 1), tolerance, adaptive_c)
             end do
             call check_dav(dav_gjd_correction_n(h, int(m, c_int), int(kt, c_int), th_sel, 300_c_int, 1.0e-10_dp, tols, inner), &
                  "dav_gjd_correction")
             call lap(phase_s, t0, 8)
          else
             call check_dav(dav_panel_select(h, DAV_PANEL_V, int(m, c_int), int(kt, c_int), sel), "dav_panel_select")
          end if
          keep_a = m_rest
       else
          keep_a = min(2 * want_new, m_rest)          ! collapse restart of the active basis (src/davidson.f90:218)
       end if
       keep = nlock + nconv + keep_a
       if (nconv > 0 .or. keep < m) then
          ! Z = [e_1 .. e_nlock | the pairs just locked | Ritz vectors kept active]
          if (allocated(z)) deallocate(z)
          allocate(z(m, keep))
          z = 0.0_dp
          do jj = 1, nlock
             z(jj, jj) = 1.0_dp
          end do
          if (allocated(za)) deallocate(za)
          allocate(za(ma, keep_a))
          do jj = 1, keep_a
             za(:, jj) = ya(:, rest(jj))
          end do
          if (gev) then
             ! The Ritz vectors of a generalized problem are S-orthonormal: the kept ones are made Euclidean-orthonormal (as after a
             ! restart of the reference policy); the pairs just locked leave the basis altogether - their eigenvectors X = V_a y go to
             ! the Ritz-vector panel, and what takes their columns are their GUARD vectors B X = (B V_a) y: the pairs still wanted are
             ! B-orthogonal to the locked ones, x^T (B x_l) = 0, so the search space is kept orthogonal to span(B X_locked) instead of
             ! span(X_locked); the remaining Ritz vectors already are (y_j^T S y_l = 0).  The guard columns carry NO images (W, B V
             ! stay zero there): nothing ever projects on them - which is why the generalized loop makes its orthonormalisation
             ! passes before the sweep (below) instead of fusing the last one with the projection, whose update reads W of every column
             if (keep_a > 0) call restart_transform(za, ma, keep_a)
             if (nconv > 0) then
                call check_dav(dav_panel_transform(h, DAV_PANEL_V, int(nlock, c_int), int(ma, c_int), yconv, int(ma, c_int64_t), int(nconv, c_int), &
                     DAV_PANEL_X, int(lockbase + nlock, c_int)), "dav_panel_transform")
                ! (into the residue panel: the correction of this iteration has been made, nothing reads it before the next Ritz phase;
                ! the scratch panel S is what a contraction of more than 64 columns goes through - dav_restart below)
                call check_dav(dav_panel_transform(h, DAV_PANEL_BV, int(nlock, c_int), int(ma, c_int), yconv, int(ma, c_int64_t), int(nconv, c_int), &
                     DAV_PANEL_R, 0_c_int), "dav_panel_transform")
             end if
          else
             do jj = 1, nconv
                z(nlock + 1:m, nlock + jj) = yconv(:, jj)
             end do
          end if
          z(nlock + 1:m, nlock + nconv + 1:keep) = za
          ! (a growing iteration keeps every active Ritz vector, keep = m: the correction block behind column m stays where it is)
          call check_dav(dav_restart(h, int(m, c_int), int(keep, c_int), z, int(m, c_int64_t)), "dav_restart")
          hm(1:keep, 1:keep) = lapack_matmul("T", "N", z, lapack_matmul("N", "N", hm(1:m, 1:m), z))
          if (gev) sm(1:keep, 1:keep) = lapack_matmul("T", "N", z, lapack_matmul("N", "N", sm(1:m, 1:m), z))
          if (keep < m) hm(keep + 1:m, :) = 0.0_dp
          if (keep < m) hm(:, keep + 1:m) = 0.0_dp
          if (gev .and. keep < m) sm(keep + 1:m, :) = 0.0_dp
          if (gev .and. keep < m) sm(:, keep + 1:m) = 0.0_dp
          if (gev .and. nconv > 0) then
             ! the guard vectors of the pairs just locked into their columns, orthonormal to the guards before them and to each other
             if (allocated(eye_k)) deallocate(eye_k)
             allocate(eye_k(nconv, nconv))
             eye_k = 0.0_dp
             do jj = 1, nconv
                eye_k(jj, jj) = 1.0_dp
             end do
             call check_dav(dav_panel_transform(h, DAV_PANEL_R, 0_c_int, int(nconv, c_int), eye_k, int(nconv, c_int64_t), int(nconv, c_int), &
                  DAV_PANEL_V, int(nlock, c_int)), "dav_panel_transform")
             call block_orthonormalise(dev, n, nlock, nconv)
          end if
          nlock = nlock + nconv
          if (grow .and. keep < m) then
             print *, "locking_loop: internal error (a growing iteration keeps every active Ritz vector)"
             error stop
          end if
          m = keep
          nrestart = nrestart + 1
          if (mod(nrestart, refresh_every) == 0) then
             ! W was contracted, not recomputed: every refresh_every-th contraction re-applies the operator (as after restarts) - to
             ! the active columns (generalized problems: the guard columns carry no images)
             if (gev) then
                call check_dav(dav_expand(h, int(nlock, c_int), int(m - nlock, c_int)), "dav_expand")
             else
                call check_dav(dav_expand(h, 0_c_int, int(m, c_int)), "dav_expand")
             end if
             call check_dav(dav_project(h, 0_c_int, int(m, c_int), hm, ld, sm, ld), "dav_project")
          end if
          call lap(phase_s, t0, 7)
       end if
       if (grow) then
          if (gev) then
             ! all passes, the sweep, the plain projection (see above: the fused last pass would read images of the guard columns)
             call block_orthonormalise(dev, n, m, kt)
             call lap(phase_s, t0, 4)
             call check_dav(dav_expand(h, int(m, c_int), int(kt, c_int)), "dav_expand")
             call lap(phase_s, t0, 5)
             call check_dav(dav_project(h, int(m, c_int), int(kt, c_int), hm, ld, sm, ld), "dav_project")
             call lap(phase_s, t0, 6)
          else
             call block_orthonormalise(dev, n, m, kt, only_first=.true., last_pass=opass)
             call lap(phase_s, t0, 4)
             call check_dav(dav_expand(h, int(m, c_int), int(kt, c_int)), "dav_expand")
             call lap(phase_s, t0, 5)
             call project_with_last_pass(h, n, m, kt, gev, hm, sm, ld, dev, opass + 1)
             call lap(phase_s, t0, 6)
          end if
          m = m + kt
       end if
    end do
    ! not converged: what is locked plus the current Ritz pairs of the active basis (the caller prints the warning)
    lam(1:nlock) = theta_lock(1:nlock)
    if (nlock < lowest) lam(nlock + 1:lowest) = th(1:lowest - nlock)
    allocate(order(lowest))
    call ascending_order(lam, order)
    eigenvalues = lam(order)
    if (size(yfull, 1) == m .or. gev) then
       if (gev) then
          ! the current Ritz vectors of the pairs still wanted behind the locked ones (the last iteration's active basis: ya)
          want = lowest - nlock
          if (want > 0 .and. size(ya, 1) == m - nlock) call check_dav(dav_panel_transform(h, DAV_PANEL_V, int(nlock, c_int), int(m - nlock, c_int), &
               ya(:, 1:want), int(m - nlock, c_int64_t), int(want, c_int), DAV_PANEL_X, int(lockbase + nlock, c_int)), "dav_panel_transform")
          allocate(perm(lowest, lowest))
          perm = 0.0_dp
          do jj = 1, lowest
             perm(order(jj), jj) = 1.0_dp
          end do
          call check_dav(dav_panel_transform(h, DAV_PANEL_X, int(lockbase, c_int), int(lowest, c_int), perm, int(lowest, c_int64_t), &
               int(lowest, c_int), DAV_PANEL_X, 0_c_int), "dav_panel_transform")
       else
          allocate(ysel(m, lowest))
          ysel = 0.0_dp
          do jj = 1, lowest
             if (order(jj) <= nlock) then
                ysel(order(jj), jj) = 1.0_dp
             else if (order(jj) - nlock <= size(yfull, 2)) then
                ysel(:, jj) = yfull(:, order(jj) - nlock)
             end if
          end do
          call check_dav(dav_ritz_vectors(h, int(m, c_int), int(lowest, c_int), ysel, int(m, c_int64_t)), "dav_ritz_vectors")
       end if
    end if
  end subroutine davidson_locking_loop

  subroutine device_ortho_gram(be, m, kt, c, g)
    class(device_ortho), intent(inout) :: be
    integer, intent(in) :: m, kt
    real(dp), intent(out) :: c(:, :), g(:, :)
    call check_dav(dav_ortho_gram(be%h, int(m, c_int), int(kt, c_int), c, int(size(c, 1), c_int64_t), g, int(size(g, 1), c_int64_t)), &
         "dav_ortho_gram")
  end subroutine device_ortho_gram

  subroutine device_ortho_apply(be, m, kt, c, mm)
    class(device_ortho), intent(inout) :: be
    integer, intent(in) :: m, kt
    real(dp), intent(in) :: c(:, :), mm(:, :)
    call check_dav(dav_ortho_apply(be%h, int(m, c_int), int(kt, c_int), c, int(size(c, 1), c_int64_t), mm, int(size(mm, 1), c_int64_t)), &
         "dav_ortho_apply")
  end subroutine device_ortho_apply

  function device_ortho_unit_column(be, m, j, entry) result(ok)
    class(device_ortho), intent(inout) :: be
    integer, intent(in) :: m, j, entry
    logical :: ok
    integer(c_int) :: ierr
    ierr = dav_panel_unit_column(be%h, DAV_PANEL_V, int(m + j - 1, c_int), int(entry, c_int))
    ! DAV_NO_SUCH_ENTRY: the engine keeps no such entry of the start order - the caller takes another direction; anything else is an error
    if (ierr /= 0_c_int .and. ierr /= DAV_NO_SUCH_ENTRY) call check_dav(ierr, "dav_panel_unit_column")
    ok = ierr == 0_c_int
  end function device_ortho_unit_column

  subroutine device_ortho_put_column(be, m, j, vec)
    class(device_ortho), intent(inout) :: be
    integer, intent(in) :: m, j
    real(dp), intent(in) :: vec(:)
    call check_dav(dav_panel_put(be%h, DAV_PANEL_V, int(m + j - 1, c_int), 1_c_int, vec, int(size(vec), c_int64_t)), "dav_panel_put")
  end subroutine device_ortho_put_column

  !> Apply host callbacks to basis columns c0+1..c0+k: download the block, call, upload A*V and B*V.
  subroutine apply_host_block(h, n, c0, k, fun_a, fun_b)
    type(c_ptr), intent(in) :: h
    integer, intent(in) :: n, c0, k
    procedure(block_operator) :: fun_a, fun_b
    real(dp), allocatable :: blk(:, :), img(:, :)
    allocate(blk(n, k), img(n, k))
    call check_dav(dav_panel_get(h, DAV_PANEL_V, int(c0, c_int), int(k, c_int), blk, int(n, c_int64_t)), &
         "dav_panel_get")
    img = fun_a(blk)
    call check_dav(dav_panel_put(h, DAV_PANEL_W, int(c0, c_int), int(k, c_int), img, int(n, c_int64_t)), &
         "dav_panel_put")
    img = fun_b(blk)
    call check_dav(dav_panel_put(h, DAV_PANEL_BV, int(c0, c_int), int(k, c_int), img, int(n, c_int64_t)), &
         "dav_panel_put")
  end subroutine apply_host_block

end module davidson_device


module davidson_dense
  use, intrinsic :: iso_c_binding
  use numeric_kinds, only: dp
  use davidson_hip_c
  use davidson_device
  implicit none
  private
  public :: generalized_eigensolver_dense

contains

  !> Dense front-end, argument list of the reference (src/davidson.f90:51-52, :74-83): the matrices
  !> are uploaded to HBM, solved there and everything is released before returning (the reference
  !> keeps no state across calls either, src/davidson.f90:238-244).
  subroutine generalized_eigensolver_dense(matrix, eigenvalues, eigenvectors, lowest, method, max_iterations, &
       tolerance, iters, max_dim_sub, second_matrix)
    integer, intent(in) :: lowest
    real(dp), dimension(:, :), intent(in) :: matrix
    real(dp), dimension(:, :), intent(in), optional :: second_matrix
    real(dp), dimension(lowest), intent(out) :: eigenvalues
    real(dp), dimension(:, :), intent(out) :: eigenvectors
    integer, intent(in) :: max_iterations
    integer, intent(in), optional :: max_dim_sub
    real(dp), intent(in) :: tolerance
    character(len=*), intent(in) :: method
    integer, intent(out) :: iters

    type(davidson_engine) :: eng
    integer :: max_dim
    logical :: symmetric
    real(dp) :: t(5)

    max_dim = 10 * lowest
    if (present(max_dim_sub)) max_dim = max_dim_sub
    t(1) = wall()
    call engine_create(eng, size(matrix, 1), lowest, max_dim, present(second_matrix), env_device())
    ! Storage of the uploaded operators.  The reference ASSUMES a symmetric matrix (it never checks; its DPR / GJD and DSYEV steps
    ! are only meaningful for one), so only the lower block triangle needs to cross PCIe and stay in HBM: half the upload - which
    ! is what a drop-in call spends its time on (N=20000: 60 of 62 ms) - and half the bytes per sweep.  Default (round 5): symmetric
    ! tiles when a cheap exact probe of the host matrices finds them symmetric (symmetry_probe: whole sampled rows against their
    ! columns); a matrix that fails the probe is uploaded in full and swept as the reference's DGEMM would read it.
    ! DAVIDSON_STORAGE=full | symmetric overrides; full rows that do not fit the device fall back to symmetric tiles as before.
    select case (env_storage())
    case (1)
       symmetric = .true.
    case (0)
       symmetric = .not. fits_as_full_rows(eng, size(matrix, 1), merge(2, 1, present(second_matrix)))
    case default
       symmetric = symmetry_probe(matrix)
       if (symmetric .and. present(second_matrix)) symmetric = symmetry_probe(second_matrix)
       if (.not. symmetric) symmetric = .not. fits_as_full_rows(eng, size(matrix, 1), merge(2, 1, present(second_matrix)))
    end select
    if (symmetric) call engine_set_storage(eng, "symmetric")
    t(2) = wall()
    call engine_set_dense(eng, 1, matrix)
    if (present(second_matrix)) call engine_set_dense(eng, 2, second_matrix)
    t(3) = wall()
    call generalized_eigensolver_device(eng, eigenvalues, eigenvectors, lowest, method, max_iterations, &
         tolerance, iters, max_dim)
    t(4) = wall()
    call engine_destroy(eng)
    t(5) = wall()
    if (dense_verbose()) print "(a, i0, a, l1, a, 4f9.3)", "davidson dense call: n=", size(matrix, 1), " symmetric tiles=", symmetric, &
         " ms[create+probe upload solve+download destroy]=", (t(2:5) - t(1:4)) * 1.0e3_dp

  contains

    function wall() result(s)
      real(dp) :: s
      integer(c_int64_t) :: cnt, rate
      call system_clock(cnt, rate)
      s = real(cnt, dp) / real(rate, dp)
    end function wall

    function dense_verbose() result(on)
      logical :: on
      integer :: stat, length
      character(len=8) :: buf
      call get_environment_variable("DAVIDSON_VERBOSE", buf, length, stat)
      on = (stat == 0 .and. length > 0)
    end function dense_verbose

  end subroutine generalized_eigensolver_dense

end module davidson_dense


module davidson_free
  use, intrinsic :: iso_c_binding
  use numeric_kinds, only: dp
  use davidson_hip_c
  use davidson_device
  implicit none
  private
  public :: generalized_eigensolver_free, free_matmul

contains

  !> Matrix-free front-end, argument list of the reference (src/davidson.f90:277-278, :305-337):
  !> A and B arrive as host callbacks that map a block of vectors.  The basis, A*V, B*V and every
  !> N-long product live on the GPU; only the NEW basis columns travel to the host for the callback
  !> (the reference re-applies both operators to the whole basis each iteration, :378-379).
  !> As in the reference the problem is always generalized and the correction always DPR (:428),
  !> and convergence is tested on all pairs at once (:416).
  subroutine generalized_eigensolver_free(fun_matrix_gemv, eigenvalues, ritz_vectors, lowest, method, &
       max_iterations, tolerance, iters, max_dim_sub, fun_second_matrix_gemv)
    integer, intent(in) :: lowest
    real(dp), dimension(lowest), intent(out) :: eigenvalues
    real(dp), dimension(:, :), intent(out) :: ritz_vectors
    integer, intent(in) :: max_iterations
    integer, intent(in), optional :: max_dim_sub
    real(dp), intent(in) :: tolerance
    character(len=*), intent(in) :: method
    integer, intent(out) :: iters
    interface
       function fun_matrix_gemv(input_vect) result(output_vect)
         use numeric_kinds, only: dp
         real(dp), dimension(:, :), intent(in) :: input_vect
         real(dp), dimension(size(input_vect, 1), size(input_vect, 2)) :: output_vect
       end function fun_matrix_gemv
       function fun_second_matrix_gemv(input_vect) result(output_vect)
         use numeric_kinds, only: dp
         real(dp), dimension(:, :), intent(in) :: input_vect
         real(dp), dimension(size(input_vect, 1), size(input_vect, 2)) :: output_vect
       end function fun_second_matrix_gemv
    end interface

    type(davidson_engine) :: eng
    real(dp), allocatable :: diag_a(:), diag_b(:)
    integer :: n, max_dim

    n = size(ritz_vectors, 1)
    max_dim = 10 * lowest
    if (present(max_dim_sub)) max_dim = max_dim_sub
    allocate(diag_a(n), diag_b(n))
    call extract_diagonal_blocked(fun_matrix_gemv, n, diag_a)
    call extract_diagonal_blocked(fun_second_matrix_gemv, n, diag_b)

    call engine_create(eng, n, lowest, max_dim, .true., env_device())
    call check_dav(dav_set_operator_host(eng%h, DAV_OP_A, diag_a), "dav_set_operator_host")
    call check_dav(dav_set_operator_host(eng%h, DAV_OP_B, diag_b), "dav_set_operator_host")
    call davidson_device_loop(eng%h, n, lowest, method, max_iterations, tolerance, iters, max_dim, .true., &
         .false., eigenvalues, fun_matrix_gemv, fun_second_matrix_gemv, eng%policy, eng%device_rr)
    call check_dav(dav_panel_get(eng%h, DAV_PANEL_X, 0_c_int, int(lowest, c_int), ritz_vectors, &
         int(size(ritz_vectors, 1), c_int64_t)), "dav_panel_get")
    call engine_destroy(eng)
  end subroutine generalized_eigensolver_free

  !> Diagonal of an operator known only through its block apply: same N unit-vector probes as
  !> extract_diagonal_free (src/davidson.f90:490-523), sent through the callback 64 at a time.
  subroutine extract_diagonal_blocked(fun, n, diag)
    integer, intent(in) :: n
    real(dp), intent(out) :: diag(n)
    interface
       function fun(input_vect) result(output_vect)
         use numeric_kinds, only: dp
         real(dp), dimension(:, :), intent(in) :: input_vect
         real(dp), dimension(size(input_vect, 1), size(input_vect, 2)) :: output_vect
       end function fun
    end interface
    integer, parameter :: chunk = 64
    real(dp), allocatable :: probe(:, :), image(:, :)
    integer :: i0, k, j
    do i0 = 1, n, chunk
       k = min(chunk, n - i0 + 1)
       allocate(probe(n, k), image(n, k))
       probe = 0.0_dp
       do j = 1, k
          probe(i0 + j - 1, j) = 1.0_dp
       end do
       image = fun(probe)
       do j = 1, k
          diag(i0 + j - 1) = image(i0 + j - 1, j)
       end do
       deallocate(probe, image)
    end do
  end subroutine extract_diagonal_blocked

  !> Apply a matrix given by a row generator fun(i, dim) to a block (public helper of the reference,
  !> src/davidson.f90:526-569).  Host code: it serves user callbacks; rows are independent.
  function free_matmul(fun, array) result(matrix)
    real(dp), dimension(:, :), intent(in) :: array
    real(dp), dimension(size(array, 1), size(array, 2)) :: matrix
    interface
       function fun(i, dim) result(vec)
         use numeric_kinds, only: dp
         integer, intent(in) :: i
         integer, intent(in) :: dim
         real(dp), dimension(dim) :: vec
       end function fun
    end interface
    real(dp), allocatable :: row(:)
    integer :: i, n
    n = size(array, 1)
    !$OMP PARALLEL DO PRIVATE(i, row)
    do i = 1, n
       row = fun(i, n)
       matrix(i, :) = matmul(row, array)
    end do
    !$OMP END PARALLEL DO
  end function free_matmul

end module davidson_free


module davidson
  use numeric_kinds, only: dp
  use davidson_dense, only: generalized_eigensolver_dense
  use davidson_free, only: generalized_eigensolver_free
  use davidson_device, only: generalized_eigensolver_device, davidson_free_buffers
  implicit none
  private
  public :: generalized_eigensolver, davidson_free_buffers

  !> Generic of the reference (src/davidson.f90:601-625), resolved by the first argument: a matrix,
  !> a block-apply procedure, or (new) a device-resident `davidson_engine`.
  interface generalized_eigensolver
     procedure generalized_eigensolver_dense
     procedure generalized_eigensolver_free
     procedure generalized_eigensolver_device
  end interface generalized_eigensolver

end module davidson
