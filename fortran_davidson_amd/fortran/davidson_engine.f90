!> The handle of a device-resident eigenproblem and everything that sets it up: engine creation, the operators A and B (dense from
!> host memory or a file, generated, matrix-free device operators, the caller's own kernel), storage, communicator, the opt-in knobs
!> of a solve.  Thin Fortran faces of the C ABI (include/davidson_hip.h); the solve itself is module davidson_device (davidson.f90).
module davidson_engine_setup
  use, intrinsic :: iso_c_binding
  use numeric_kinds, only: dp
  use davidson_hip_c
  use davidson_knobs, only: basis_capacity
  implicit none
  private
  public :: davidson_engine, engine_create, engine_destroy, engine_set_dense, engine_set_storage, fits_as_full_rows, engine_set_device_rr, &
       engine_set_inner_precision, engine_read_matrix, engine_dense_begin, engine_dense_put_rows, engine_dense_end, &
       engine_set_correction_policy, engine_generate_diagonal_dominant, engine_set_hashed_operator, engine_set_harness_operator, &
       engine_set_identity, engine_set_device_operator, engine_comm_unique_id, engine_comm_init, davidson_free_buffers

  !> Handle of a device-resident problem: operators A (and B) plus all work panels in HBM.
  type :: davidson_engine
     type(c_ptr) :: h = c_null_ptr
     integer :: n = 0
     integer :: max_cols = 0
     logical :: gev = .false.
     integer :: policy = 0          !< POLICY_ALL (the reference) or POLICY_UNCONVERGED (opt-in)
     !> .true. when operator A is matrix-free (a device operator instead of a stored matrix): the solve then
     !> follows the reference's matrix-free driver - convergence of all wanted pairs tested at once, no
     !> sticky flags (src/davidson.f90:416) - instead of the dense one (:176)
     logical :: free_semantics = .false.
     !> opt-in (engine_set_device_rr / DAVIDSON_DEVICE_RR=1): the Rayleigh-Ritz problem is solved on the device
     !> (one-workgroup Jacobi, order <= 128) and the projected matrices, Ritz values and vectors never leave HBM
     logical :: device_rr = .false.
     !> Wall time of the last solve on this engine by phase (seconds): 1 setup (init basis + first projection),
     !> 2 host Rayleigh-Ritz (DSYEV/DSYGV), 3 Ritz/residue/correction phase, 4 orthonormalisation,
     !> 5 operator apply (expand), 6 projection, 7 restart, 8 GJD inner solves.  Printed when the
     !> environment variable DAVIDSON_VERBOSE is set; never printed otherwise (drop-in silence).
     real(dp) :: phase_seconds(8) = 0.0_dp
  end type davidson_engine

  !> Correction policies of the outer loop (see davidson_device_loop)
  integer, parameter, public :: POLICY_ALL = 0, POLICY_UNCONVERGED = 1, POLICY_LOCKING = 2

contains

  !> Give the device and pinned blocks the library keeps from the engine destroyed last back to the device (the buffer cache of
  !> include/davidson_hip.h: a call per eigenproblem - the reference's interface - otherwise pays hipMalloc / hipFree of the whole
  !> problem every time).  What mkl_free_buffers is to MKL.
  subroutine davidson_free_buffers()
    integer(c_int) :: ierr
    ierr = dav_free_buffers()
  end subroutine davidson_free_buffers

  !> Do `nmat` dense operators of order n fit the engine's device as full rows (8 n^2 bytes each), with a tenth of the memory left
  !> for the panels and the partial-sum slabs?  (dav_device_memory: what is free now.)
  function fits_as_full_rows(eng, n, nmat) result(fits)
    type(davidson_engine), intent(in) :: eng
    integer, intent(in) :: n, nmat
    logical :: fits
    integer(c_int64_t) :: free_bytes, total_bytes
    call check_dav(dav_device_memory(eng%h, free_bytes, total_bytes), "dav_device_memory")
    fits = 8.0_dp * real(n, dp) * real(n, dp) * real(nmat, dp) <= 0.9_dp * real(free_bytes, dp)
  end function fits_as_full_rows

  subroutine engine_create(eng, n, lowest, max_dim_sub, gev, device, rank, nranks)
    type(davidson_engine), intent(out) :: eng
    integer, intent(in) :: n, lowest
    integer, intent(in), optional :: max_dim_sub, device, rank, nranks
    logical, intent(in), optional :: gev
    integer :: max_dim, dev, rk, nr, envlen, envstat
    character(len=32) :: envbuf
    max_dim = 10 * lowest
    if (present(max_dim_sub)) max_dim = max_dim_sub
    dev = 0; rk = 0; nr = 1
    if (present(device)) dev = device
    if (present(rank)) rk = rank
    if (present(nranks)) nr = nranks
    eng%n = n
    eng%gev = .false.
    if (present(gev)) eng%gev = gev
    eng%max_cols = basis_capacity(lowest, max_dim)
    if (dav_version() /= DAV_HIP_ABI_VERSION) then
       print *, "engine_create: libdavidson_hip.so reports ABI version ", dav_version(), ", these modules were built for ", &
            DAV_HIP_ABI_VERSION
       error stop
    end if
    call check_dav(dav_create(eng%h, int(dev, c_int), int(n, c_int64_t), int(eng%max_cols, c_int), &
         merge(1_c_int, 0_c_int, eng%gev), int(rk, c_int), int(nr, c_int)), "dav_create")
    ! engine knob that does not touch the reference's argument lists (reaches the dense and matrix-free
    ! front ends too): DAVIDSON_CORRECTION_POLICY=unconverged
    call get_environment_variable("DAVIDSON_CORRECTION_POLICY", envbuf, envlen, envstat)
    eng%policy = POLICY_ALL
    if (envstat == 0 .and. envlen > 0) call engine_set_correction_policy(eng, envbuf(1:envlen))
    call get_environment_variable("DAVIDSON_INNER_PRECISION", envbuf, envlen, envstat)
    if (envstat == 0 .and. envlen >= 2) then
       if (envbuf(1:2) == "32") call engine_set_inner_precision(eng, 32)
    end if
    call get_environment_variable("DAVIDSON_DEVICE_RR", envbuf, envlen, envstat)
    eng%device_rr = (envstat == 0 .and. envlen > 0 .and. envbuf(1:1) == "1")
  end subroutine engine_create

  !> Mixed-precision correction path (SURVEY 8f-4): bits = 32 lets the block sweeps inside the GJD correction solve
  !> read an fp32 copy of the stored symmetric tiles (fp64 products and sums); residuals, projections and the
  !> convergence test stay on the fp64 matrix.  bits = 64 (default) = the reference's precision throughout.
  subroutine engine_set_inner_precision(eng, bits)
    type(davidson_engine), intent(inout) :: eng
    integer, intent(in) :: bits
    call check_dav(dav_set_inner_precision(eng%h, int(bits, c_int)), "dav_set_inner_precision")
  end subroutine engine_set_inner_precision

  !> Device-side Rayleigh-Ritz (SURVEY 8f-1) on or off for the solves of this engine.  Off (default): the projected
  !> problem is solved by host LAPACK, as the reference does (src/lapack_wrapper.f90:14-91).  On: one-workgroup Jacobi
  !> eigensolver on the device for bases up to 128 columns (wider bases fall back to the host); same Ritz pairs to
  !> rounding, same iteration counts; the H-down / Y-up transfers and one synchronisation per iteration disappear.
  subroutine engine_set_device_rr(eng, on)
    type(davidson_engine), intent(inout) :: eng
    logical, intent(in) :: on
    eng%device_rr = on
  end subroutine engine_set_device_rr

  !> Which Ritz pairs get a correction vector each iteration.  "all" (default) is the reference's policy:
  !> one correction per basis vector, the basis doubles (src/davidson.f90:195-213), sticky convergence on the
  !> dense path.  "unconverged" (opt-in; not in the reference, changes the iteration count): only those of
  !> the `lowest` wanted pairs whose residual is still above the tolerance are corrected, convergence is
  !> tested on all wanted pairs at once, and the basis grows by at most `lowest` columns per iteration -
  !> narrower panels, one 16-column pass of the symmetric sweep per iteration, and GJD inner solves only for
  !> the pairs that need them.  "locking" (opt-in; standard problems; the deflation the reference's header cites and never
  !> implements, src/davidson.f90:7-8): a wanted pair whose residual is below the tolerance is locked - its Ritz vector leaves the
  !> active basis, which is kept orthogonal to it, its value is final - and the Rayleigh-Ritz problem, the corrections and the
  !> restarts only concern the pairs still wanted (locking_loop in davidson_device_loop; oracle:
  !> generalized_eigensolver_dense_locking).
  subroutine engine_set_correction_policy(eng, policy)
    type(davidson_engine), intent(inout) :: eng
    character(len=*), intent(in) :: policy
    select case (trim(policy))
    case ("all")
       eng%policy = POLICY_ALL
    case ("unconverged")
       eng%policy = POLICY_UNCONVERGED
    case ("locking")
       eng%policy = POLICY_LOCKING
    case default
       print *, "engine_set_correction_policy: policy must be 'all', 'unconverged' or 'locking', got '", trim(policy), "'"
       error stop
    end select
  end subroutine engine_set_correction_policy

  subroutine engine_destroy(eng)
    type(davidson_engine), intent(inout) :: eng
    if (c_associated(eng%h)) call check_dav(dav_destroy(eng%h), "dav_destroy")
    eng%h = c_null_ptr
  end subroutine engine_destroy

  subroutine engine_comm_unique_id(id)
    character(kind=c_char), intent(out) :: id(128)
    call check_dav(dav_comm_unique_id(id), "dav_comm_unique_id")
  end subroutine engine_comm_unique_id

  subroutine engine_comm_init(eng, id)
    type(davidson_engine), intent(inout) :: eng
    character(kind=c_char), intent(in) :: id(128)
    call check_dav(dav_comm_init(eng%h, id), "dav_comm_init")
  end subroutine engine_comm_init

  !> Storage of the dense operators set afterwards: "full" (default) or "symmetric" = only the lower
  !> block triangle is kept in HBM (N(N+1)/2 entries: N = 200000 fits one MI355X) and every
  !> off-diagonal tile is used twice per sweep.  Single GPU only.
  subroutine engine_set_storage(eng, storage)
    type(davidson_engine), intent(inout) :: eng
    character(len=*), intent(in) :: storage
    integer(c_int) :: mode
    select case (trim(storage))
    case ("full")
       mode = 0
    case ("symmetric")
       mode = 1
    case default
       print *, "engine_set_storage: storage must be 'full' or 'symmetric'"
       error stop
    end select
    call check_dav(dav_set_storage(eng%h, mode), "dav_set_storage")
  end subroutine engine_set_storage

  !> Upload a host matrix (full storage) as operator A (which=1) or B (which=2).
  subroutine engine_set_dense(eng, which, matrix)
    type(davidson_engine), intent(inout) :: eng
    integer, intent(in) :: which
    real(dp), dimension(:, :), intent(in) :: matrix
    if (size(matrix, 1) /= eng%n .or. size(matrix, 2) /= eng%n) then
       print *, "engine_set_dense: matrix must be ", eng%n, " x ", eng%n
       error stop
    end if
    call check_dav(dav_set_dense_host(eng%h, int(which - 1, c_int), matrix, int(size(matrix, 1), c_int64_t)), &
         "dav_set_dense_host")
    if (which == 1) eng%free_semantics = .false.
  end subroutine engine_set_dense

  !> Operator A (which=1) or B (which=2) from a file, streamed to HBM by blocks of rows - no host N x N
  !> array.  fmt = "text" (default): the format read_matrix reads and write_matrix writes in the reference's
  !> test_utils (src/tests/test_utils.f90:118-135,150-166: list-directed reals, row-major); fmt = "f64": raw
  !> float64, row-major, 8 n^2 bytes.
  subroutine engine_read_matrix(eng, which, path_file, fmt)
    type(davidson_engine), intent(inout) :: eng
    integer, intent(in) :: which
    character(len=*), intent(in) :: path_file
    character(len=*), intent(in), optional :: fmt
    integer(c_int) :: code
    code = 0
    if (present(fmt)) then
       select case (trim(fmt))
       case ("text")
          code = 0
       case ("f64")
          code = 1
       case default
          print *, "engine_read_matrix: fmt must be 'text' or 'f64'"
          error stop
       end select
    end if
    call check_dav(dav_set_dense_file(eng%h, int(which - 1, c_int), trim(path_file) // c_null_char, code), &
         "dav_set_dense_file")
    if (which == 1) eng%free_semantics = .false.
  end subroutine engine_read_matrix

  !> Streaming upload for hosts that produce the matrix row by row: begin, any number of put_rows (each a
  !> block of complete rows; rows(j, r) = element (row0 + r - 1, j), i.e. one matrix row per COLUMN of the
  !> Fortran array, which is the row-major order of the file format), end.
  subroutine engine_dense_begin(eng, which)
    type(davidson_engine), intent(inout) :: eng
    integer, intent(in) :: which
    call check_dav(dav_dense_begin(eng%h, int(which - 1, c_int)), "dav_dense_begin")
    if (which == 1) eng%free_semantics = .false.
  end subroutine engine_dense_begin

  subroutine engine_dense_put_rows(eng, which, row0, rows)
    type(davidson_engine), intent(inout) :: eng
    integer, intent(in) :: which
    integer, intent(in) :: row0                          !< 1-based global index of the first row
    real(dp), dimension(:, :), contiguous, intent(in) :: rows   !< (n, nrows)
    if (size(rows, 1) /= eng%n) then
       print *, "engine_dense_put_rows: rows must be (", eng%n, ", nrows)"
       error stop
    end if
    call check_dav(dav_dense_put_rows(eng%h, int(which - 1, c_int), int(row0 - 1, c_int64_t), &
         int(size(rows, 2), c_int64_t), rows, int(size(rows, 1), c_int64_t)), "dav_dense_put_rows")
  end subroutine engine_dense_put_rows

  subroutine engine_dense_end(eng, which)
    type(davidson_engine), intent(inout) :: eng
    integer, intent(in) :: which
    call check_dav(dav_dense_end(eng%h, int(which - 1, c_int)), "dav_dense_end")
  end subroutine engine_dense_end

  !> generate_diagonal_dominant(n, sparsity[, diag_val]) built directly in HBM (same entries as the
  !> host function of array_utils with the same seed).
  subroutine engine_generate_diagonal_dominant(eng, which, sparsity, diag_val, seed)
    type(davidson_engine), intent(inout) :: eng
    integer, intent(in) :: which
    real(dp), intent(in) :: sparsity
    real(dp), intent(in), optional :: diag_val
    integer, intent(in), optional :: seed
    integer(c_int64_t) :: s
    real(c_double) :: dv
    s = 1
    if (present(seed)) s = int(seed, c_int64_t)
    dv = 0.0_dp
    if (present(diag_val)) dv = diag_val
    call check_dav(dav_set_dense_generated(eng%h, int(which - 1, c_int), s, sparsity, &
         merge(1_c_int, 0_c_int, present(diag_val)), dv), "dav_set_dense_generated")
    if (which == 1) eng%free_semantics = .false.
  end subroutine engine_generate_diagonal_dominant

  !> Same matrix as engine_generate_diagonal_dominant but never stored: entries are generated on
  !> the fly inside the block matvec (matrix-free device operator).
  subroutine engine_set_hashed_operator(eng, which, sparsity, diag_val, seed)
    type(davidson_engine), intent(inout) :: eng
    integer, intent(in) :: which
    real(dp), intent(in) :: sparsity
    real(dp), intent(in), optional :: diag_val
    integer, intent(in), optional :: seed
    integer(c_int64_t) :: s
    real(c_double) :: dv
    s = 1
    if (present(seed)) s = int(seed, c_int64_t)
    dv = 0.0_dp
    if (present(diag_val)) dv = diag_val
    call check_dav(dav_set_operator_hashed(eng%h, int(which - 1, c_int), s, sparsity, &
         merge(1_c_int, 0_c_int, present(diag_val)), dv), "dav_set_operator_hashed")
    if (which == 1) eng%free_semantics = .true.
  end subroutine engine_set_hashed_operator

  !> The operators of the reference's matrix-free tests (src/tests/test_utils.f90:38-116) evaluated
  !> on the device: which=1 -> cos generator + i on the diagonal, which=2 -> sin generator, unit diagonal.
  subroutine engine_set_harness_operator(eng, which)
    type(davidson_engine), intent(inout) :: eng
    integer, intent(in) :: which
    real(dp), allocatable :: e(:)
    integer :: i
    allocate(e(eng%n))
    do i = 1, eng%n
       e(i) = exp(real(i) / real(eng%n))      ! single precision on purpose (test_utils.f90:82)
    end do
    call check_dav(dav_set_operator_harness(eng%h, int(which - 1, c_int), e), "dav_set_operator_harness")
    if (which == 1) eng%free_semantics = .true.
  end subroutine engine_set_harness_operator

  subroutine engine_set_identity(eng, which)
    type(davidson_engine), intent(inout) :: eng
    integer, intent(in) :: which
    call check_dav(dav_set_operator_identity(eng%h, int(which - 1, c_int)), "dav_set_operator_identity")
  end subroutine engine_set_identity

  !> The caller's OWN operator as a block apply on device memory: the device counterpart of the reference's matrix-free interface
  !> (src/davidson.f90:277-337 takes a procedure on host arrays).  `fn` = c_funloc of a bind(C) function with the signature
  !> dav_device_apply_fn of include/davidson_hip.h - it enqueues Y = Op(row0 : row0 + nloc, :) X on the stream it is handed (its own
  !> HIP kernels, hipBLAS, ...) - `ctx` is passed through to it, `diag` is the operator's diagonal (n entries).
  subroutine engine_set_device_operator(eng, which, fn, ctx, diag)
    type(davidson_engine), intent(inout) :: eng
    integer, intent(in) :: which
    type(c_funptr), intent(in) :: fn
    type(c_ptr), intent(in) :: ctx
    real(dp), intent(in) :: diag(:)
    if (size(diag) /= eng%n) then
       print *, "engine_set_device_operator: diag must have n entries"
       error stop
    end if
    call check_dav(dav_set_operator_device(eng%h, int(which - 1, c_int), fn, ctx, diag), "dav_set_operator_device")
  end subroutine engine_set_device_operator

end module davidson_engine_setup
