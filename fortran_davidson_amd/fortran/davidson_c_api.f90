!> bind(C) doors into the Fortran API so that non-Fortran callers (the pytest suite and bench.py
!> through ctypes) exercise exactly what a Fortran user calls: `generalized_eigensolver` of module
!> `davidson` and the helper modules.  Nothing here computes; it only forwards.
module davidson_c_api
  use, intrinsic :: iso_c_binding
  use numeric_kinds, only: dp
  use davidson, only: generalized_eigensolver
  use davidson_device
  use davidson_free, only: free_matmul
  use lapack_wrapper
  use array_utils
  implicit none

  abstract interface
     subroutine c_block_apply(n, k, x, y) bind(C)
       import :: c_int, c_double
       integer(c_int), value :: n, k
       real(c_double), intent(in) :: x(n, k)
       real(c_double), intent(out) :: y(n, k)
     end subroutine c_block_apply
  end interface
  procedure(c_block_apply), pointer, save :: cb_a => null(), cb_b => null()

contains

  function method_name(code) result(name)
    integer(c_int), intent(in) :: code
    character(len=3) :: name
    name = "DPR"
    if (code == 1) name = "GJD"
    if (code > 1) name = "XXX"
  end function method_name

  !> generalized_eigensolver(matrix, ...) - dense specific.  max_dim < 0: argument absent.
  subroutine fd_dense_solve(n, a, has_b, b, lowest, method, max_it, tol, max_dim, evals, evecs, iters) &
       bind(C, name="fd_dense_solve")
    integer(c_int), value :: n, has_b, lowest, method, max_it, max_dim
    real(c_double), value :: tol
    real(c_double), intent(in) :: a(n, n), b(n, *)
    real(c_double), intent(out) :: evals(lowest), evecs(n, lowest)
    integer(c_int), intent(out) :: iters
    integer :: it
    if (has_b /= 0) then
       if (max_dim >= 0) then
          call generalized_eigensolver(a, evals, evecs, lowest, method_name(method), max_it, tol, it, max_dim, b(:, 1:n))
       else
          call generalized_eigensolver(a, evals, evecs, lowest, method_name(method), max_it, tol, it, &
               second_matrix=b(:, 1:n))
       end if
    else
       if (max_dim >= 0) then
          call generalized_eigensolver(a, evals, evecs, lowest, method_name(method), max_it, tol, it, max_dim)
       else
          call generalized_eigensolver(a, evals, evecs, lowest, method_name(method), max_it, tol, it)
       end if
    end if
    iters = it
  end subroutine fd_dense_solve

  function apply_cb_a(input_vect) result(output_vect)
    real(dp), dimension(:, :), intent(in) :: input_vect
    real(dp), dimension(size(input_vect, 1), size(input_vect, 2)) :: output_vect
    real(dp), allocatable :: x(:, :)
    x = input_vect
    call cb_a(int(size(x, 1), c_int), int(size(x, 2), c_int), x, output_vect)
  end function apply_cb_a

  function apply_cb_b(input_vect) result(output_vect)
    real(dp), dimension(:, :), intent(in) :: input_vect
    real(dp), dimension(size(input_vect, 1), size(input_vect, 2)) :: output_vect
    real(dp), allocatable :: x(:, :)
    x = input_vect
    call cb_b(int(size(x, 1), c_int), int(size(x, 2), c_int), x, output_vect)
  end function apply_cb_b

  !> generalized_eigensolver(fun_A, ..., fun_B) - matrix-free specific with C callbacks.
  subroutine fd_free_solve(n, fa, fb, lowest, max_it, tol, max_dim, evals, evecs, iters) bind(C, name="fd_free_solve")
    integer(c_int), value :: n, lowest, max_it, max_dim
    type(c_funptr), value :: fa, fb
    real(c_double), value :: tol
    real(c_double), intent(out) :: evals(lowest), evecs(n, lowest)
    integer(c_int), intent(out) :: iters
    integer :: it
    call c_f_procpointer(fa, cb_a)
    call c_f_procpointer(fb, cb_b)
    call generalized_eigensolver(apply_cb_a, evals, evecs, lowest, "DPR", max_it, tol, it, max_dim, apply_cb_b)
    iters = it
  end subroutine fd_free_solve

  ! ---- device-resident engine ---------------------------------------------------------------------
  function fd_engine_create(n, lowest, max_dim, gev, device, rank, nranks) result(p) bind(C, name="fd_engine_create")
    integer(c_int), value :: n, lowest, max_dim, gev, device, rank, nranks
    type(c_ptr) :: p
    type(davidson_engine), pointer :: eng
    allocate(eng)
    if (max_dim >= 0) then
       call engine_create(eng, n, lowest, max_dim, gev /= 0, device, rank, nranks)
    else
       call engine_create(eng, n, lowest, gev=gev /= 0, device=device, rank=rank, nranks=nranks)
    end if
    p = c_loc(eng)
  end function fd_engine_create

  subroutine fd_engine_destroy(p) bind(C, name="fd_engine_destroy")
    type(c_ptr), value :: p
    type(davidson_engine), pointer :: eng
    call c_f_pointer(p, eng)
    call engine_destroy(eng)
    deallocate(eng)
  end subroutine fd_engine_destroy

  !> raw C handle (for dav_get_stats / dav_bench_apply from the harness)
  function fd_engine_handle(p) result(h) bind(C, name="fd_engine_handle")
    type(c_ptr), value :: p
    type(c_ptr) :: h
    type(davidson_engine), pointer :: eng
    call c_f_pointer(p, eng)
    h = eng%h
  end function fd_engine_handle

  subroutine fd_engine_comm_unique_id(id) bind(C, name="fd_engine_comm_unique_id")
    character(kind=c_char), intent(out) :: id(128)
    call engine_comm_unique_id(id)
  end subroutine fd_engine_comm_unique_id

  subroutine fd_engine_comm_init(p, id) bind(C, name="fd_engine_comm_init")
    type(c_ptr), value :: p
    character(kind=c_char), intent(in) :: id(128)
    type(davidson_engine), pointer :: eng
    call c_f_pointer(p, eng)
    call engine_comm_init(eng, id)
  end subroutine fd_engine_comm_init

  subroutine fd_engine_set_storage(p, mode) bind(C, name="fd_engine_set_storage")
    type(c_ptr), value :: p
    integer(c_int), value :: mode
    type(davidson_engine), pointer :: eng
    call c_f_pointer(p, eng)
    if (mode == 1) then
       call engine_set_storage(eng, "symmetric")
    else
       call engine_set_storage(eng, "full")
    end if
  end subroutine fd_engine_set_storage

  subroutine fd_engine_set_inner_precision(p, bits) bind(C, name="fd_engine_set_inner_precision")
    type(c_ptr), value :: p
    integer(c_int), value :: bits
    type(davidson_engine), pointer :: eng
    call c_f_pointer(p, eng)
    call engine_set_inner_precision(eng, int(bits))
  end subroutine fd_engine_set_inner_precision

  subroutine fd_engine_set_device_rr(p, on) bind(C, name="fd_engine_set_device_rr")
    type(c_ptr), value :: p
    integer(c_int), value :: on
    type(davidson_engine), pointer :: eng
    call c_f_pointer(p, eng)
    call engine_set_device_rr(eng, on /= 0)
  end subroutine fd_engine_set_device_rr

  subroutine fd_engine_set_policy(p, code) bind(C, name="fd_engine_set_policy")
    type(c_ptr), value :: p
    integer(c_int), value :: code
    type(davidson_engine), pointer :: eng
    call c_f_pointer(p, eng)
    if (code == 1) then
       call engine_set_correction_policy(eng, "unconverged")
    else if (code == 2) then
       call engine_set_correction_policy(eng, "locking")
    else
       call engine_set_correction_policy(eng, "all")
    end if
  end subroutine fd_engine_set_policy

  subroutine fd_engine_set_dense(p, which, a) bind(C, name="fd_engine_set_dense")
    type(c_ptr), value :: p
    integer(c_int), value :: which
    real(c_double), intent(in), target :: a(*)
    type(davidson_engine), pointer :: eng
    real(c_double), pointer :: mat(:, :)
    call c_f_pointer(p, eng)
    call c_f_pointer(c_loc(a), mat, [eng%n, eng%n])
    call engine_set_dense(eng, int(which), mat)
  end subroutine fd_engine_set_dense

  !> kind 0: dense generated in HBM, 1: hashed matrix-free operator, 2: harness operator, 3: identity
  subroutine fd_engine_set_operator(p, which, kind, seed, sparsity, use_diag_val, diag_val) &
       bind(C, name="fd_engine_set_operator")
    type(c_ptr), value :: p
    integer(c_int), value :: which, kind, seed, use_diag_val
    real(c_double), value :: sparsity, diag_val
    type(davidson_engine), pointer :: eng
    call c_f_pointer(p, eng)
    select case (kind)
    case (0)
       if (use_diag_val /= 0) then
          call engine_generate_diagonal_dominant(eng, int(which), sparsity, diag_val, int(seed))
       else
          call engine_generate_diagonal_dominant(eng, int(which), sparsity, seed=int(seed))
       end if
    case (1)
       if (use_diag_val /= 0) then
          call engine_set_hashed_operator(eng, int(which), sparsity, diag_val, int(seed))
       else
          call engine_set_hashed_operator(eng, int(which), sparsity, seed=int(seed))
       end if
    case (2)
       call engine_set_harness_operator(eng, int(which))
    case default
       call engine_set_identity(eng, int(which))
    end select
  end subroutine fd_engine_set_operator

  !> generalized_eigensolver(engine, ...) - device specific.  want_vectors = 0 leaves X in HBM.
  subroutine fd_engine_solve(p, lowest, method, max_it, tol, max_dim, evals, want_vectors, evecs, iters) &
       bind(C, name="fd_engine_solve")
    type(c_ptr), value :: p
    integer(c_int), value :: lowest, method, max_it, max_dim, want_vectors
    real(c_double), value :: tol
    real(c_double), intent(out) :: evals(lowest)
    real(c_double), intent(out), target :: evecs(*)
    integer(c_int), intent(out) :: iters
    type(davidson_engine), pointer :: eng
    real(c_double), pointer :: vec(:, :)
    integer :: it, md
    call c_f_pointer(p, eng)
    md = max_dim
    if (md < 0) md = 10 * lowest
    if (want_vectors /= 0) then
       call c_f_pointer(c_loc(evecs), vec, [eng%n, int(lowest)])
       call generalized_eigensolver(eng, evals, vec, lowest, method_name(method), max_it, tol, it, md)
    else
       call generalized_eigensolver(eng, eigenvalues=evals, lowest=lowest, method=method_name(method), &
            max_iterations=max_it, tolerance=tol, iters=it, max_dim_sub=md)
    end if
    iters = it
  end subroutine fd_engine_solve

  !> wall time of the last solve on this engine by phase (davidson_engine%phase_seconds)
  subroutine fd_engine_phase_seconds(p, out) bind(C, name="fd_engine_phase_seconds")
    type(c_ptr), value :: p
    real(c_double), intent(out) :: out(8)
    type(davidson_engine), pointer :: eng
    call c_f_pointer(p, eng)
    out = eng%phase_seconds
  end subroutine fd_engine_phase_seconds

  !> what the dense front end decides before it uploads (davidson_device: fits_as_full_rows): 1 = nmat matrices of order n fit the
  !> engine's device as full rows, 0 = the front end switches to symmetric tiles
  function fd_engine_fits_as_full_rows(p, n, nmat) bind(C, name="fd_engine_fits_as_full_rows") result(fits)
    type(c_ptr), value :: p
    integer(c_int), value :: n, nmat
    integer(c_int) :: fits
    type(davidson_engine), pointer :: eng
    call c_f_pointer(p, eng)
    fits = merge(1_c_int, 0_c_int, fits_as_full_rows(eng, int(n), int(nmat)))
  end function fd_engine_fits_as_full_rows

  ! ---- helper modules (unit tests mirror src/tests/test_call_lapack.f90) ----------------------------
  subroutine fd_lapack_eigensolver(n, mtx, has_stx, stx, evals, evecs) bind(C, name="fd_lapack_eigensolver")
    integer(c_int), value :: n, has_stx
    real(c_double), intent(in) :: mtx(n, n), stx(n, *)
    real(c_double), intent(out) :: evals(n), evecs(n, n)
    if (has_stx /= 0) then
       call lapack_generalized_eigensolver(mtx, evals, evecs, stx(:, 1:n))
    else
       call lapack_generalized_eigensolver(mtx, evals, evecs)
    end if
  end subroutine fd_lapack_eigensolver

  subroutine fd_lapack_rayleigh_ritz(n, mtx, has_stx, stx, nvec, evals, evecs) bind(C, name="fd_lapack_rayleigh_ritz")
    integer(c_int), value :: n, has_stx, nvec
    real(c_double), intent(in) :: mtx(n, n), stx(n, *)
    real(c_double), intent(out) :: evals(n), evecs(n, n)
    if (has_stx /= 0) then
       call lapack_rayleigh_ritz(mtx, evals, evecs, int(nvec), stx(:, 1:n))
    else
       call lapack_rayleigh_ritz(mtx, evals, evecs, int(nvec))
    end if
  end subroutine fd_lapack_rayleigh_ritz

  subroutine fd_lapack_qr(m, n, basis) bind(C, name="fd_lapack_qr")
    integer(c_int), value :: m, n
    real(c_double), intent(inout) :: basis(m, n)
    call lapack_qr(basis)
  end subroutine fd_lapack_qr

  subroutine fd_lapack_solver(n, arr, brr) bind(C, name="fd_lapack_solver")
    integer(c_int), value :: n
    real(c_double), intent(inout) :: arr(n, n), brr(n, 1)
    call lapack_solver(arr, brr)
  end subroutine fd_lapack_solver

  subroutine fd_lapack_matmul(ta, tb, m, k, n, a, b, c) bind(C, name="fd_lapack_matmul")
    integer(c_int), value :: ta, tb, m, k, n
    real(c_double), intent(in), target :: a(*), b(*)
    real(c_double), intent(out) :: c(m, n)
    real(c_double), pointer :: am(:, :), bm(:, :)
    character(len=1) :: ca, cb
    ca = merge('T', 'N', ta /= 0)
    cb = merge('T', 'N', tb /= 0)
    if (ta /= 0) then
       call c_f_pointer(c_loc(a), am, [k, m])
    else
       call c_f_pointer(c_loc(a), am, [m, k])
    end if
    if (tb /= 0) then
       call c_f_pointer(c_loc(b), bm, [n, k])
    else
       call c_f_pointer(c_loc(b), bm, [k, n])
    end if
    c = lapack_matmul(ca, cb, am, bm)
  end subroutine fd_lapack_matmul

  subroutine fd_lapack_sort(n, decreasing, vector, keys) bind(C, name="fd_lapack_sort")
    integer(c_int), value :: n, decreasing
    real(c_double), intent(inout) :: vector(n)
    integer(c_int), intent(out) :: keys(n)
    keys = lapack_sort(merge('D', 'I', decreasing /= 0), vector)
  end subroutine fd_lapack_sort

  subroutine fd_generate_preconditioner(n, diag, dim_sub, precond) bind(C, name="fd_generate_preconditioner")
    integer(c_int), value :: n, dim_sub
    real(c_double), intent(inout) :: diag(n)
    real(c_double), intent(out) :: precond(n, dim_sub)
    precond = generate_preconditioner(diag, dim_sub)
  end subroutine fd_generate_preconditioner

  subroutine fd_generate_diagonal_dominant(n, sparsity, use_diag_val, diag_val, seed, arr) &
       bind(C, name="fd_generate_diagonal_dominant")
    integer(c_int), value :: n, use_diag_val, seed
    real(c_double), value :: sparsity, diag_val
    real(c_double), intent(out) :: arr(n, n)
    real(dp) :: sp, dv
    sp = sparsity
    dv = diag_val
    if (use_diag_val /= 0) then
       arr = generate_diagonal_dominant(n, sp, dv, int(seed))
    else
       arr = generate_diagonal_dominant(n, sp, seed=int(seed))
    end if
  end subroutine fd_generate_diagonal_dominant

  subroutine fd_norm(n, v, res) bind(C, name="fd_norm")
    integer(c_int), value :: n
    real(c_double), intent(in) :: v(n)
    real(c_double), intent(out) :: res
    res = norm(v)
  end subroutine fd_norm

end module davidson_c_api
