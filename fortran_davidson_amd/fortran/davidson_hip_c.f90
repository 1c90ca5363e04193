!> ISO_C_BINDING view of include/davidson_hip.h - the thin boundary between the Fortran host and the
!> HIP engine (libdavidson_hip.so).  One interface per C entry point, same order as the header.
module davidson_hip_c
  use, intrinsic :: iso_c_binding
  implicit none
  public

  integer(c_int), parameter :: DAV_OP_A = 0, DAV_OP_B = 1
  integer(c_int), parameter :: DAV_PANEL_V = 0, DAV_PANEL_W = 1, DAV_PANEL_BV = 2, DAV_PANEL_X = 3, &
       DAV_PANEL_R = 4, DAV_PANEL_S = 5
  integer(c_int), parameter :: DAV_METHOD_DPR = 0, DAV_METHOD_GJD = 1, DAV_METHOD_NONE = 2
  !> dav_panel_unit_column: "the engine keeps no such entry of the start order" (not an error)
  integer(c_int), parameter :: DAV_NO_SUCH_ENTRY = 2

  type, bind(C) :: dav_stats
     integer(c_int64_t) :: n, nloc
     integer(c_int32_t) :: nranks, rank, m, applies
     integer(c_int64_t) :: apply_cols
     real(c_double) :: apply_ms, apply_bytes, last_apply_ms, last_apply_bytes, gram_ms, panel_ms, comm_ms
     real(c_double) :: apply_kernel_ms, apply_flops
     integer(c_int64_t) :: apply_launches
     integer(c_int64_t) :: restarts
     real(c_double) :: allgather_ms, reduce_scatter_ms, allreduce_ms
     real(c_double) :: allgather_bytes, reduce_scatter_bytes, allreduce_bytes
     integer(c_int64_t) :: collectives
     integer(c_int32_t) :: comm_ranks, comm_overlap
     real(c_double) :: apply_comm_ms
     real(c_double) :: b_stored_kernel_ms, b_stored_bytes, b_stored_flops, b_generated_kernel_ms, b_generated_entries, b_generated_flops
     integer(c_int64_t) :: b_stored_launches, b_generated_launches
  end type dav_stats
  !> DAV_HIP_ABI_VERSION of include/davidson_hip.h these interfaces were written against: engine_create checks that the
  !> loaded libdavidson_hip.so reports the same number (the layout of dav_stats grew in 101, 102 and 104)
  integer(c_int), parameter :: DAV_HIP_ABI_VERSION = 108

  interface
     function dav_last_error() bind(C, name="dav_last_error") result(p)
       import :: c_ptr
       type(c_ptr) :: p
     end function
     function dav_version() bind(C, name="dav_version") result(v)
       import :: c_int
       integer(c_int) :: v
     end function
     function dav_create(h, device, n, max_cols, gev, rank, nranks) bind(C, name="dav_create") result(ierr)
       import :: c_ptr, c_int, c_int64_t
       type(c_ptr), intent(out) :: h
       integer(c_int), value :: device, max_cols, gev, rank, nranks
       integer(c_int64_t), value :: n
       integer(c_int) :: ierr
     end function
     function dav_destroy(h) bind(C, name="dav_destroy") result(ierr)
       import :: c_ptr, c_int
       type(c_ptr), value :: h
       integer(c_int) :: ierr
     end function
     function dav_comm_unique_id(id) bind(C, name="dav_comm_unique_id") result(ierr)
       import :: c_char, c_int
       character(kind=c_char), intent(out) :: id(128)
       integer(c_int) :: ierr
     end function
     function dav_comm_init(h, id) bind(C, name="dav_comm_init") result(ierr)
       import :: c_ptr, c_char, c_int
       type(c_ptr), value :: h
       character(kind=c_char), intent(in) :: id(128)
       integer(c_int) :: ierr
     end function
     function dav_synchronize(h) bind(C, name="dav_synchronize") result(ierr)
       import :: c_ptr, c_int
       type(c_ptr), value :: h
       integer(c_int) :: ierr
     end function
     function dav_get_stats(h, st) bind(C, name="dav_get_stats") result(ierr)
       import :: c_ptr, c_int, dav_stats
       type(c_ptr), value :: h
       type(dav_stats), intent(out) :: st
       integer(c_int) :: ierr
     end function
     function dav_device_memory(h, free_bytes, total_bytes) bind(C, name="dav_device_memory") result(ierr)
       import :: c_ptr, c_int, c_int64_t
       type(c_ptr), value :: h
       integer(c_int64_t), intent(out) :: free_bytes, total_bytes
       integer(c_int) :: ierr
     end function
     function dav_free_buffers() bind(C, name="dav_free_buffers") result(ierr)
       import :: c_int
       integer(c_int) :: ierr
     end function
     function dav_set_storage(h, mode) bind(C, name="dav_set_storage") result(ierr)
       import :: c_ptr, c_int
       type(c_ptr), value :: h
       integer(c_int), value :: mode
       integer(c_int) :: ierr
     end function
     function dav_set_dense_host(h, which, a, lda) bind(C, name="dav_set_dense_host") result(ierr)
       import :: c_ptr, c_int, c_int64_t, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: which
       real(c_double), intent(in) :: a(*)
       integer(c_int64_t), value :: lda
       integer(c_int) :: ierr
     end function
     function dav_dense_begin(h, which) bind(C, name="dav_dense_begin") result(ierr)
       import :: c_ptr, c_int
       type(c_ptr), value :: h
       integer(c_int), value :: which
       integer(c_int) :: ierr
     end function
     function dav_dense_put_rows(h, which, row0, nrows, rows, ldr) bind(C, name="dav_dense_put_rows") result(ierr)
       import :: c_ptr, c_int, c_int64_t, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: which
       integer(c_int64_t), value :: row0, nrows, ldr
       real(c_double), intent(in) :: rows(*)
       integer(c_int) :: ierr
     end function
     function dav_dense_end(h, which) bind(C, name="dav_dense_end") result(ierr)
       import :: c_ptr, c_int
       type(c_ptr), value :: h
       integer(c_int), value :: which
       integer(c_int) :: ierr
     end function
     function dav_set_dense_file(h, which, path, fmt) bind(C, name="dav_set_dense_file") result(ierr)
       import :: c_ptr, c_int, c_char
       type(c_ptr), value :: h
       integer(c_int), value :: which, fmt
       character(kind=c_char), intent(in) :: path(*)
       integer(c_int) :: ierr
     end function
     function dav_set_dense_generated(h, which, seed, sparsity, use_diag_val, diag_val) &
          bind(C, name="dav_set_dense_generated") result(ierr)
       import :: c_ptr, c_int, c_int64_t, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: which, use_diag_val
       integer(c_int64_t), value :: seed
       real(c_double), value :: sparsity, diag_val
       integer(c_int) :: ierr
     end function
     function dav_set_operator_hashed(h, which, seed, sparsity, use_diag_val, diag_val) &
          bind(C, name="dav_set_operator_hashed") result(ierr)
       import :: c_ptr, c_int, c_int64_t, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: which, use_diag_val
       integer(c_int64_t), value :: seed
       real(c_double), value :: sparsity, diag_val
       integer(c_int) :: ierr
     end function
     function dav_set_operator_harness(h, which, e_table) bind(C, name="dav_set_operator_harness") result(ierr)
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: which
       real(c_double), intent(in) :: e_table(*)
       integer(c_int) :: ierr
     end function
     function dav_set_operator_identity(h, which) bind(C, name="dav_set_operator_identity") result(ierr)
       import :: c_ptr, c_int
       type(c_ptr), value :: h
       integer(c_int), value :: which
       integer(c_int) :: ierr
     end function
     !> the caller's own block apply on device memory (include/davidson_hip.h: dav_device_apply_fn); fn = c_funloc of a bind(C) function
     function dav_set_operator_device(h, which, fn, ctx, diag) bind(C, name="dav_set_operator_device") result(ierr)
       import :: c_ptr, c_int, c_funptr, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: which
       type(c_funptr), value :: fn
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: diag(*)
       integer(c_int) :: ierr
     end function
     function dav_set_operator_host(h, which, diag) bind(C, name="dav_set_operator_host") result(ierr)
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: which
       real(c_double), intent(in) :: diag(*)
       integer(c_int) :: ierr
     end function
     function dav_get_diagonal(h, which, diag) bind(C, name="dav_get_diagonal") result(ierr)
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: which
       real(c_double), intent(out) :: diag(*)
       integer(c_int) :: ierr
     end function
     function dav_init_basis(h, ncols, idx) bind(C, name="dav_init_basis") result(ierr)
       import :: c_ptr, c_int, c_int64_t
       type(c_ptr), value :: h
       integer(c_int), value :: ncols
       integer(c_int64_t), intent(out) :: idx(*)
       integer(c_int) :: ierr
     end function
     function dav_apply(h, which, src_panel, c0, k, dst_panel, d0) bind(C, name="dav_apply") result(ierr)
       import :: c_ptr, c_int
       type(c_ptr), value :: h
       integer(c_int), value :: which, src_panel, c0, k, dst_panel, d0
       integer(c_int) :: ierr
     end function
     function dav_gram(h, panel_p, p0, p, panel_q, q0, q, out, ldo) bind(C, name="dav_gram") result(ierr)
       import :: c_ptr, c_int, c_int64_t, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: panel_p, p0, p, panel_q, q0, q
       real(c_double), intent(out) :: out(*)
       integer(c_int64_t), value :: ldo
       integer(c_int) :: ierr
     end function
     function dav_project(h, c0, k, hm, ldh, sm, lds) bind(C, name="dav_project") result(ierr)
       import :: c_ptr, c_int, c_int64_t, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: c0, k
       real(c_double), intent(inout) :: hm(*), sm(*)
       integer(c_int64_t), value :: ldh, lds
       integer(c_int) :: ierr
     end function
     function dav_ritz_residual_correction_n(h, m, ncorr, lowest, y, ldy, theta, method, resnorm) &
          bind(C, name="dav_ritz_residual_correction_n") result(ierr)
       import :: c_ptr, c_int, c_int64_t, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: m, ncorr, lowest, method
       real(c_double), intent(in) :: y(*), theta(*)
       integer(c_int64_t), value :: ldy
       real(c_double), intent(out) :: resnorm(*)
       integer(c_int) :: ierr
     end function
     function dav_ritz_residual_correction_g(h, m, ncorr, lowest, y, ldy, theta, resnorm, c, ldc, g, ldg) &
          bind(C, name="dav_ritz_residual_correction_g") result(ierr)
       import :: c_ptr, c_int, c_int64_t, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: m, ncorr, lowest
       real(c_double), intent(in) :: y(*), theta(*)
       integer(c_int64_t), value :: ldy, ldc, ldg
       real(c_double), intent(out) :: resnorm(*), c(*), g(*)
       integer(c_int) :: ierr
     end function
     function dav_set_lazy_ritz_vectors(h, on) bind(C, name="dav_set_lazy_ritz_vectors") result(ierr)
       import :: c_ptr, c_int
       type(c_ptr), value :: h
       integer(c_int), value :: on
       integer(c_int) :: ierr
     end function
     function dav_ritz_vectors(h, m, nx, y, ldy) bind(C, name="dav_ritz_vectors") result(ierr)
       import :: c_ptr, c_int, c_int64_t, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: m, nx
       real(c_double), intent(in) :: y(*)
       integer(c_int64_t), value :: ldy
       integer(c_int) :: ierr
     end function
     function dav_panel_select(h, panel, c0, nsel, sel) bind(C, name="dav_panel_select") result(ierr)
       import :: c_ptr, c_int
       type(c_ptr), value :: h
       integer(c_int), value :: panel, c0, nsel
       integer(c_int), intent(in) :: sel(*)
       integer(c_int) :: ierr
     end function
     function dav_gjd_correction_n(h, m, ncols, theta, max_inner, inner_tol, tol_per_col, inner_iters) &
          bind(C, name="dav_gjd_correction_n") result(ierr)
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: m, ncols, max_inner
       real(c_double), intent(in) :: theta(*), tol_per_col(*)
       real(c_double), value :: inner_tol
       integer(c_int), intent(out) :: inner_iters
       integer(c_int) :: ierr
     end function
     function dav_ritz_residual_correction(h, m, lowest, y, ldy, theta, method, resnorm) &
          bind(C, name="dav_ritz_residual_correction") result(ierr)
       import :: c_ptr, c_int, c_int64_t, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: m, lowest, method
       real(c_double), intent(in) :: y(*), theta(*)
       integer(c_int64_t), value :: ldy
       real(c_double), intent(out) :: resnorm(*)
       integer(c_int) :: ierr
     end function
     function dav_gjd_correction(h, m, theta, max_inner, inner_tol, inner_iters) &
          bind(C, name="dav_gjd_correction") result(ierr)
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: m, max_inner
       real(c_double), intent(in) :: theta(*)
       real(c_double), value :: inner_tol
       integer(c_int), intent(out) :: inner_iters
       integer(c_int) :: ierr
     end function
     function dav_ortho_gram(h, m, kt, c, ldc, g, ldg) bind(C, name="dav_ortho_gram") result(ierr)
       import :: c_ptr, c_int, c_int64_t, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: m, kt
       real(c_double), intent(out) :: c(*), g(*)
       integer(c_int64_t), value :: ldc, ldg
       integer(c_int) :: ierr
     end function
     function dav_ortho_apply(h, m, kt, c, ldc, mm, ldm) bind(C, name="dav_ortho_apply") result(ierr)
       import :: c_ptr, c_int, c_int64_t, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: m, kt
       real(c_double), intent(in) :: c(*), mm(*)
       integer(c_int64_t), value :: ldc, ldm
       integer(c_int) :: ierr
     end function
     function dav_ortho_apply_all(h, m, kt, c, ldc, mm, ldm) bind(C, name="dav_ortho_apply_all") result(ierr)
       import :: c_ptr, c_int, c_int64_t, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: m, kt
       integer(c_int64_t), value :: ldc, ldm
       real(c_double), intent(in) :: c(*), mm(*)
       integer(c_int) :: ierr
     end function
     function dav_project_ortho(h, m, k, hraw, ldh, sraw, lds, c, ldc, g, ldg) bind(C, name="dav_project_ortho") result(ierr)
       import :: c_ptr, c_int, c_int64_t, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: m, k
       integer(c_int64_t), value :: ldh, lds, ldc, ldg
       real(c_double), intent(out) :: hraw(*), c(*), g(*)
       type(c_ptr), value :: sraw
       integer(c_int) :: ierr
     end function
     function dav_expand(h, m, kt) bind(C, name="dav_expand") result(ierr)
       import :: c_ptr, c_int
       type(c_ptr), value :: h
       integer(c_int), value :: m, kt
       integer(c_int) :: ierr
     end function
     function dav_ranks_agree(h, words, nwords) bind(C, name="dav_ranks_agree") result(ierr)
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h
       real(c_double), intent(in) :: words(*)
       integer(c_int), value :: nwords
       integer(c_int) :: ierr
     end function
     function dav_agree_inputs(h, words, nwords) bind(C, name="dav_agree_inputs") result(ierr)
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h
       real(c_double), intent(in) :: words(*)
       integer(c_int), value :: nwords
       integer(c_int) :: ierr
     end function
     function dav_agree_next(h, words, nwords) bind(C, name="dav_agree_next") result(ierr)
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: h
       real(c_double), intent(in) :: words(*)
       integer(c_int), value :: nwords
       integer(c_int) :: ierr
     end function
     function dav_set_inner_precision(h, bits) bind(C, name="dav_set_inner_precision") result(ierr)
       import :: c_ptr, c_int
       type(c_ptr), value :: h
       integer(c_int), value :: bits
       integer(c_int) :: ierr
     end function
     !> device-resident Rayleigh-Ritz (opt-in): the projected matrices, Ritz values and vectors stay in HBM
     function dav_rr_enable(h, on) bind(C, name="dav_rr_enable") result(ierr)
       import :: c_ptr, c_int
       type(c_ptr), value :: h
       integer(c_int), value :: on
       integer(c_int) :: ierr
     end function
     function dav_project_dev(h, c0, k) bind(C, name="dav_project_dev") result(ierr)
       import :: c_ptr, c_int
       type(c_ptr), value :: h
       integer(c_int), value :: c0, k
       integer(c_int) :: ierr
     end function
     function dav_rr_ritz(h, m, ncorr, lowest, method, theta, resnorm, c, ldc, g, ldg, sweeps) &
          bind(C, name="dav_rr_ritz") result(ierr)
       import :: c_ptr, c_int, c_int64_t, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: m, ncorr, lowest, method
       real(c_double), intent(out) :: theta(*), resnorm(*)
       type(c_ptr), value :: c, g                       ! c_null_ptr: no Gram blocks wanted
       integer(c_int64_t), value :: ldc, ldg
       integer(c_int), intent(out) :: sweeps
       integer(c_int) :: ierr
     end function
     function dav_rr_restart(h, m, keep) bind(C, name="dav_rr_restart") result(ierr)
       import :: c_ptr, c_int
       type(c_ptr), value :: h
       integer(c_int), value :: m, keep
       integer(c_int) :: ierr
     end function
     function dav_rr_get(h, m, ncols, theta, y, ldy) bind(C, name="dav_rr_get") result(ierr)
       import :: c_ptr, c_int, c_int64_t, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: m, ncols
       integer(c_int64_t), value :: ldy
       real(c_double), intent(out) :: theta(*), y(ldy, *)
       integer(c_int) :: ierr
     end function
     function dav_restart(h, m, keep, yk, ldy) bind(C, name="dav_restart") result(ierr)
       import :: c_ptr, c_int, c_int64_t, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: m, keep
       real(c_double), intent(in) :: yk(*)
       integer(c_int64_t), value :: ldy
       integer(c_int) :: ierr
     end function
     function dav_panel_transform(h, src_panel, s0, p, mm, ldm, q, dst_panel, d0) &
          bind(C, name="dav_panel_transform") result(ierr)
       import :: c_ptr, c_int, c_int64_t, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: src_panel, s0, p, q, dst_panel, d0
       real(c_double), intent(in) :: mm(*)
       integer(c_int64_t), value :: ldm
       integer(c_int) :: ierr
     end function
     function dav_panel_get(h, panel, c0, k, out, ld) bind(C, name="dav_panel_get") result(ierr)
       import :: c_ptr, c_int, c_int64_t, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: panel, c0, k
       real(c_double), intent(out) :: out(*)
       integer(c_int64_t), value :: ld
       integer(c_int) :: ierr
     end function
     function dav_panel_unit_column(h, panel, col, k) bind(C, name="dav_panel_unit_column") result(ierr)
       import :: c_ptr, c_int
       type(c_ptr), value :: h
       integer(c_int), value :: panel, col, k
       integer(c_int) :: ierr
     end function
     function dav_panel_put(h, panel, c0, k, in, ld) bind(C, name="dav_panel_put") result(ierr)
       import :: c_ptr, c_int, c_int64_t, c_double
       type(c_ptr), value :: h
       integer(c_int), value :: panel, c0, k
       real(c_double), intent(in) :: in(*)
       integer(c_int64_t), value :: ld
       integer(c_int) :: ierr
     end function
     function c_strlen(s) bind(C, name="strlen") result(n)
       import :: c_ptr, c_size_t
       type(c_ptr), value :: s
       integer(c_size_t) :: n
     end function
  end interface

contains

  !> Error convention of the boundary: non-zero status -> print and `error stop`, as
  !> check_lapack_call does for LAPACK (src/lapack_wrapper.f90:395-408).
  subroutine check_dav(ierr, name)
    integer(c_int), intent(in) :: ierr
    character(len=*), intent(in) :: name
    type(c_ptr) :: p
    character(kind=c_char), pointer :: s(:)
    integer :: n, i
    character(len=:), allocatable :: msg
    if (ierr == 0) return
    p = dav_last_error()
    msg = ""
    if (c_associated(p)) then
       n = int(c_strlen(p))
       call c_f_pointer(p, s, [n])
       allocate(character(len=n) :: msg)
       do i = 1, n
          msg(i:i) = s(i)
       end do
    end if
    print *, "call to HIP engine: ", name, " has failed!"
    print *, "info: ", ierr, " ", msg
    error stop
  end subroutine check_dav

end module davidson_hip_c
