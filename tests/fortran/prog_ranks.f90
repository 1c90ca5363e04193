!> Several ranks from Fortran: every process creates its engine with (rank, nranks), uploads the SAME host
!> matrix (the engine keeps the rank's row slab), joins the transport and calls the generic.  Launched as
!> `prog_ranks <rank> <nranks> <tag>`; the ranks share GPU 0 and exchange through the shared-memory test
!> transport (on a multi-GPU node: device = rank and engine_comm_init with the RCCL id instead).
program prog_ranks
  use iso_c_binding, only: c_ptr, c_int, c_char, c_null_char
  use numeric_kinds, only: dp
  use davidson, only: generalized_eigensolver
  use davidson_device
  use array_utils, only: generate_diagonal_dominant, norm
  implicit none
  integer, parameter :: dim = 700, lowest = 4
  real(dp), allocatable :: mtx(:, :), stx(:, :)
  real(dp) :: ev(lowest), x(dim, lowest), ev1(lowest), x1(dim, lowest), r(dim)
  type(davidson_engine) :: eng
  integer :: rank, nranks, it, it1, j, nfail
  character(len=64) :: arg, tag
  ! the shared-memory transport exists in the TEST build of libdavidson_hip.so only (csrc/davidson_hip_private.h): it is
  ! not part of the product's Fortran modules, the test binds it itself
  interface
     function dav_comm_init_shm(h, name) bind(C, name="dav_comm_init_shm") result(ierr)
       import :: c_ptr, c_int, c_char
       type(c_ptr), value :: h
       character(kind=c_char), intent(in) :: name(*)
       integer(c_int) :: ierr
     end function
  end interface

  call get_command_argument(1, arg); read (arg, *) rank
  call get_command_argument(2, arg); read (arg, *) nranks
  call get_command_argument(3, tag)
  nfail = 0
  allocate(mtx(dim, dim), stx(dim, dim))
  mtx = generate_diagonal_dominant(dim, 1d-2, seed=1)
  stx = generate_diagonal_dominant(dim, 1d-2, 1d0, 2)

  call engine_create(eng, dim, lowest, gev=.true., device=0, rank=rank, nranks=nranks)
  if (dav_comm_init_shm(eng%h, "/" // trim(tag) // c_null_char) /= 0) error stop "dav_comm_init_shm failed"
  call engine_set_dense(eng, 1, mtx)
  call engine_set_dense(eng, 2, stx)
  call generalized_eigensolver(eng, ev, x, lowest, "DPR", 200, 1d-8, it, 24)
  call engine_destroy(eng)

  do j = 1, lowest
     r = matmul(mtx, x(:, j)) - ev(j) * matmul(stx, x(:, j))
     call check("residual", norm(r) < 1d-8)
  end do
  if (rank == 0) then
     ! the single-rank answer, same process
     call generalized_eigensolver(mtx, ev1, x1, lowest, "DPR", 200, 1d-8, it1, 24, stx)
     call check("same_eigenvalues_as_one_rank", maxval(abs(ev - ev1)) < 1d-12)
     call check("same_iterations_as_one_rank", it == it1)
  end if
  print "(a, i0, a, i0, a, 4es24.16)", "RANK ", rank, " ITERS ", it, " EVALS", ev
  if (nfail > 0) error stop 2

contains
  subroutine check(name, ok)
    character(len=*), intent(in) :: name
    logical, intent(in) :: ok
    print "(a, a, 1x, l1)", "CHECK ", name, ok
    if (.not. ok) nfail = nfail + 1
  end subroutine check
end program prog_ranks
