!> On-disk ingest as a Fortran user sees it: a matrix dumped in the reference's text format (list-directed,
!> one value per line, row i outer / column j inner - what write_matrix of src/tests/test_utils.f90 writes
!> and read_matrix reads) goes from the file to HBM through engine_read_matrix, in full and in
!> symmetric-tiled storage, and row block by row block through engine_dense_put_rows; every route must
!> give the eigenpairs of the in-memory solve.
program prog_ingest
  use numeric_kinds, only: dp
  use davidson, only: generalized_eigensolver
  use davidson_device
  use array_utils, only: generate_diagonal_dominant, norm
  implicit none
  integer, parameter :: dim = 300, lowest = 4
  real(dp) :: mtx(dim, dim), ev_mem(lowest), x_mem(dim, lowest), ev(lowest), x(dim, lowest), r(dim)
  real(dp), allocatable :: rows(:, :)
  type(davidson_engine) :: eng
  integer :: it_mem, it, i, j, u, nfail, r0, nr
  character(len=*), parameter :: path_text = "prog_ingest_matrix.txt", path_bin = "prog_ingest_matrix.f64"

  nfail = 0
  mtx = generate_diagonal_dominant(dim, 5d-3)
  call generalized_eigensolver(mtx, ev_mem, x_mem, lowest, "DPR", 1000, 1d-8, it_mem)

  open(newunit=u, file=path_text, status="replace")
  do i = 1, dim
     do j = 1, dim
        write(u, *) mtx(i, j)
     end do
  end do
  close(u)
  open(newunit=u, file=path_bin, status="replace", access="stream", form="unformatted")
  write(u) transpose(mtx)          ! row-major on disk
  close(u)

  ! text file -> full storage
  call engine_create(eng, dim, lowest)
  call engine_read_matrix(eng, 1, path_text)
  call generalized_eigensolver(eng, ev, x, lowest, "DPR", 1000, 1d-8, it)
  call engine_destroy(eng)
  call compare("text_full")

  ! text file -> symmetric-tiled storage
  call engine_create(eng, dim, lowest)
  call engine_set_storage(eng, "symmetric")
  call engine_read_matrix(eng, 1, path_text, "text")
  call generalized_eigensolver(eng, ev, x, lowest, "DPR", 1000, 1d-8, it)
  call engine_destroy(eng)
  call compare("text_symmetric")

  ! raw float64 -> full storage
  call engine_create(eng, dim, lowest)
  call engine_read_matrix(eng, 1, path_bin, "f64")
  call generalized_eigensolver(eng, ev, x, lowest, "DPR", 1000, 1d-8, it)
  call engine_destroy(eng)
  call compare("f64_full")

  ! producer that hands over blocks of rows, last block first
  call engine_create(eng, dim, lowest)
  call engine_dense_begin(eng, 1)
  r0 = dim + 1
  do while (r0 > 1)
     nr = min(77, r0 - 1)
     r0 = r0 - nr
     allocate(rows(dim, nr))
     rows = transpose(mtx(r0:r0 + nr - 1, :))
     call engine_dense_put_rows(eng, 1, r0, rows)
     deallocate(rows)
  end do
  call engine_dense_end(eng, 1)
  call generalized_eigensolver(eng, ev, x, lowest, "DPR", 1000, 1d-8, it)
  call engine_destroy(eng)
  call compare("put_rows")

  open(newunit=u, file=path_text, status="old"); close(u, status="delete")
  open(newunit=u, file=path_bin, status="old"); close(u, status="delete")
  print "(a, 4es24.16)", "EVALS", ev_mem
  if (nfail > 0) error stop 2

contains
  subroutine compare(name)
    character(len=*), intent(in) :: name
    integer :: k
    call check(name // "_eigenvalues", maxval(abs(ev - ev_mem)) < 1d-12)
    call check(name // "_iterations", it == it_mem)
    do k = 1, lowest
       r = matmul(mtx, x(:, k)) - ev(k) * x(:, k)
       call check(name // "_residual", norm(r) < 1d-8)
    end do
  end subroutine compare

  subroutine check(name, ok)
    character(len=*), intent(in) :: name
    logical, intent(in) :: ok
    print "(a, a, 1x, l1)", "CHECK ", name, ok
    if (.not. ok) nfail = nfail + 1
  end subroutine check
end program prog_ingest
