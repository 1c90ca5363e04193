!> Test support: matrix-free operators written against the public helper `free_matmul`, in the style
!> a user of the library writes them (row generators + block apply).  Same mathematical operator as
!> the reference's test harness: off-diagonal trig(log(sqrt(atan2(e_lo, e_hi)))) * 1e-4 with
!> e_i = exp(real(i)/real(n)) in single precision; A adds i on the diagonal, B has a unit diagonal.
module harness_ops
  use numeric_kinds, only: dp
  use davidson_free, only: free_matmul
  implicit none
contains

  function row_a(i, dim) result(vec)
    integer, intent(in) :: i
    integer, intent(in) :: dim
    real(dp), dimension(dim) :: vec
    vec = offdiag(i, dim, .true.)
    vec(i) = vec(i) + real(i)
  end function row_a

  function row_b(i, dim) result(vec)
    integer, intent(in) :: i
    integer, intent(in) :: dim
    real(dp), dimension(dim) :: vec
    vec = offdiag(i, dim, .false.)
    vec(i) = 1.0_dp
  end function row_b

  function offdiag(i, dim, use_cos) result(vec)
    integer, intent(in) :: i, dim
    logical, intent(in) :: use_cos
    real(dp), dimension(dim) :: vec
    real(dp) :: ei, ej, t
    integer :: j
    ei = exp(real(i) / real(dim))
    do j = 1, dim
       ej = exp(real(j) / real(dim))
       t = log(sqrt(atan2(merge(ei, ej, j >= i), merge(ej, ei, j >= i))))
       if (use_cos) then
          vec(j) = cos(t) * 1e-4
       else
          vec(j) = sin(t) * 1e-4
       end if
    end do
  end function offdiag

  function apply_a(input_vect) result(output_vect)
    real(dp), dimension(:, :), intent(in) :: input_vect
    real(dp), dimension(size(input_vect, 1), size(input_vect, 2)) :: output_vect
    output_vect = free_matmul(row_a, input_vect)
  end function apply_a

  function apply_b(input_vect) result(output_vect)
    real(dp), dimension(:, :), intent(in) :: input_vect
    real(dp), dimension(size(input_vect, 1), size(input_vect, 2)) :: output_vect
    output_vect = free_matmul(row_b, input_vect)
  end function apply_b

end module harness_ops
