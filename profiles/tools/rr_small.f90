!> Host eigensolvers of the Rayleigh-Ritz step at the orders of configs[1] (16, 32, 64) and configs[2] (128): DSYEV / DSYEVD / DSYEVR (all pairs, leading
!> eighth), sequential MKL, microseconds per call.  flang -O2 rr_small.f90 -L/opt/conda/lib -Wl,--no-as-needed -lmkl_intel_lp64 -lmkl_sequential -lmkl_core -Wl,-rpath,/opt/conda/lib
program t
  implicit none
  integer, parameter :: dp = kind(1.d0)
  integer :: n, rep, i, j, info, lwork, liwork, m, ns(4), k, il, iu
  real(dp), allocatable :: a(:,:), a0(:,:), w(:), work(:), z(:,:)
  integer, allocatable :: iwork(:), isuppz(:)
  integer(8) :: c0, c1, rate
  real(dp) :: q(1), vl, vu
  integer :: iq(1)
  ns = [16, 32, 64, 128]
  do k = 1, 4
    n = ns(k)
    allocate(a(n,n), a0(n,n), w(n), z(n,n), isuppz(2*n))
    call random_number(a0); a0 = 1.0e-3_dp*(a0 + transpose(a0))
    do i = 1, n; a0(i,i) = real(i,dp); end do
    ! dsyev
    call dsyev('V','U',n,a,n,w,q,-1,info); lwork = int(q(1)); allocate(work(lwork))
    call system_clock(c0, rate)
    do rep = 1, 200; a = a0; call dsyev('V','U',n,a,n,w,work,lwork,info); end do
    call system_clock(c1); print '(a,i4,a,f8.1,a)', 'n=', n, ' dsyev  ', 1e6*real(c1-c0,dp)/rate/200, ' us'
    deallocate(work)
    call dsyevd('V','U',n,a,n,w,q,-1,iq,-1,info); lwork=int(q(1)); liwork=iq(1); allocate(work(lwork), iwork(liwork))
    call system_clock(c0)
    do rep = 1, 200; a = a0; call dsyevd('V','U',n,a,n,w,work,lwork,iwork,liwork,info); end do
    call system_clock(c1); print '(a,i4,a,f8.1,a)', 'n=', n, ' dsyevd ', 1e6*real(c1-c0,dp)/rate/200, ' us'
    deallocate(work, iwork)
    call dsyevr('V','A','U',n,a,n,vl,vu,1,n,0.0_dp,m,w,z,n,isuppz,q,-1,iq,-1,info); lwork=int(q(1)); liwork=iq(1); allocate(work(lwork), iwork(liwork))
    call system_clock(c0)
    do rep = 1, 200; a = a0; call dsyevr('V','A','U',n,a,n,vl,vu,1,n,0.0_dp,m,w,z,n,isuppz,work,lwork,iwork,liwork,info); end do
    call system_clock(c1); print '(a,i4,a,f8.1,a)', 'n=', n, ' dsyevr all ', 1e6*real(c1-c0,dp)/rate/200, ' us'
    il = 1; iu = max(1, n/8)
    call system_clock(c0)
    do rep = 1, 200; a = a0; call dsyevr('V','I','U',n,a,n,vl,vu,il,iu,0.0_dp,m,w,z,n,isuppz,work,lwork,iwork,liwork,info); end do
    call system_clock(c1); print '(a,i4,a,f8.1,a)', 'n=', n, ' dsyevr n/8 ', 1e6*real(c1-c0,dp)/rate/200, ' us'
    deallocate(work, iwork, a, a0, w, z, isuppz)
  end do
end program
