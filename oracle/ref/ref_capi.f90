! TEST INFRASTRUCTURE -- not part of the product, never linked into libcales_hip.so.
!
! bind(C) entry points around the *reference's own* compiled modules
! (/root/reference/src/*.f90, compiled where they lie by oracle/ref/Makefile
! into oracle/_ref/). Used only by tests/golden/gen_golden.py (to produce the
! committed golden vectors) and by tests that validate the C restatement in
! oracle/cales_oracle.c when oracle/_ref/ is present.
!
! The set-up below follows the order of the reference driver
! (src/main.f90:135-317) for ONE rank with x-aligned pencils: what initmpi
! (src/initmpi.f90:56-73,178-204,208-223) would produce for nproc=1 is written
! out by hand because initmpi itself needs 2decomp-fft, which is not vendored.
! solver.f90 / fft.f90 plans are NOT reachable here (FFTW, 2decomp absent).
module ref_state
  use mpi
  use mod_precision , only: rp
  use mod_typedef   , only: bound
  implicit none
  integer :: n(3),lo(3),hi(3)
  real(rp), allocatable, dimension(:) :: dzc,dzf,zc,zf,dzci,dzfi,gvr_c,gvr_f
  type(bound) :: bcu,bcv,bcw,bcp,bcs,bcuf,bcvf,bcwf,bcu_mag,bcv_mag,bcw_mag,rhsbp
  real(rp), allocatable, dimension(:,:,:) :: rhsbz
  logical :: is_setup = .false.
contains
  subroutine alloc_bound(b,nn,nh)
    type(bound), intent(inout) :: b
    integer, intent(in) :: nn(3),nh
    allocate(b%x(1-nh:nn(2)+nh,1-nh:nn(3)+nh,0:1), &
             b%y(1-nh:nn(1)+nh,1-nh:nn(3)+nh,0:1), &
             b%z(1-nh:nn(1)+nh,1-nh:nn(2)+nh,0:1))
    b%x = 0._rp; b%y = 0._rp; b%z = 0._rp
  end subroutine alloc_bound
end module ref_state
!
subroutine ref_init(istat) bind(C,name='ref_init')
  use, intrinsic :: iso_c_binding
  use mpi
  use ref_state
  use mod_common_mpi, only: myid,ierr,halo,ipencil_axis
  use mod_param
  use mod_initgrid  , only: initgrid
  use mod_bound     , only: initbc,cmpt_rhs_b
  implicit none
  integer(c_int), intent(out) :: istat
  logical :: flag
  integer :: idir,nn(3)
  istat = 0
  call MPI_INITIALIZED(flag,ierr)
  if(.not.flag) call MPI_INIT(ierr)
  call MPI_COMM_RANK(MPI_COMM_WORLD,myid,ierr)
  call read_input(myid)          ! reads ./input.nml (src/param.f90:88)
  !
  ! one rank, x-pencils (what initmpi yields for nproc = 1)
  !
  ipencil_axis = 1
  n(:)  = ng(:); lo(:) = 1; hi(:) = ng(:)
  dims(:) = 1
  is_bound(:,:) = .false.
  nb(:,1) = MPI_PROC_NULL
  do idir = 2,3
    if(cbcpre(0,idir)//cbcpre(1,idir) == 'PP') then
      nb(:,idir) = 0             ! periodic: the rank is its own neighbour
    else
      nb(:,idir) = MPI_PROC_NULL
    end if
  end do
  where(nb(:,:) == MPI_PROC_NULL) is_bound(:,:) = .true.
  nn(:) = n(:) + 2
  call MPI_TYPE_VECTOR(nn(2)*nn(3),1          ,nn(1)            ,MPI_DOUBLE_PRECISION,halo(1),ierr)
  call MPI_TYPE_VECTOR(      nn(3),nn(1)      ,nn(1)*nn(2)      ,MPI_DOUBLE_PRECISION,halo(2),ierr)
  call MPI_TYPE_VECTOR(          1,nn(1)*nn(2),nn(1)*nn(2)*nn(3),MPI_DOUBLE_PRECISION,halo(3),ierr)
  do idir = 1,3
    call MPI_TYPE_COMMIT(halo(idir),ierr)
  end do
  !
  allocate(dzc(0:n(3)+1),dzf(0:n(3)+1),zc(0:n(3)+1),zf(0:n(3)+1),dzci(0:n(3)+1),dzfi(0:n(3)+1))
  allocate(gvr_c(0:n(3)+1),gvr_f(0:n(3)+1))
  call initgrid(gtype,ng(3),gr,l(3),dzc,dzf,zc,zf)
  dzci(:) = dzc(:)**(-1)
  dzfi(:) = dzf(:)**(-1)
  gvr_c(:) = dl(1)*dl(2)*dzc(:)/(l(1)*l(2)*l(3))
  gvr_f(:) = dl(1)*dl(2)*dzf(:)/(l(1)*l(2)*l(3))
  !
  call alloc_bound(bcu,n,1); call alloc_bound(bcv,n,1); call alloc_bound(bcw,n,1)
  call alloc_bound(bcp,n,1); call alloc_bound(bcs,n,1)
  call alloc_bound(bcuf,n,1); call alloc_bound(bcvf,n,1); call alloc_bound(bcwf,n,1)
  call alloc_bound(bcu_mag,n,1); call alloc_bound(bcv_mag,n,1); call alloc_bound(bcw_mag,n,1)
  call alloc_bound(rhsbp,n,0)
  allocate(rhsbz(n(1),n(2),0:1))
  index_wm(:,:) = 0
  call initbc(sgstype,cbcvel,bcvel,bcpre,bcsgs,bcu,bcv,bcw,bcp,bcs,bcu_mag,bcv_mag,bcw_mag, &
              bcuf,bcvf,bcwf,n,is_bound,lwm,l,zc,dl,dzc,hwm,index_wm)
  call cmpt_rhs_b(ng,dl,dzc,dzf,cbcpre,bcp,['c','c','c'],rhsbp%x,rhsbp%y,rhsbp%z)
  is_setup = .true.
end subroutine ref_init
!
subroutine ref_finalize() bind(C,name='ref_finalize')
  use mpi
  implicit none
  integer :: ierr
  logical :: flag
  call MPI_FINALIZED(flag,ierr)
  if(.not.flag) call MPI_FINALIZE(ierr)
end subroutine ref_finalize
!
subroutine ref_get_ints(iv) bind(C,name='ref_get_ints')
  ! ng(3) gtype nstep restart is_overwrite_save nsaves_max icheck iout0d iout1d iout2d iout3d isave
  ! stop_type(3) is_forced(3) is_wallturb dims(2) lwm(6) index_wm(6)
  use, intrinsic :: iso_c_binding
  use mod_param
  implicit none
  integer(c_int), intent(out) :: iv(40)
  iv(:) = 0
  iv(1:3) = ng; iv(4) = gtype; iv(5) = nstep
  iv(6) = merge(1,0,restart); iv(7) = merge(1,0,is_overwrite_save); iv(8) = nsaves_max
  iv(9) = icheck; iv(10) = iout0d; iv(11) = iout1d; iv(12) = iout2d; iv(13) = iout3d; iv(14) = isave
  iv(15:17) = merge(1,0,stop_type); iv(18:20) = merge(1,0,is_forced); iv(21) = merge(1,0,is_wallturb)
  iv(22:23) = dims
  iv(24:29) = reshape(lwm,[6]); iv(30:35) = reshape(index_wm,[6])
end subroutine ref_get_ints
!
subroutine ref_get_reals(rv) bind(C,name='ref_get_reals')
  ! l(3) dl(3) dli(3) gr cfl dtmax dt_f visci visc time_max tw_max bforce(3) velf(3) hwm
  ! bcvel(18) bcpre(6) bcsgs(6)
  use, intrinsic :: iso_c_binding
  use mod_param
  implicit none
  real(c_double), intent(out) :: rv(64)
  rv(:) = 0.
  rv(1:3) = l; rv(4:6) = dl; rv(7:9) = dli; rv(10) = gr; rv(11) = cfl; rv(12) = dtmax; rv(13) = dt_f
  rv(14) = visci; rv(15) = visc; rv(16) = time_max; rv(17) = tw_max
  rv(18:20) = bforce; rv(21:23) = velf; rv(24) = hwm
  rv(25:42) = reshape(bcvel,[18]); rv(43:48) = reshape(bcpre,[6]); rv(49:54) = reshape(bcsgs,[6])
end subroutine ref_get_reals
!
subroutine ref_get_chars(cv) bind(C,name='ref_get_chars')
  ! cbcvel(18, AFTER initbc's wall-model rewrite) cbcpre(6) cbcsgs(6) inivel(100) sgstype(100)
  use, intrinsic :: iso_c_binding
  use mod_param
  implicit none
  character(kind=c_char), intent(out) :: cv(230)
  integer :: i,j,k,m
  m = 0
  do k=1,3; do j=1,3; do i=0,1
    m = m+1; cv(m) = cbcvel(i,j,k)
  end do; end do; end do
  do j=1,3; do i=0,1
    m = m+1; cv(m) = cbcpre(i,j)
  end do; end do
  do j=1,3; do i=0,1
    m = m+1; cv(m) = cbcsgs(i,j)
  end do; end do
  do i=1,100
    m = m+1; cv(m) = inivel(i:i)
  end do
  do i=1,100
    m = m+1; cv(m) = sgstype(i:i)
  end do
end subroutine ref_get_chars
!
subroutine ref_get_grid(dzc_o,dzf_o,zc_o,zf_o) bind(C,name='ref_get_grid')
  use, intrinsic :: iso_c_binding
  use ref_state
  implicit none
  real(c_double), intent(out), dimension(0:n(3)+1) :: dzc_o,dzf_o,zc_o,zf_o
  dzc_o = dzc; dzf_o = dzf; zc_o = zc; zf_o = zf
end subroutine ref_get_grid
!
subroutine ref_get_rhsbp(x,y,z) bind(C,name='ref_get_rhsbp')
  use, intrinsic :: iso_c_binding
  use ref_state
  implicit none
  real(c_double), intent(out) :: x(n(2),n(3),0:1),y(n(1),n(3),0:1),z(n(1),n(2),0:1)
  x = rhsbp%x; y = rhsbp%y; z = rhsbp%z
end subroutine ref_get_rhsbp
!
subroutine ref_get_bcvel(ivel,x,y,z) bind(C,name='ref_get_bcvel')
  ! current bcu/bcv/bcw planes (they carry the wall-model Neumann data after bounduvw)
  use, intrinsic :: iso_c_binding
  use ref_state
  implicit none
  integer(c_int), value :: ivel
  real(c_double), intent(out) :: x(0:n(2)+1,0:n(3)+1,0:1),y(0:n(1)+1,0:n(3)+1,0:1),z(0:n(1)+1,0:n(2)+1,0:1)
  select case(ivel)
  case(1); x = bcu%x; y = bcu%y; z = bcu%z
  case(2); x = bcv%x; y = bcv%y; z = bcv%z
  case(3); x = bcw%x; y = bcw%y; z = bcw%z
  end select
end subroutine ref_get_bcvel
!
subroutine ref_initflow(u,v,w,p) bind(C,name='ref_initflow')
  use, intrinsic :: iso_c_binding
  use ref_state
  use mod_param
  use mod_initflow, only: initflow
  implicit none
  real(c_double), intent(inout), dimension(0:n(1)+1,0:n(2)+1,0:n(3)+1) :: u,v,w,p
  call initflow(inivel,bcvel,ng,lo,l,dl,zc,zf,dzc,dzf,visc,is_forced,velf,bforce,is_wallturb,u,v,w,p)
end subroutine ref_initflow
!
subroutine ref_bounduvw(is_updt_wm,is_correc,u,v,w) bind(C,name='ref_bounduvw')
  use, intrinsic :: iso_c_binding
  use ref_state
  use mod_param
  use mod_bound, only: bounduvw
  implicit none
  integer(c_int), value :: is_updt_wm,is_correc
  real(c_double), intent(inout), dimension(0:n(1)+1,0:n(2)+1,0:n(3)+1) :: u,v,w
  call bounduvw(cbcvel,n,bcu,bcv,bcw,bcu_mag,bcv_mag,bcw_mag,nb,is_bound,lwm,l,dl,zc,zf,dzc,dzf, &
                visc,hwm,index_wm,is_updt_wm /= 0,is_correc /= 0,u,v,w)
end subroutine ref_bounduvw
!
subroutine ref_boundp(which,p) bind(C,name='ref_boundp')
  ! which = 0: pressure BCs (cbcpre,bcp); 1: sgs BCs (cbcsgs,bcs)
  use, intrinsic :: iso_c_binding
  use ref_state
  use mod_param
  use mod_bound, only: boundp
  implicit none
  integer(c_int), value :: which
  real(c_double), intent(inout), dimension(0:n(1)+1,0:n(2)+1,0:n(3)+1) :: p
  if(which == 0) then
    call boundp(cbcpre,n,bcp,nb,is_bound,dl,dzc,p)
  else
    call boundp(cbcsgs,n,bcs,nb,is_bound,dl,dzc,p)
  end if
end subroutine ref_boundp
!
subroutine ref_mom(u,v,w,visct,dudt,dvdt,dwdt,dudtd,dvdtd,dwdtd) bind(C,name='ref_mom')
  use, intrinsic :: iso_c_binding
  use ref_state
  use mod_param
  use mod_mom, only: mom_xyz_ad
  implicit none
  real(c_double), intent(in   ), dimension(0:n(1)+1,0:n(2)+1,0:n(3)+1) :: u,v,w,visct
  real(c_double), intent(inout), dimension(n(1),n(2),n(3)) :: dudt,dvdt,dwdt,dudtd,dvdtd,dwdtd
#if defined(_IMPDIFF)
  call mom_xyz_ad(n(1),n(2),n(3),dli(1),dli(2),dzci,dzfi,visc,u,v,w,visct,dudt,dvdt,dwdt,dudtd,dvdtd,dwdtd)
#else
  call mom_xyz_ad(n(1),n(2),n(3),dli(1),dli(2),dzci,dzfi,visc,u,v,w,visct,dudt,dvdt,dwdt)
#endif
end subroutine ref_mom
!
subroutine ref_rk(irk,dt,p,visct,u,v,w,f) bind(C,name='ref_rk')
  use, intrinsic :: iso_c_binding
  use ref_state
  use mod_param
  use mod_rk, only: rk
  implicit none
  integer(c_int), value :: irk
  real(c_double), value :: dt
  real(c_double), intent(in   ), dimension(0:n(1)+1,0:n(2)+1,0:n(3)+1) :: p,visct
  real(c_double), intent(inout), dimension(0:n(1)+1,0:n(2)+1,0:n(3)+1) :: u,v,w
  real(c_double), intent(out) :: f(3)
  call rk(rkcoeff(:,irk),n,dli,dzci,dzfi,gvr_c,gvr_f,visc,dt,p,is_forced,velf,bforce,visct,u,v,w,f)
end subroutine ref_rk
!
subroutine ref_bulk_forcing(f,u,v,w) bind(C,name='ref_bulk_forcing')
  use, intrinsic :: iso_c_binding
  use ref_state
  use mod_param
  use mod_mom, only: bulk_forcing
  implicit none
  real(c_double), intent(in) :: f(3)
  real(c_double), intent(inout), dimension(0:n(1)+1,0:n(2)+1,0:n(3)+1) :: u,v,w
  call bulk_forcing(n,is_forced,f,u,v,w)
end subroutine ref_bulk_forcing
!
subroutine ref_bulk_mean(c_or_f,p,mean) bind(C,name='ref_bulk_mean')
  use, intrinsic :: iso_c_binding
  use ref_state
  use mod_utils, only: bulk_mean
  implicit none
  integer(c_int), value :: c_or_f  ! 0: grid_vol_ratio_c, 1: grid_vol_ratio_f
  real(c_double), intent(in), dimension(0:n(1)+1,0:n(2)+1,0:n(3)+1) :: p
  real(c_double), intent(out) :: mean
  if(c_or_f == 0) then
    call bulk_mean(n,gvr_c,p,mean)
  else
    call bulk_mean(n,gvr_f,p,mean)
  end if
end subroutine ref_bulk_mean
!
subroutine ref_fillps(dtrki,u,v,w,pp) bind(C,name='ref_fillps')
  use, intrinsic :: iso_c_binding
  use ref_state
  use mod_param
  use mod_fillps, only: fillps
  implicit none
  real(c_double), value :: dtrki
  real(c_double), intent(in   ), dimension(0:n(1)+1,0:n(2)+1,0:n(3)+1) :: u,v,w
  real(c_double), intent(inout), dimension(0:n(1)+1,0:n(2)+1,0:n(3)+1) :: pp
  call fillps(n,dli,dzfi,dtrki,u,v,w,pp)
end subroutine ref_fillps
!
subroutine ref_updt_rhs_b_p(pp) bind(C,name='ref_updt_rhs_b_p')
  use, intrinsic :: iso_c_binding
  use ref_state
  use mod_param
  use mod_bound, only: updt_rhs_b
  implicit none
  real(c_double), intent(inout), dimension(0:n(1)+1,0:n(2)+1,0:n(3)+1) :: pp
  call updt_rhs_b(['c','c','c'],cbcpre,n,is_bound,rhsbp%x,rhsbp%y,rhsbp%z,pp)
end subroutine ref_updt_rhs_b_p
!
subroutine ref_updt_rhs_b_velz(ivel,alpha,q) bind(C,name='ref_updt_rhs_b_velz')
  ! z-implicit diffusion (IMPDIFF_1D): boundary term of the Helmholtz r.h.s.,
  ! as driven from src/main.f90:425-433 / 447-455 / 469-477
  use, intrinsic :: iso_c_binding
  use ref_state
  use mod_param
  use mod_bound, only: cmpt_rhs_b,updt_rhs_b
  implicit none
  integer(c_int), value :: ivel
  real(c_double), value :: alpha
  real(c_double), intent(inout), dimension(0:n(1)+1,0:n(2)+1,0:n(3)+1) :: q
  character(len=1) :: cf(3)
  cf(:) = 'c'; cf(ivel) = 'f'
  select case(ivel)
  case(1); call cmpt_rhs_b(ng,dl,dzc,dzf,cbcvel(:,:,1),bcu,cf,rhsbz=rhsbz)
  case(2); call cmpt_rhs_b(ng,dl,dzc,dzf,cbcvel(:,:,2),bcv,cf,rhsbz=rhsbz)
  case(3); call cmpt_rhs_b(ng,dl,dzc,dzf,cbcvel(:,:,3),bcw,cf,rhsbz=rhsbz)
  end select
  rhsbz(:,:,0:1) = rhsbz(:,:,0:1)*alpha
  call updt_rhs_b(cf,cbcvel(:,:,ivel),n,is_bound,rhsbz=rhsbz,p=q)
end subroutine ref_updt_rhs_b_velz
!
subroutine ref_correc(dtrk,pp,u,v,w) bind(C,name='ref_correc')
  use, intrinsic :: iso_c_binding
  use ref_state
  use mod_param
  use mod_correc, only: correc
  implicit none
  real(c_double), value :: dtrk
  real(c_double), intent(in   ), dimension(0:n(1)+1,0:n(2)+1,0:n(3)+1) :: pp
  real(c_double), intent(inout), dimension(0:n(1)+1,0:n(2)+1,0:n(3)+1) :: u,v,w
  call correc(n,dli,dzci,dtrk,pp,u,v,w)
end subroutine ref_correc
!
subroutine ref_updatep(alpha,pp,p) bind(C,name='ref_updatep')
  use, intrinsic :: iso_c_binding
  use ref_state
  use mod_param
  use mod_updatep, only: updatep
  implicit none
  real(c_double), value :: alpha
  real(c_double), intent(in   ), dimension(0:n(1)+1,0:n(2)+1,0:n(3)+1) :: pp
  real(c_double), intent(inout), dimension(0:n(1)+1,0:n(2)+1,0:n(3)+1) :: p
  call updatep(n,dli,dzci,dzfi,alpha,pp,p)
end subroutine ref_updatep
!
subroutine ref_cmpt_sgs(u,v,w,visct) bind(C,name='ref_cmpt_sgs')
  use, intrinsic :: iso_c_binding
  use ref_state
  use mod_param
  use mod_sgs, only: cmpt_sgs
  implicit none
  real(c_double), intent(in   ), dimension(0:n(1)+1,0:n(2)+1,0:n(3)+1) :: u,v,w
  real(c_double), intent(inout), dimension(0:n(1)+1,0:n(2)+1,0:n(3)+1) :: visct
  call cmpt_sgs(sgstype,n,ng,lo,hi,cbcvel,cbcsgs,bcs,nb,is_bound,lwm,l,dl,dli,zc,zf,dzc,dzf, &
                dzci,dzfi,visc,hwm,index_wm,u,v,w,bcuf,bcvf,bcwf,bcu_mag,bcv_mag,bcw_mag,visct)
end subroutine ref_cmpt_sgs
!
subroutine ref_chkdt(visct,u,v,w,dtmax_o) bind(C,name='ref_chkdt')
  use, intrinsic :: iso_c_binding
  use ref_state
  use mod_param
  use mod_chkdt, only: chkdt
  implicit none
  real(c_double), intent(in), dimension(0:n(1)+1,0:n(2)+1,0:n(3)+1) :: visct,u,v,w
  real(c_double), intent(out) :: dtmax_o
  call chkdt(n,dl,dzci,dzfi,visc,visct,u,v,w,dtmax_o)
end subroutine ref_chkdt
!
subroutine ref_chkdiv(u,v,w,divtot,divmax) bind(C,name='ref_chkdiv')
  use, intrinsic :: iso_c_binding
  use ref_state
  use mod_param
  use mod_chkdiv, only: chkdiv
  implicit none
  real(c_double), intent(in), dimension(0:n(1)+1,0:n(2)+1,0:n(3)+1) :: u,v,w
  real(c_double), intent(out) :: divtot,divmax
  call chkdiv(lo,hi,dli,dzfi,u,v,w,divtot,divmax)
end subroutine ref_chkdiv
!
! out1d_single_point_chan (src/output.f90:509-1061, idir = 3): writes <fname>.out/.bin (27 plane statistics),
! <fname>_reystr_budget.out/.bin (38) and <fname>_leakage.out/.bin (6) into the current directory
subroutine ref_out1d_single_point_chan(fname_c,nchar,u,v,w,p,visct) bind(C,name='ref_out1d_single_point_chan')
  use, intrinsic :: iso_c_binding
  use ref_state
  use mod_param, only: ng,l,dl
  use mod_output_stats, only: out1d_single_point_chan
  implicit none
  integer(c_int), intent(in), value :: nchar
  character(kind=c_char), intent(in) :: fname_c(nchar)
  real(c_double), intent(in), dimension(0:n(1)+1,0:n(2)+1,0:n(3)+1) :: u,v,w,p,visct
  character(len=:), allocatable :: fname
  integer :: q
  allocate(character(len=nchar) :: fname)
  do q = 1,nchar
    fname(q:q) = fname_c(q)
  end do
  call out1d_single_point_chan(fname,ng,lo,hi,3,l,dl,dzc,dzf,zc,zf,u,v,w,p,visct)
end subroutine ref_out1d_single_point_chan
!
! out1d (src/output.f90:50-163), out1d_chan (317-405, idir = 3), out2d_duct (406-507, streamwise x): each writes ONE text file of 8 significant digits
subroutine ref_out1d(fname_c,nchar,idir,use_dzc,p) bind(C,name='ref_out1d')
  use, intrinsic :: iso_c_binding
  use ref_state
  use mod_param, only: ng,l,dl
  use mod_output_stats, only: out1d
  implicit none
  integer(c_int), intent(in), value :: nchar,idir,use_dzc
  character(kind=c_char), intent(in) :: fname_c(nchar)
  real(c_double), intent(in), dimension(0:n(1)+1,0:n(2)+1,0:n(3)+1) :: p
  character(len=:), allocatable :: fname
  integer :: q
  allocate(character(len=nchar) :: fname)
  do q = 1,nchar
    fname(q:q) = fname_c(q)
  end do
  if(use_dzc /= 0) then
    call out1d(fname,ng,lo,hi,idir,l,dl,zf,dzc,p)
  else
    call out1d(fname,ng,lo,hi,idir,l,dl,zc,dzf,p)
  end if
end subroutine ref_out1d
subroutine ref_out_uvw(which,fname_c,nchar,u,v,w) bind(C,name='ref_out_uvw')      ! which = 0: out1d_chan, 1: out2d_duct
  use, intrinsic :: iso_c_binding
  use ref_state
  use mod_param, only: ng,l,dl
  use mod_output_stats, only: out1d_chan,out2d_duct
  implicit none
  integer(c_int), intent(in), value :: which,nchar
  character(kind=c_char), intent(in) :: fname_c(nchar)
  real(c_double), intent(in), dimension(0:n(1)+1,0:n(2)+1,0:n(3)+1) :: u,v,w
  character(len=:), allocatable :: fname
  integer :: q
  allocate(character(len=nchar) :: fname)
  do q = 1,nchar
    fname(q:q) = fname_c(q)
  end do
  if(which == 0) then
    call out1d_chan(fname,ng,lo,hi,3,l,dl,zc,u,v,w)
  else
    call out2d_duct(fname,ng,lo,hi,1,l,dl,zc,u,v,w)
  end if
end subroutine ref_out_uvw
!
! the plain-arithmetic routines of initsolver.f90 / solver.f90 (compiled from their own lines, see the Makefile)
subroutine ref_eigenvalues(n,cbc2,c_or_f,lambda) bind(C,name='ref_eigenvalues')
  use, intrinsic :: iso_c_binding
  use mod_initsolver_pure, only: eigenvalues
  implicit none
  integer(c_int), intent(in), value :: n
  character(kind=c_char), intent(in) :: cbc2(0:1),c_or_f
  real(c_double), intent(out) :: lambda(n)
  character(len=1) :: cbc(0:1)
  cbc(0) = cbc2(0); cbc(1) = cbc2(1)
  call eigenvalues(n,cbc,c_or_f,lambda)
end subroutine ref_eigenvalues
subroutine ref_tridmatrix(cbc2,n,dzi,dzci,dzfi,c_or_f,a,b,c) bind(C,name='ref_tridmatrix')
  use, intrinsic :: iso_c_binding
  use mod_initsolver_pure, only: tridmatrix
  implicit none
  integer(c_int), intent(in), value :: n
  real(c_double), intent(in), value :: dzi
  character(kind=c_char), intent(in) :: cbc2(0:1),c_or_f
  real(c_double), intent(in) :: dzci(0:n+1),dzfi(0:n+1)
  real(c_double), intent(out) :: a(n),b(n),c(n)
  character(len=1) :: cbc(0:1)
  cbc(0) = cbc2(0); cbc(1) = cbc2(1)
  call tridmatrix(cbc,n,dzi,dzci,dzfi,c_or_f,a,b,c)
end subroutine ref_tridmatrix
! gaussel / gaussel_periodic on p(1-nh:nx+nh,1-nh:ny+nh,1-nh:nz+nh), n unknowns per column (n = nz - q), with or without lambdaxy
subroutine ref_gaussel(periodic,nx,ny,nz,n,nh,a,b,c,p,has_lam,lambdaxy) bind(C,name='ref_gaussel')
  use, intrinsic :: iso_c_binding
  use mod_solver_pure, only: gaussel,gaussel_periodic
  implicit none
  integer(c_int), intent(in), value :: periodic,nx,ny,nz,n,nh,has_lam
  real(c_double), intent(in) :: a(nz),b(nz),c(nz)
  real(c_double), intent(inout) :: p(1-nh:nx+nh,1-nh:ny+nh,1-nh:nz+nh)
  real(c_double), intent(in) :: lambdaxy(nx,ny)
  if(periodic /= 0) then
    if(has_lam /= 0) then
      call gaussel_periodic(nx,ny,n,nh,a,b,c,p,lambdaxy)
    else
      call gaussel_periodic(nx,ny,n,nh,a,b,c,p)
    end if
  else
    if(has_lam /= 0) then
      call gaussel(nx,ny,n,nh,a,b,c,p,lambdaxy)
    else
      call gaussel(nx,ny,n,nh,a,b,c,p)
    end if
  end if
end subroutine ref_gaussel
