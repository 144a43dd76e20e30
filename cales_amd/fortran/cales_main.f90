! cales -- Fortran host of the MI355X hot path. Keeps the CaNS/CaLES case interface (input.nml, fld.bin,
! time.out, forcing.out, grid.bin/grid.out/geometry.out) and the control flow of the reference driver
! (reference src/main.f90:135-632), and hands every per-step operator to libcales_hip.so through
! ISO_C_BINDING (module cales_c). One rank / one GPU; the multi-GPU host is cales_amd/decomp.py.
!
! usage: cales [impdiff]      (impdiff = 0 explicit [default], 1 implicit in x,y,z for periodic x,y, 2 z-implicit: the reference's build switches
!                              _IMPDIFF/_IMPDIFF_1D are run-time here); reads ./input.nml
program cales
  use, intrinsic :: iso_c_binding
  use, intrinsic :: iso_fortran_env, only: int64
  use, intrinsic :: ieee_arithmetic, only: is_nan => ieee_is_nan
  use cales_c
  implicit none
  integer, parameter :: rp = c_double
  real(rp), parameter :: small = epsilon(1._rp)*10**(precision(1._rp)/2)   ! src/param.f90:24
  ! ---- namelist variables, declared as in src/param.f90:37-76
  integer  :: ng(3),gtype,nstep,nsaves_max,icheck,iout0d,iout1d,iout2d,iout3d,isave,dims(2),lwm(0:1,3)
  real(rp) :: l(3),gr,cfl,dtmax,dt_f,visci,time_max,tw_max,bforce(3),velf(3),hwm
  real(rp) :: bcvel(0:1,3,3),bcpre(0:1,3),bcsgs(0:1,3)
  character(len=100) :: inivel,sgstype
  character(len=1) :: cbcvel(0:1,3,3),cbcpre(0:1,3),cbcsgs(0:1,3)
  logical :: is_wallturb,stop_type(3),restart,is_overwrite_save,is_forced(3)
  namelist /dns/ ng,l,gtype,gr,cfl,dtmax,dt_f,visci,inivel,is_wallturb,nstep,time_max,tw_max,stop_type, &
                 restart,is_overwrite_save,nsaves_max,icheck,iout0d,iout1d,iout2d,iout3d,isave, &
                 cbcvel,cbcpre,cbcsgs,bcvel,bcpre,bcsgs,bforce,is_forced,velf,dims
  namelist /les/ sgstype,lwm,hwm
  ! ----
  type(cales_case) :: cs
  type(c_ptr) :: ctx
  real(rp), allocatable, dimension(:,:,:) :: u,v,w,p,visct
  real(rp), allocatable, dimension(:) :: dzc,dzf,zc,zf
  real(rp) :: dt,dti,dt_cfl,time,divtot,divmax,dpdl(3),meanvel(3),var(7),tw,dt12
  integer(int64) :: c0,c1,crate,cstep0
  integer :: istep,iunit,ierr,i,j,k,m,impdiff,savecounter,rc
  logical :: is_done,kill
  character(len=512) :: iomsg,arg
  character(len=100) :: filename
  character(len=7) :: fldnum
  character(len=4) :: chkptnum
  character(kind=c_char) :: cmsg(512)
  !
  impdiff = 0
  if(command_argument_count() >= 1) then
    call get_command_argument(1,arg); read(arg,*) impdiff
  end if
  !
  ! read parameter file (src/param.f90:88-157)
  !
  dt_f = -1.; sgstype = ''; lwm = 0; hwm = 0.
  open(newunit=iunit,file='input.nml',status='old',action='read',iostat=ierr,iomsg=iomsg)
  if(ierr /= 0) then
    print*, 'Error reading the input file: ', trim(iomsg); print*, 'Aborting...'; error stop
  end if
  read(iunit,nml=dns,iostat=ierr,iomsg=iomsg)
  if(ierr /= 0) then
    print*, 'Error reading dns namelist: ', trim(iomsg); print*, 'Aborting...'; error stop
  end if
  read(iunit,nml=les,iostat=ierr,iomsg=iomsg)
  if(ierr /= 0) then
    ! examples/dns/* of the reference close &les with '\': this run-time reports it after reading the values
    if(len_trim(sgstype) == 0) then
      print*, 'Error reading les namelist: ', trim(iomsg); print*, 'Aborting...'; error stop
    end if
  end if
  close(iunit)
  !
  cs%ng = ng; cs%l = l; cs%gtype = gtype; cs%gr = gr; cs%visci = visci
  cs%cbcvel = reshape(cbcvel,[18]); cs%cbcpre = reshape(cbcpre,[6]); cs%cbcsgs = reshape(cbcsgs,[6])
  cs%bcvel = reshape(bcvel,[18]); cs%bcpre = reshape(bcpre,[6]); cs%bcsgs = reshape(bcsgs,[6])
  cs%bforce = bforce; cs%is_forced = merge(1,0,is_forced); cs%velf = velf
  select case(trim(sgstype))
  case('none');  cs%sgstype = 0
  case('smag');  cs%sgstype = 1
  case('dsmag'); cs%sgstype = 2
  case('amd');   print*, 'ERROR: AMD model not yet implemented'; error stop   ! src/sgs.f90:381-382
  case default;  print*, 'ERROR: unknown SGS model'; error stop               ! src/sgs.f90:383-384
  end select
  cs%lwm = reshape(lwm,[6]); cs%hwm = hwm; cs%impdiff = impdiff; cs%nranks = 1; cs%rank = 0
  !
  ! a-priori input checks (src/sanity.f90:33-67)
  !
  if(.not.any(stop_type(:))) then
    print*, 'ERROR: stopping criterion not chosen.'; call abortit
  end if
  if(cales_check_case(cs,cmsg,512) /= 0) then
    print*, 'ERROR: ', cstr(cmsg); call abortit
  end if
  !
  print*, '*******************************'
  print*, '*** Beginning of simulation ***'
  print*, '*******************************'
  allocate(dzc(0:ng(3)+1),dzf(0:ng(3)+1),zc(0:ng(3)+1),zf(0:ng(3)+1))
  rc = cales_initgrid(gtype,ng(3),gr,l(3),dzc,dzf,zc,zf)
  open(newunit=iunit,file='grid.bin',action='write',form='unformatted',access='stream',status='replace')   ! main.f90:248-250
  write(iunit) dzc(1:ng(3)),dzf(1:ng(3)),zc(1:ng(3)),zf(1:ng(3)); close(iunit)
  open(newunit=iunit,file='grid.out')
  do k=0,ng(3)+1
    write(iunit,'(*(E16.7e3))') 0.,zf(k),zc(k),dzf(k),dzc(k)
  end do
  close(iunit)
  open(newunit=iunit,file='geometry.out'); write(iunit,*) ng(1),ng(2),ng(3); write(iunit,*) l(1),l(2),l(3); close(iunit)
  !
  allocate(u(0:ng(1)+1,0:ng(2)+1,0:ng(3)+1),v(0:ng(1)+1,0:ng(2)+1,0:ng(3)+1),w(0:ng(1)+1,0:ng(2)+1,0:ng(3)+1), &
           p(0:ng(1)+1,0:ng(2)+1,0:ng(3)+1),visct(0:ng(1)+1,0:ng(2)+1,0:ng(3)+1))
  u = 0.; v = 0.; w = 0.; p = 0.; visct = 0.
  if(cales_create(cs,c_null_ptr,ctx) /= 0) then
    print*, 'ERROR: ', cstr_ptr(cales_last_error(c_null_ptr)); error stop
  end if
  if(.not.restart) then
    istep = 0; time = 0.
    rc = cales_initflow(cs,trim(inivel)//c_null_char,merge(1,0,is_wallturb),u,v,w,p)
    if(rc /= 0) then
      print*, 'ERROR: invalid name for initial velocity field'     ! src/initflow.f90:203-209
      print*, '*** Simulation aborted due to errors in the case file ***'; error stop
    end if
    print*, '*** Initial condition succesfully set ***'
  else
    call load_all('r','fld.bin')
    print*, '*** Checkpoint loaded at time = ', time, 'time step = ', istep, '. ***'
  end if
  call chk(cales_upload_state(ctx,u,v,w,p))
  call chk(cales_bounduvw(ctx,1,0)); call chk(cales_boundp(ctx,CALES_P,0))          ! main.f90:370-375
  call chk(cales_cmpt_sgs(ctx));     call chk(cales_boundp(ctx,CALES_VISCT,1))
  call chk(cales_chkdt(ctx,dt_cfl))
  dt = merge(dt_f,min(cfl*dt_cfl,dtmax),dt_f > 0.)
  print*, 'dt_cfl = ', dt_cfl, 'dt = ', dt
  dti = 1./dt
  kill = .false.; savecounter = 0
  call system_clock(c0,crate); cstep0 = c0
  !
  ! main loop (src/main.f90:403-619)
  !
  print*, '*** Calculation loop starts now ***'
  is_done = .false.
  do while(.not.is_done)
    call system_clock(cstep0)
    istep = istep + 1
    time = time + dt
    print*, 'Time step #', istep, 'Time = ', time
    call chk(cales_step(ctx,dt))            ! 3 RK substeps, main.f90:417-508
    if(stop_type(1)) then
      if(istep >= nstep   ) is_done = is_done.or..true.
    end if
    if(stop_type(2)) then
      if(time  >= time_max) is_done = is_done.or..true.
    end if
    if(stop_type(3)) then
      call system_clock(c1); tw = real(c1-c0,rp)/real(crate,rp)/3600.
      if(tw    >= tw_max  ) is_done = is_done.or..true.
    end if
    if(icheck > 0.and.mod(istep,max(icheck,1)) == 0) then
      print*, 'Checking stability and divergence...'
      call chk(cales_chkdt(ctx,dt_cfl))
      dt = merge(dt_f,min(cfl*dt_cfl,dtmax),dt_f > 0.)
      print*, 'dt_cfl = ', dt_cfl, 'dt = ', dt
      if(dt_cfl < small) then
        print*, 'ERROR: time step is too small.'; print*, 'Aborting...'
        is_done = .true.; kill = .true.
      end if
      dti = 1./dt
      call chk(cales_chkdiv(ctx,divtot,divmax))
      print*, 'Total divergence = ', divtot, '| Maximum divergence = ', divmax
      if(divmax > small.or.is_nan(divtot)) then
        print*, 'ERROR: maximum divergence is too large.'; print*, 'Aborting...'
        is_done = .true.; kill = .true.
      end if
    end if
    if(iout0d > 0.and.mod(istep,max(iout0d,1)) == 0) then     ! main.f90:548-573
      var(1) = 1.*istep; var(2) = dt; var(3) = time
      call out0d('time.out',3,var)
      if(any(is_forced(:)).or.any(abs(bforce(:)) > 0.)) then
        meanvel(:) = 0.
        do m=1,3
          if(is_forced(m).or.abs(bforce(m)) > 0.) call chk(cales_bulk_mean(ctx,m-1,merge(0,1,m==3),meanvel(m)))
        end do
        call chk(cales_get_dpdl(ctx,dpdl))
        if(.not.any(is_forced(:))) dpdl(:) = -bforce(:)
        var(1) = time; var(2:4) = dpdl(1:3); var(5:7) = meanvel(1:3)
        call out0d('forcing.out',7,var)
      end if
    end if
    write(fldnum,'(i7.7)') istep
    if(iout1d > 0.and.mod(istep,max(iout1d,1)) == 0) call out1d_chan_stats('velstats_fld_'//fldnum)     ! main.f90:575-579, out1d.h90
    if((iout2d > 0.and.mod(istep,max(iout2d,1)) == 0).or.(iout3d > 0.and.mod(istep,max(iout3d,1)) == 0)) then     ! main.f90:580-589
      call chk(cales_download_state(ctx,u,v,w,p,visct))
      if(iout2d > 0.and.mod(istep,max(iout2d,1)) == 0) then     ! out2d.h90: the plane j = ng(2)/2 of the five fields
        call visu_2d('vex_slice_fld_'//fldnum//'.bin','Velocity_X',u); call visu_2d('vey_slice_fld_'//fldnum//'.bin','Velocity_Y',v)
        call visu_2d('vez_slice_fld_'//fldnum//'.bin','Velocity_Z',w); call visu_2d('pre_slice_fld_'//fldnum//'.bin','Pressure_P',p)
        call visu_2d('visct_slice_fld_'//fldnum//'.bin','Viscosity',visct)
      end if
      if(iout3d > 0.and.mod(istep,max(iout3d,1)) == 0) then     ! out3d.h90: the five fields without halos
        call visu_3d('vex_fld_'//fldnum//'.bin','Velocity_X',u); call visu_3d('vey_fld_'//fldnum//'.bin','Velocity_Y',v)
        call visu_3d('vez_fld_'//fldnum//'.bin','Velocity_Z',w); call visu_3d('pre_fld_'//fldnum//'.bin','Pressure',p)
        call visu_3d('visct_fld_'//fldnum//'.bin','Viscosity',visct)
      end if
    end if
    if((isave > 0.and.mod(istep,max(isave,1)) == 0).or.(is_done.and..not.kill)) then     ! main.f90:590-611
      if(is_overwrite_save) then
        filename = 'fld.bin'
      else
        filename = 'fld_'//fldnum//'.bin'
        if(nsaves_max > 0) then
          if(savecounter >= nsaves_max) savecounter = 0
          savecounter = savecounter + 1
          write(chkptnum,'(i4.4)') savecounter
          filename = 'fld_'//chkptnum//'.bin'
          var(1) = 1.*istep; var(2) = time; var(3) = 1.*savecounter
          call out0d('log_checkpoints.out',3,var)
        end if
        call execute_command_line('ln -sf '//trim(filename)//' fld.bin')
      end if
      call chk(cales_download_state(ctx,u,v,w,p,visct))
      call load_all('w',trim(filename))
      print*, '*** Checkpoint saved at time = ', time, 'time step = ', istep, '. ***'
    end if
    call chk(cales_sync(ctx))
    call system_clock(c1); dt12 = real(c1-cstep0,rp)/real(crate,rp)
    print*, 'Avrg, min & max elapsed time: '
    print*, dt12,dt12,dt12
  end do
  call cales_destroy(ctx)
  if(.not.kill) print*, '*** Fim ***'
contains
  subroutine chk(ist)
    integer(c_int), intent(in) :: ist
    if(ist /= 0) then
      print*, 'ERROR (libcales_hip): ', cstr_ptr(cales_last_error(ctx)); error stop
    end if
  end subroutine chk
  subroutine abortit
    print*, ''
    print*, '*** Simulation aborted due to errors in the input file ***'
    print*, '    check `input.nml`.'
    error stop
  end subroutine abortit
  subroutine visu_log(flog,fbin,varname,nmin,nmax)    ! write_log_output, src/output.f90:244-272
    character(len=*), intent(in) :: flog,fbin,varname
    integer, intent(in) :: nmin(3),nmax(3)
    integer :: iu
    open(newunit=iu,file=flog,position='append')
    write(iu,'(A30,A15,9I5,E16.7E3,I7)') fbin,varname,nmin,nmax,[1,1,1],time,istep
    close(iu)
  end subroutine visu_log
  subroutine visu_2d(fbin,varname,q)    ! write_visu_2d with inorm = 2, islice = ng(2)/2, src/output.f90:289-315
    character(len=*), intent(in) :: fbin,varname
    real(rp), intent(in) :: q(0:,0:,0:)
    integer :: iu,js
    js = ng(2)/2
    open(newunit=iu,file=fbin,access='stream',status='replace'); write(iu) q(1:ng(1),js,1:ng(3)); close(iu)
    call visu_log('log_visu_2d_slice_1.out',fbin,varname,[1,js,1],[ng(1),js,ng(3)])
  end subroutine visu_2d
  subroutine visu_3d(fbin,varname,q)    ! write_visu_3d with nskip = 1, src/output.f90:274-287
    character(len=*), intent(in) :: fbin,varname
    real(rp), intent(in) :: q(0:,0:,0:)
    integer :: iu
    open(newunit=iu,file=fbin,access='stream',status='replace'); write(iu) q(1:ng(1),1:ng(2),1:ng(3)); close(iu)
    call visu_log('log_visu_3d.out',fbin,varname,[1,1,1],ng)
  end subroutine visu_3d
  subroutine out1d_chan_stats(fname)    ! the velstats_fld_*.out/.bin pair of out1d_single_point_chan, src/output.f90:683-699
    character(len=*), intent(in) :: fname
    real(rp), allocatable :: buf(:,:),leak(:,:)
    integer :: iu,kk,q
    allocate(buf(27,ng(3)))
    call chk(cales_out1d_single_point_chan(ctx,buf))
    open(newunit=iu,file=fname//'.out')
    do kk=1,ng(3)
      write(iu,'(*(es24.16e3,1x))') zc(kk),zf(kk),(buf(q,kk),q=1,27),dzc(kk),dzf(kk)
    end do
    close(iu)
    open(newunit=iu,file=fname//'.bin',access='stream'); write(iu) buf; close(iu)
    deallocate(buf)
    allocate(buf(38,ng(3)),leak(6,ng(3)))      ! budgets and leakage, src/output.f90:990-1055
    call chk(cales_out1d_chan_budgets(ctx,buf,leak))
    open(newunit=iu,file=fname//'_reystr_budget.out')
    do kk=1,ng(3)
      write(iu,'(*(es24.16e3,1x))') zc(kk),zf(kk),(buf(q,kk),q=1,38),dzc(kk),dzf(kk)
    end do
    close(iu)
    open(newunit=iu,file=fname//'_reystr_budget.bin',access='stream'); write(iu) buf; close(iu)
    open(newunit=iu,file=fname//'_leakage.out')
    do kk=1,ng(3)
      write(iu,'(*(es24.16e3,1x))') zc(kk),zf(kk),(leak(q,kk),q=1,6),dzc(kk),dzf(kk)
    end do
    close(iu)
    open(newunit=iu,file=fname//'_leakage.bin',access='stream'); write(iu) leak; close(iu)
  end subroutine out1d_chan_stats
  subroutine out0d(fname,n,vv)    ! src/output.f90:18-37
    character(len=*), intent(in) :: fname
    integer, intent(in) :: n
    real(rp), intent(in) :: vv(:)
    integer :: iu
    open(newunit=iu,file=fname,position='append')
    write(iu,'(*(E16.7e3))') vv(1:n)
    close(iu)
  end subroutine out0d
  subroutine load_all(io,fname)   ! byte layout of src/load.f90:20-153: u,v,w,p (no halos) then [time, real(istep)]
    character(len=1), intent(in) :: io
    character(len=*), intent(in) :: fname
    integer :: iu
    integer(int64) :: fsize,good
    real(rp) :: fldinfo(2)
    select case(io)
    case('r')
      inquire(file=fname,size=fsize)
      good = (int(ng(1),int64)*ng(2)*ng(3)*4+2)*8
      if(fsize /= good) then
        print*, '*** Simulation aborted due a checkpoint file with incorrect size ***'
        print*, '    file: ', fname, ' | expected size: ', good, '| actual size: ', fsize
        error stop
      end if
      open(newunit=iu,file=fname,action='read',form='unformatted',access='stream',status='old')
      read(iu) u(1:ng(1),1:ng(2),1:ng(3)),v(1:ng(1),1:ng(2),1:ng(3)),w(1:ng(1),1:ng(2),1:ng(3)),p(1:ng(1),1:ng(2),1:ng(3)),fldinfo
      close(iu)
      time = fldinfo(1); istep = nint(fldinfo(2))
    case('w')
      open(newunit=iu,file=fname,action='write',form='unformatted',access='stream',status='replace')
      fldinfo = [time,1._rp*istep]
      write(iu) u(1:ng(1),1:ng(2),1:ng(3)),v(1:ng(1),1:ng(2),1:ng(3)),w(1:ng(1),1:ng(2),1:ng(3)),p(1:ng(1),1:ng(2),1:ng(3)),fldinfo
      close(iu)
    end select
  end subroutine load_all
  function cstr(c) result(s)
    character(kind=c_char), intent(in) :: c(:)
    character(len=:), allocatable :: s
    integer :: q
    s = ''
    do q=1,size(c)
      if(c(q) == c_null_char) exit
      s = s//c(q)
    end do
  end function cstr
  function cstr_ptr(pc) result(s)
    type(c_ptr), intent(in) :: pc
    character(len=:), allocatable :: s
    character(kind=c_char), pointer :: f(:)
    if(.not.c_associated(pc)) then
      s = ''; return
    end if
    call c_f_pointer(pc,f,[512])
    s = cstr(f)
  end function cstr_ptr
end program cales
