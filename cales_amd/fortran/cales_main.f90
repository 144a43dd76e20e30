! cales -- Fortran host of the MI355X hot path. Keeps the CaNS/CaLES case interface (input.nml, fld.bin,
! time.out, forcing.out, grid.bin/grid.out/geometry.out) and the control flow of the reference driver
! (reference src/main.f90:135-632), and hands every per-step operator to libcales_hip.so through
! ISO_C_BINDING (module cales_c).
!
! Two builds of this one source (cales_amd/fortran/Makefile):
!   cales      one rank, one GPU, no MPI
!   cales_mpi  -DCALES_MPI: one MPI rank per GPU (mpiexec -n P ./cales_mpi), y-slab decomposition as in cales_amd/decomp.py. The
!              device is chosen from the node-local rank (reference src/initmpi.f90:64-73), rank 0 creates the RCCL rendezvous
!              token and MPI_Bcast carries it (cales_comm_unique_id -> cales_comm_init_rccl), after which every exchange of the
!              time step (halo rows, all-to-all of the Poisson solve, reductions) is done by the library itself over RCCL/xGMI.
!              MPI is only used for that start-up, for the sums of the plane statistics and for the timing lines. Checkpoints
!              and field dumps keep the reference's byte layout (src/load.f90:20-153): every rank writes its rows of each plane
!              at their offset in the one shared file.
!
! usage: cales [impdiff]      (impdiff = 0 explicit [default], 1 implicit in x,y,z, 2 z-implicit: the reference's build switches
!                              _IMPDIFF/_IMPDIFF_1D are run-time here); reads ./input.nml
program cales
  use, intrinsic :: iso_c_binding
  use, intrinsic :: iso_fortran_env, only: int64
  use, intrinsic :: ieee_arithmetic, only: is_nan => ieee_is_nan
  use cales_c
  implicit none
#ifdef CALES_MPI
  include 'mpif.h'
#endif
#if defined(CALES_MPI)
#if defined(_SINGLE_PRECISION)
#define MPI_REAL_RP MPI_REAL
#else
#define MPI_REAL_RP MPI_DOUBLE_PRECISION
#endif
#endif
  integer, parameter :: rp = c_rp
  integer(int64), parameter :: rsz = storage_size(1._rp)/8             ! bytes per real in the stream files
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
  real(rp) :: dt,dti,dt_cfl,time,divtot,divmax,dpdl(3),meanvel(3),var(7),tw,dt12,dt12av,dt12min,dt12max
  integer(int64) :: c0,c1,crate,cstep0
  integer :: istep,iunit,ierr,k,m,impdiff,savecounter,rc
  integer :: myid,nranks,n2l,jlo          ! rank, number of ranks, rows of the slab, global row of local row 1 minus 1
  logical :: is_done,kill,is_chan,is_duct
  character(len=512) :: iomsg,arg
  character(len=100) :: filename
  character(len=7) :: fldnum
  character(len=4) :: chkptnum
  character(kind=c_char) :: cmsg(512), cplan(1024)
  character(kind=c_char) :: token(128)
#ifdef CALES_MPI
  integer :: comm_node,locrank,ndev
  real(rp) :: rbuf(3),sbuf(3)
#endif
  !
  myid = 0; nranks = 1
#ifdef CALES_MPI
  call MPI_INIT(ierr)
  call MPI_COMM_RANK(MPI_COMM_WORLD,myid,ierr)
  call MPI_COMM_SIZE(MPI_COMM_WORLD,nranks,ierr)
  ! one GPU per rank, from the rank's position on its node (src/initmpi.f90:64-73)
  call MPI_COMM_SPLIT_TYPE(MPI_COMM_WORLD,MPI_COMM_TYPE_SHARED,0,MPI_INFO_NULL,comm_node,ierr)
  call MPI_COMM_RANK(comm_node,locrank,ierr)
  if(cales_device_count(ndev) /= 0 .or. ndev < 1) then
    print*, 'ERROR: no HIP device (the hot path has no CPU fallback).'; call die
  end if
  if(cales_set_device(mod(locrank,ndev)) /= 0) then
    print*, 'ERROR: hipSetDevice failed on rank ', myid; call die
  end if
#endif
  impdiff = 0
  if(command_argument_count() >= 1) then
    call get_command_argument(1,arg); read(arg,*) impdiff
  end if
  !
  ! read parameter file (src/param.f90:88-157); every rank reads it, as in the reference
  !
  dt_f = -1.; sgstype = ''; lwm = 0; hwm = 0.
  open(newunit=iunit,file='input.nml',status='old',action='read',iostat=ierr,iomsg=iomsg)
  if(ierr /= 0) then
    if(myid == 0) print*, 'Error reading the input file: ', trim(iomsg)
    if(myid == 0) print*, 'Aborting...'
    call die
  end if
  read(iunit,nml=dns,iostat=ierr,iomsg=iomsg)
  if(ierr /= 0) then
    if(myid == 0) print*, 'Error reading dns namelist: ', trim(iomsg)
    if(myid == 0) print*, 'Aborting...'
    call die
  end if
  read(iunit,nml=les,iostat=ierr,iomsg=iomsg)
  if(ierr /= 0) then
    ! examples/dns/* of the reference close &les with '\': this run-time reports it after reading the values
    if(len_trim(sgstype) == 0) then
      if(myid == 0) print*, 'Error reading les namelist: ', trim(iomsg)
      if(myid == 0) print*, 'Aborting...'
      call die
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
  case('amd')                                                                  ! src/sgs.f90:381-382
    if(myid == 0) print*, 'ERROR: AMD model not yet implemented'
    call die
  case default                                                                 ! src/sgs.f90:383-384
    if(myid == 0) print*, 'ERROR: unknown SGS model'
    call die
  end select
  cs%lwm = reshape(lwm,[6]); cs%hwm = hwm; cs%impdiff = impdiff; cs%nranks = nranks; cs%rank = myid
  !
  ! a-priori input checks (src/sanity.f90:33-67)
  !
  if(.not.any(stop_type(:))) then
    if(myid == 0) print*, 'ERROR: stopping criterion not chosen.'
    call abortit
  end if
  if(mod(ng(2),nranks) /= 0) then
    if(myid == 0) print*, 'ERROR: ng(2) must be divisible by the number of ranks (y-slab decomposition).'
    call abortit
  end if
  if(cales_check_case(cs,cmsg,512) /= 0) then
    if(myid == 0) print*, 'ERROR: ', cstr(cmsg)
    call abortit
  end if
  n2l = ng(2)/nranks; jlo = myid*n2l
  ! the plane statistics of out1d.h90's default (out1d_single_point_chan) are those of a channel: walls in z, periodic x and y
  is_chan = all(cbcpre(:,1) == 'P') .and. all(cbcpre(:,2) == 'P') .and. all(cbcvel(:,3,3) == 'D')
  ! ... and the cross-stream maps of out2d_duct (the alternative its out1d.h90 names, src/out1d.h90:37) those of a duct along x: walls in y and z
  is_duct = all(cbcpre(:,1) == 'P') .and. all(cbcvel(:,2,2) == 'D') .and. all(cbcvel(:,3,3) == 'D')
  !
  if(myid == 0) then
    print*, '*******************************'
    print*, '*** Beginning of simulation ***'
    print*, '*******************************'
    if(nranks > 1) print*, '*** ', nranks, ' ranks, y-slabs of ', n2l, ' rows ***'
  end if
  allocate(dzc(0:ng(3)+1),dzf(0:ng(3)+1),zc(0:ng(3)+1),zf(0:ng(3)+1))
  rc = cales_initgrid(gtype,ng(3),gr,l(3),dzc,dzf,zc,zf)
  if(myid == 0) then
    open(newunit=iunit,file='grid.bin',action='write',form='unformatted',access='stream',status='replace')   ! main.f90:248-250
    write(iunit) dzc(1:ng(3)),dzf(1:ng(3)),zc(1:ng(3)),zf(1:ng(3)); close(iunit)
    open(newunit=iunit,file='grid.out')
    do k=0,ng(3)+1
      write(iunit,'(*(E16.7e3))') 0.,zf(k),zc(k),dzf(k),dzc(k)
    end do
    close(iunit)
    open(newunit=iunit,file='geometry.out'); write(iunit,*) ng(1),ng(2),ng(3); write(iunit,*) l(1),l(2),l(3); close(iunit)
  end if
  !
  allocate(u(0:ng(1)+1,0:n2l+1,0:ng(3)+1),v(0:ng(1)+1,0:n2l+1,0:ng(3)+1),w(0:ng(1)+1,0:n2l+1,0:ng(3)+1), &
           p(0:ng(1)+1,0:n2l+1,0:ng(3)+1),visct(0:ng(1)+1,0:n2l+1,0:ng(3)+1))
  u = 0.; v = 0.; w = 0.; p = 0.; visct = 0.
  ctx = c_null_ptr
  if(cales_create(cs,c_null_ptr,ctx) /= 0) then
    print*, 'ERROR: ', cstr_ptr(cales_last_error(c_null_ptr)); call die
  end if
#ifdef CALES_MPI
  ! exchanges by the library itself over RCCL: the token of rank 0 travels by MPI (include/cales.h, "multi-GPU").
  ! CALES_FORCE_COMM: also with one rank (exercises this start-up on a one-GPU machine)
  call get_environment_variable('CALES_FORCE_COMM',arg,status=ierr)
  if(nranks > 1 .or. ierr == 0) then
    token(:) = c_null_char
    if(myid == 0) then
      if(cales_comm_unique_id(token) /= 0) then
        print*, 'ERROR: RCCL not available (cales_comm_unique_id).'; call die
      end if
    end if
    call MPI_BCAST(token,128,MPI_BYTE,0,MPI_COMM_WORLD,ierr)
    call chk(cales_comm_init_rccl(ctx,token))
    if(myid == 0) print*, '*** RCCL communicator of ', nranks, ' rank(s) initialised ***'
  end if
#endif
  if(.not.restart) then
    istep = 0; time = 0.
    rc = cales_initflow_slab(cs,trim(inivel)//c_null_char,merge(1,0,is_wallturb),u,v,w,p)
    if(rc /= 0) then
      if(myid == 0) print*, 'ERROR: invalid name for initial velocity field'     ! src/initflow.f90:203-209
      if(myid == 0) print*, '*** Simulation aborted due to errors in the case file ***'
      call die
    end if
    if(myid == 0) print*, '*** Initial condition succesfully set ***'
  else
    call load_all('r','fld.bin')
    if(myid == 0) print*, '*** Checkpoint loaded at time = ', time, 'time step = ', istep, '. ***'
  end if
  call chk(cales_upload_state(ctx,u,v,w,p))
  call chk(cales_bounduvw(ctx,1,0)); call chk(cales_boundp(ctx,CALES_P,0))          ! main.f90:370-375
  call chk(cales_cmpt_sgs(ctx));     call chk(cales_boundp(ctx,CALES_VISCT,1))
  ! which fused / folded form of every operator the steps below take (cales_describe_plan; not part of the reference's log)
  if(cales_describe_plan(ctx,cplan,1024_c_int) == 0 .and. myid == 0) then
    k = 1
    do while(k < 1024 .and. cplan(k) /= c_null_char); k = k + 1; end do
    write(*,'(a,*(a1))') ' *** Path of a time step: ', cplan(1:k-1)
  end if
  !
  ! post-process and write initial condition (main.f90:377-395)
  !
  write(fldnum,'(i7.7)') istep
  call write_outputs(.true.)
  !
  call chk(cales_chkdt(ctx,dt_cfl))
  dt = merge(dt_f,min(cfl*dt_cfl,dtmax),dt_f > 0.)
  if(myid == 0) print*, 'dt_cfl = ', dt_cfl, 'dt = ', dt
  dti = 1./dt
  kill = .false.; savecounter = 0
  call system_clock(c0,crate); cstep0 = c0
  !
  ! main loop (src/main.f90:403-619)
  !
  if(myid == 0) print*, '*** Calculation loop starts now ***'
  is_done = .false.
  do while(.not.is_done)
    call system_clock(cstep0)
    istep = istep + 1
    time = time + dt
    if(myid == 0) print*, 'Time step #', istep, 'Time = ', time
    call chk(cales_step(ctx,dt))            ! 3 RK substeps, main.f90:417-508
    if(stop_type(1)) then
      if(istep >= nstep   ) is_done = is_done.or..true.
    end if
    if(stop_type(2)) then
      if(time  >= time_max) is_done = is_done.or..true.
    end if
    if(stop_type(3)) then
      call system_clock(c1); tw = real(c1-c0,rp)/real(crate,rp)/3600.
#ifdef CALES_MPI
      sbuf(1) = tw; call MPI_ALLREDUCE(sbuf,rbuf,1,MPI_REAL_RP,MPI_MAX,MPI_COMM_WORLD,ierr); tw = rbuf(1)     ! every rank stops together
#endif
      if(tw    >= tw_max  ) is_done = is_done.or..true.
    end if
    if(icheck > 0.and.mod(istep,max(icheck,1)) == 0) then
      if(myid == 0) print*, 'Checking stability and divergence...'
      call chk(cales_chkdt(ctx,dt_cfl))
      dt = merge(dt_f,min(cfl*dt_cfl,dtmax),dt_f > 0.)
      if(myid == 0) print*, 'dt_cfl = ', dt_cfl, 'dt = ', dt
      if(dt_cfl < small) then
        if(myid == 0) print*, 'ERROR: time step is too small.'
        if(myid == 0) print*, 'Aborting...'
        is_done = .true.; kill = .true.
      end if
      dti = 1./dt
      call chk(cales_chkdiv(ctx,divtot,divmax))
      if(myid == 0) print*, 'Total divergence = ', divtot, '| Maximum divergence = ', divmax
      if(divmax > small.or.is_nan(divtot)) then
        if(myid == 0) print*, 'ERROR: maximum divergence is too large.'
        if(myid == 0) print*, 'Aborting...'
        is_done = .true.; kill = .true.
      end if
    end if
    write(fldnum,'(i7.7)') istep
    call write_outputs(.false.)
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
          if(myid == 0) call out0d('log_checkpoints.out',3,var)
        end if
        if(myid == 0) call execute_command_line('ln -sf '//trim(filename)//' fld.bin')
      end if
      call chk(cales_download_state(ctx,u,v,w,p,visct))
      call load_all('w',trim(filename))
      if(myid == 0) print*, '*** Checkpoint saved at time = ', time, 'time step = ', istep, '. ***'
    end if
    call chk(cales_sync(ctx))
    call system_clock(c1); dt12 = real(c1-cstep0,rp)/real(crate,rp)
    dt12av = dt12; dt12min = dt12; dt12max = dt12
#ifdef CALES_MPI
    sbuf(1) = dt12; call MPI_ALLREDUCE(sbuf,rbuf,1,MPI_REAL_RP,MPI_SUM,MPI_COMM_WORLD,ierr); dt12av  = rbuf(1)/(1.*nranks)   ! main.f90:612-618
    sbuf(1) = dt12; call MPI_ALLREDUCE(sbuf,rbuf,1,MPI_REAL_RP,MPI_MIN,MPI_COMM_WORLD,ierr); dt12min = rbuf(1)
    sbuf(1) = dt12; call MPI_ALLREDUCE(sbuf,rbuf,1,MPI_REAL_RP,MPI_MAX,MPI_COMM_WORLD,ierr); dt12max = rbuf(1)
#endif
    if(myid == 0) print*, 'Avrg, min & max elapsed time: '
    if(myid == 0) print*, dt12av,dt12min,dt12max
  end do
  call cales_destroy(ctx)
  if(myid == 0.and..not.kill) print*, '*** Fim ***'
#ifdef CALES_MPI
  call MPI_FINALIZE(ierr)
#endif
contains
  subroutine write_outputs(is_initial)     ! main.f90:377-395 (before the loop, every iout* > 0) and :548-589 (inside it)
    logical, intent(in) :: is_initial
    logical :: do0d,do1d,do2d,do3d
    do0d = iout0d > 0.and.mod(istep,max(iout0d,1)) == 0; do1d = iout1d > 0.and.mod(istep,max(iout1d,1)) == 0
    do2d = iout2d > 0.and.mod(istep,max(iout2d,1)) == 0; do3d = iout3d > 0.and.mod(istep,max(iout3d,1)) == 0
    if(do0d.and..not.is_initial) then     ! main.f90:548-573 (the reference writes no 0-d line for the initial field)
      var(1) = 1.*istep; var(2) = dt; var(3) = time
      if(myid == 0) call out0d('time.out',3,var)
      if(any(is_forced(:)).or.any(abs(bforce(:)) > 0.)) then
        meanvel(:) = 0.
        do m=1,3
          if(is_forced(m).or.abs(bforce(m)) > 0.) call chk(cales_bulk_mean(ctx,m-1,merge(0,1,m==3),meanvel(m)))
        end do
        call chk(cales_get_dpdl(ctx,dpdl))
        if(.not.any(is_forced(:))) dpdl(:) = -bforce(:)
        var(1) = time; var(2:4) = dpdl(1:3); var(5:7) = meanvel(1:3)
        if(myid == 0) call out0d('forcing.out',7,var)
      end if
    end if
    ! out1d.h90 is a case-specific include of the reference; its default computes the channel statistics, which only mean
    ! something for a channel (walls in z, periodic x and y): other cases get no velstats files here
    if(do1d.and.is_chan) call out1d_chan_stats('velstats_fld_'//fldnum)
    if(do1d.and.is_duct) call out2d_duct_stats('velstats_fld_'//fldnum//'.out')
    if(do2d.or.do3d) then     ! main.f90:580-589
      call chk(cales_download_state(ctx,u,v,w,p,visct))
      if(do2d) then     ! out2d.h90: the plane j = ng(2)/2 of the five fields
        call visu_2d('vex_slice_fld_'//fldnum//'.bin','Velocity_X',u); call visu_2d('vey_slice_fld_'//fldnum//'.bin','Velocity_Y',v)
        call visu_2d('vez_slice_fld_'//fldnum//'.bin','Velocity_Z',w); call visu_2d('pre_slice_fld_'//fldnum//'.bin','Pressure_P',p)
        call visu_2d('visct_slice_fld_'//fldnum//'.bin','Viscosity',visct)
      end if
      if(do3d) then     ! out3d.h90: the five fields without halos
        call visu_3d('vex_fld_'//fldnum//'.bin','Velocity_X',u); call visu_3d('vey_fld_'//fldnum//'.bin','Velocity_Y',v)
        call visu_3d('vez_fld_'//fldnum//'.bin','Velocity_Z',w); call visu_3d('pre_fld_'//fldnum//'.bin','Pressure',p)
        call visu_3d('visct_fld_'//fldnum//'.bin','Viscosity',visct)
      end if
    end if
  end subroutine write_outputs
  subroutine chk(ist)
    integer(c_int), intent(in) :: ist
    if(ist /= 0) then
      print*, 'ERROR (libcales_hip), rank ', myid, ': ', cstr_ptr(cales_last_error(ctx)); call die
    end if
  end subroutine chk
  subroutine die
#ifdef CALES_MPI
    integer :: ie
    call MPI_ABORT(MPI_COMM_WORLD,1,ie)
#endif
    error stop
  end subroutine die
  subroutine abortit
    if(myid == 0) then
      print*, ''
      print*, '*** Simulation aborted due to errors in the input file ***'
      print*, '    check `input.nml`.'
    end if
    call die
  end subroutine abortit
  subroutine barrier
#ifdef CALES_MPI
    integer :: ie
    call MPI_BARRIER(MPI_COMM_WORLD,ie)
#endif
  end subroutine barrier
  subroutine allsum(a,n)      ! sum over the ranks, result everywhere (plane statistics, output.f90:691)
    integer, intent(in) :: n
    real(rp), intent(inout) :: a(n)
#ifdef CALES_MPI
    real(rp), allocatable :: t(:)
    integer :: ie
    if(nranks == 1) return
    allocate(t(n)); t(:) = a(:)
    call MPI_ALLREDUCE(t,a,n,MPI_REAL_RP,MPI_SUM,MPI_COMM_WORLD,ie)
#endif
  end subroutine allsum
  subroutine visu_log(flog,fbin,varname,nmin,nmax)    ! write_log_output, src/output.f90:244-272
    character(len=*), intent(in) :: flog,fbin,varname
    integer, intent(in) :: nmin(3),nmax(3)
    integer :: iu
    if(myid /= 0) return
    open(newunit=iu,file=flog,position='append')
    write(iu,'(A30,A15,9I5,E16.7E3,I7)') fbin,varname,nmin,nmax,[1,1,1],time,istep
    close(iu)
  end subroutine visu_log
  subroutine visu_2d(fbin,varname,q)    ! write_visu_2d with inorm = 2, islice = ng(2)/2, src/output.f90:289-315
    character(len=*), intent(in) :: fbin,varname
    real(rp), intent(in) :: q(0:,0:,0:)
    integer :: iu,js
    js = ng(2)/2
    if(js > jlo .and. js <= jlo+n2l) then      ! the rank that owns the row writes the plane
      open(newunit=iu,file=fbin,access='stream',status='replace'); write(iu) q(1:ng(1),js-jlo,1:ng(3)); close(iu)
    end if
    call visu_log('log_visu_2d_slice_1.out',fbin,varname,[1,js,1],[ng(1),js,ng(3)])
  end subroutine visu_2d
  subroutine visu_3d(fbin,varname,q)    ! write_visu_3d with nskip = 1, src/output.f90:274-287
    character(len=*), intent(in) :: fbin,varname
    real(rp), intent(in) :: q(0:,0:,0:)
    integer :: iu
    call open_shared(fbin,iu)
    call write_slab(iu,0_int64,q)
    close(iu)
    call visu_log('log_visu_3d.out',fbin,varname,[1,1,1],ng)
  end subroutine visu_3d
  subroutine out1d_chan_stats(fname)    ! the velstats_fld_*.out/.bin pair of out1d_single_point_chan, src/output.f90:683-699
    character(len=*), intent(in) :: fname
    real(rp), allocatable :: buf(:,:),leak(:,:)
    integer :: iu,kk,q
    allocate(buf(27,ng(3)))
    call chk(cales_out1d_single_point_chan(ctx,buf))
    call allsum(buf,27*ng(3))
    if(myid == 0) then
      open(newunit=iu,file=fname//'.out')
      do kk=1,ng(3)
        write(iu,'(*(es24.16e3,1x))') zc(kk),zf(kk),(buf(q,kk),q=1,27),dzc(kk),dzf(kk)
      end do
      close(iu)
      open(newunit=iu,file=fname//'.bin',access='stream'); write(iu) buf; close(iu)
    end if
    deallocate(buf)
    allocate(buf(38,ng(3)),leak(6,ng(3)))      ! budgets and leakage, src/output.f90:990-1055
    call chk(cales_out1d_chan_budgets(ctx,buf,leak))
    call allsum(buf,38*ng(3)); call allsum(leak,6*ng(3))
    if(myid /= 0) return
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
  subroutine out2d_duct_stats(fname)    ! out2d_duct with the streamwise direction x, src/output.f90:406-507: its one text file, its format
    character(len=*), intent(in) :: fname
    real(rp), allocatable :: loc(:,:,:),glob(:,:,:)
    integer :: iu,jj,kk,q
    allocate(loc(9,n2l,ng(3)),glob(9,ng(2),ng(3)))
    call chk(cales_out2d_duct(ctx,loc))
    glob(:,:,:) = 0.; glob(:,jlo+1:jlo+n2l,:) = loc(:,:,:)      ! every rank holds its rows: the sum over ranks of output.f90:481-489
    call allsum(glob,9*ng(2)*ng(3))
    if(myid /= 0) return
    open(newunit=iu,file=fname)
    do kk=1,ng(3)
      do jj=1,ng(2)
        write(iu,'(11E16.7e3)') (jj-0.5)*(l(2)/(1.*ng(2))),zc(kk),(glob(q,jj,kk),q=1,9)
      end do
    end do
    close(iu)
  end subroutine out2d_duct_stats
  subroutine out0d(fname,n,vv)    ! src/output.f90:18-37
    character(len=*), intent(in) :: fname
    integer, intent(in) :: n
    real(rp), intent(in) :: vv(:)
    integer :: iu
    open(newunit=iu,file=fname,position='append')
    write(iu,'(*(E16.7e3))') vv(1:n)
    close(iu)
  end subroutine out0d
  ! ---- one file shared by the ranks: the global array a(1:ng1,1:ng2,1:ng3) in Fortran order, as the reference's MPI-IO subarray
  ! views produce it (src/load.f90:71-131). A y-slab is, for every plane k, one contiguous block of ng1*n2l values.
  subroutine open_shared(fname,iu)
    character(len=*), intent(in) :: fname
    integer, intent(out) :: iu
    if(myid == 0) then      ! rank 0 creates (truncates) the file, then everybody opens it
      open(newunit=iu,file=fname,action='write',form='unformatted',access='stream',status='replace'); close(iu)
    end if
    call barrier
    open(newunit=iu,file=fname,action='readwrite',form='unformatted',access='stream',status='old')
  end subroutine open_shared
  subroutine write_slab(iu,ifld,q)      ! field number ifld (0-based) of a file of fields
    integer, intent(in) :: iu
    integer(int64), intent(in) :: ifld
    real(rp), intent(in) :: q(0:,0:,0:)
    integer(int64) :: pos
    integer :: kk
    do kk=1,ng(3)
      pos = 1 + rsz*(((ifld*ng(3) + (kk-1))*ng(2) + jlo)*int(ng(1),int64))
      write(iu,pos=pos) q(1:ng(1),1:n2l,kk)
    end do
  end subroutine write_slab
  subroutine read_slab(iu,ifld,q)
    integer, intent(in) :: iu
    integer(int64), intent(in) :: ifld
    real(rp), intent(inout) :: q(0:,0:,0:)
    integer(int64) :: pos
    integer :: kk
    do kk=1,ng(3)
      pos = 1 + rsz*(((ifld*ng(3) + (kk-1))*ng(2) + jlo)*int(ng(1),int64))
      read(iu,pos=pos) q(1:ng(1),1:n2l,kk)
    end do
  end subroutine read_slab
  subroutine load_all(io,fname)   ! byte layout of src/load.f90:20-153: u,v,w,p (no halos) then [time, real(istep)]
    character(len=1), intent(in) :: io
    character(len=*), intent(in) :: fname
    integer :: iu
    integer(int64) :: fsize,good,nfld
    real(rp) :: fldinfo(2)
    nfld = int(ng(1),int64)*ng(2)*ng(3)
    select case(io)
    case('r')
      inquire(file=fname,size=fsize)
      good = (nfld*4+2)*rsz                    ! (4*N+2)*sizeof(rp), load.f90:44-52
      if(fsize /= good) then
        if(myid == 0) print*, '*** Simulation aborted due a checkpoint file with incorrect size ***'
        if(myid == 0) print*, '    file: ', fname, ' | expected size: ', good, '| actual size: ', fsize
        call die
      end if
      open(newunit=iu,file=fname,action='read',form='unformatted',access='stream',status='old')
      call read_slab(iu,0_int64,u); call read_slab(iu,1_int64,v); call read_slab(iu,2_int64,w); call read_slab(iu,3_int64,p)
      read(iu,pos=1+rsz*4*nfld) fldinfo
      close(iu)
      time = fldinfo(1); istep = nint(fldinfo(2))
    case('w')
      call open_shared(fname,iu)
      call write_slab(iu,0_int64,u); call write_slab(iu,1_int64,v); call write_slab(iu,2_int64,w); call write_slab(iu,3_int64,p)
      fldinfo = [time,1._rp*istep]
      if(myid == 0) write(iu,pos=1+rsz*4*nfld) fldinfo
      close(iu)
      call barrier
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
