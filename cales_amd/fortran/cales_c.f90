! ISO_C_BINDING view of include/cales.h -- what a Fortran host (the reference's own language) binds to.
! `type(cales_case)` mirrors `struct cales_case`; character and multi-dimensional members keep the
! storage order of the reference's namelist variables (src/param.f90:37-76), so they are copied unchanged.
module cales_c
  use, intrinsic :: iso_c_binding
  implicit none
  public
  ! rp of the reference (src/precision.f90:11-20): cales_real of include/cales.h. -D_SINGLE_PRECISION binds libcales_hip_sp.so.
#if defined(_SINGLE_PRECISION)
  integer, parameter :: c_rp = c_float
#else
  integer, parameter :: c_rp = c_double
#endif
  integer(c_int), parameter :: CALES_U = 0, CALES_V = 1, CALES_W = 2, CALES_P = 3, CALES_PP = 4, CALES_VISCT = 5
  type, bind(C) :: cales_case
    integer(c_int32_t) :: ng(3)
    real(c_rp)     :: l(3)
    integer(c_int32_t) :: gtype
    real(c_rp)     :: gr
    real(c_rp)     :: visci
    character(kind=c_char) :: cbcvel(18),cbcpre(6),cbcsgs(6)
    real(c_rp)     :: bcvel(18),bcpre(6),bcsgs(6)
    real(c_rp)     :: bforce(3)
    integer(c_int32_t) :: is_forced(3)
    real(c_rp)     :: velf(3)
    integer(c_int32_t) :: sgstype
    integer(c_int32_t) :: lwm(6)
    real(c_rp)     :: hwm
    integer(c_int32_t) :: impdiff
    integer(c_int32_t) :: nranks,rank
  end type cales_case
  interface
    integer(c_int) function cales_initgrid(gtype,n,gr,lz,dzc,dzf,zc,zf) bind(C,name='cales_initgrid')
      import; integer(c_int), value :: gtype,n; real(c_rp), value :: gr,lz; real(c_rp) :: dzc(*),dzf(*),zc(*),zf(*)
    end function
    integer(c_int) function cales_initflow(c,inivel,is_wallturb,u,v,w,p) bind(C,name='cales_initflow')
      import; type(cales_case), intent(in) :: c; character(kind=c_char) :: inivel(*); integer(c_int), value :: is_wallturb
      real(c_rp) :: u(*),v(*),w(*),p(*)
    end function
    integer(c_int) function cales_check_case(c,msg,msglen) bind(C,name='cales_check_case')
      import; type(cales_case), intent(in) :: c; character(kind=c_char) :: msg(*); integer(c_int), value :: msglen
    end function
    integer(c_int) function cales_create(c,stream,ctx) bind(C,name='cales_create')
      import; type(cales_case), intent(in) :: c; type(c_ptr), value :: stream; type(c_ptr) :: ctx
    end function
    subroutine cales_destroy(ctx) bind(C,name='cales_destroy')
      import; type(c_ptr), value :: ctx
    end subroutine
    type(c_ptr) function cales_last_error(ctx) bind(C,name='cales_last_error')
      import; type(c_ptr), value :: ctx
    end function
    integer(c_int) function cales_upload_state(ctx,u,v,w,p) bind(C,name='cales_upload_state')
      import; type(c_ptr), value :: ctx; real(c_rp) :: u(*),v(*),w(*),p(*)
    end function
    integer(c_int) function cales_download_state(ctx,u,v,w,p,visct) bind(C,name='cales_download_state')
      import; type(c_ptr), value :: ctx; real(c_rp) :: u(*),v(*),w(*),p(*),visct(*)
    end function
    integer(c_int) function cales_bounduvw(ctx,is_updt_wm,is_correc) bind(C,name='cales_bounduvw')
      import; type(c_ptr), value :: ctx; integer(c_int), value :: is_updt_wm,is_correc
    end function
    integer(c_int) function cales_boundp(ctx,field,which) bind(C,name='cales_boundp')
      import; type(c_ptr), value :: ctx; integer(c_int), value :: field,which
    end function
    integer(c_int) function cales_cmpt_sgs(ctx) bind(C,name='cales_cmpt_sgs')
      import; type(c_ptr), value :: ctx
    end function
    integer(c_int) function cales_chkdt(ctx,dtmax) bind(C,name='cales_chkdt')
      import; type(c_ptr), value :: ctx; real(c_rp) :: dtmax
    end function
    integer(c_int) function cales_chkdiv(ctx,divtot,divmax) bind(C,name='cales_chkdiv')
      import; type(c_ptr), value :: ctx; real(c_rp) :: divtot,divmax
    end function
    integer(c_int) function cales_out1d_single_point_chan(ctx,buf) bind(C,name='cales_out1d_single_point_chan')
      import; type(c_ptr), value :: ctx; real(c_rp) :: buf(27,*)
    end function
    integer(c_int) function cales_out1d_chan_budgets(ctx,budget,leakage) bind(C,name='cales_out1d_chan_budgets')
      import; type(c_ptr), value :: ctx; real(c_rp) :: budget(38,*),leakage(6,*)
    end function
    integer(c_int) function cales_out1d(ctx,field,idir,use_dzc,buf) bind(C,name='cales_out1d')
      import; type(c_ptr), value :: ctx; integer(c_int), value :: field,idir,use_dzc; real(c_rp) :: buf(*)
    end function
    integer(c_int) function cales_out1d_chan(ctx,buf) bind(C,name='cales_out1d_chan')
      import; type(c_ptr), value :: ctx; real(c_rp) :: buf(7,*)
    end function
    integer(c_int) function cales_out2d_duct(ctx,buf) bind(C,name='cales_out2d_duct')
      import; type(c_ptr), value :: ctx; real(c_rp) :: buf(9,*)
    end function
    integer(c_int) function cales_step(ctx,dt) bind(C,name='cales_step')
      import; type(c_ptr), value :: ctx; real(c_rp), value :: dt
    end function
    integer(c_int) function cales_get_dpdl(ctx,dpdl) bind(C,name='cales_get_dpdl')
      import; type(c_ptr), value :: ctx; real(c_rp) :: dpdl(3)
    end function
    integer(c_int) function cales_bulk_mean(ctx,field,c_or_f,mean) bind(C,name='cales_bulk_mean')
      import; type(c_ptr), value :: ctx; integer(c_int), value :: field,c_or_f; real(c_rp) :: mean
    end function
    integer(c_int) function cales_sync(ctx) bind(C,name='cales_sync')
      import; type(c_ptr), value :: ctx
    end function
    ! the per-operator entries (cales_rk, cales_fillps, cales_solver, cales_correc, cales_updatep, ...) follow the
    ! same pattern; the driver below uses the fused cales_step, which queues exactly their sequence.
    integer(c_int) function cales_rk(ctx,irk,dt) bind(C,name='cales_rk')
      import; type(c_ptr), value :: ctx; integer(c_int), value :: irk; real(c_rp), value :: dt
    end function
    integer(c_int) function cales_rk_par(ctx,rkpar,dt,f) bind(C,name='cales_rk_par')      ! rk(rkpar,...,dt,...,f), src/rk.f90:17
      import; type(c_ptr), value :: ctx; real(c_rp), intent(in) :: rkpar(2); real(c_rp), value :: dt; real(c_rp) :: f(3)
    end function
    ! the path the next cales_step takes, as text "key=value;..." (struct StepPlan; the sequence it protects: src/main.f90:417-508)
    integer(c_int) function cales_describe_plan(ctx,buf,buflen) bind(C,name='cales_describe_plan')
      import; type(c_ptr), value :: ctx; character(kind=c_char) :: buf(*); integer(c_int), value :: buflen
    end function
    ! same-box calibration: read / write / copy streams over the context's own fields, GB/s (bench line)
    integer(c_int) function cales_calibrate(ctx,reps,gbps,nbytes) bind(C,name='cales_calibrate')
      import; type(c_ptr), value :: ctx; integer(c_int), value :: reps; real(c_rp) :: gbps(3); integer(c_int64_t) :: nbytes
    end function
    integer(c_int) function cales_fillps(ctx,dtrki) bind(C,name='cales_fillps')
      import; type(c_ptr), value :: ctx; real(c_rp), value :: dtrki
    end function
    integer(c_int) function cales_solver(ctx) bind(C,name='cales_solver')
      import; type(c_ptr), value :: ctx
    end function
    integer(c_int) function cales_correc(ctx,dtrk) bind(C,name='cales_correc')
      import; type(c_ptr), value :: ctx; real(c_rp), value :: dtrk
    end function
    integer(c_int) function cales_updatep(ctx,alpha) bind(C,name='cales_updatep')
      import; type(c_ptr), value :: ctx; real(c_rp), value :: alpha
    end function
    integer(c_int) function cales_bulk_forcing(ctx) bind(C,name='cales_bulk_forcing')
      import; type(c_ptr), value :: ctx
    end function
    integer(c_int) function cales_updt_rhs_b(ctx) bind(C,name='cales_updt_rhs_b')
      import; type(c_ptr), value :: ctx
    end function
    integer(c_int) function cales_helmholtz_z(ctx,ivel,alpha) bind(C,name='cales_helmholtz_z')   ! _IMPDIFF + _IMPDIFF_1D
      import; type(c_ptr), value :: ctx; integer(c_int), value :: ivel; real(c_rp), value :: alpha
    end function
    integer(c_int) function cales_helmholtz(ctx,ivel,alpha) bind(C,name='cales_helmholtz')       ! _IMPDIFF (periodic x,y)
      import; type(c_ptr), value :: ctx; integer(c_int), value :: ivel; real(c_rp), value :: alpha
    end function
    ! multi-GPU, exchanges done by the library with RCCL: rank 0 fills id(128), the host broadcasts it
    ! (call MPI_Bcast(id,128,MPI_BYTE,0,comm,ierr)), every rank joins
    integer(c_int) function cales_comm_unique_id(id) bind(C,name='cales_comm_unique_id')
      import; character(kind=c_char) :: id(128)
    end function
    integer(c_int) function cales_comm_init_rccl(ctx,id) bind(C,name='cales_comm_init_rccl')
      import; type(c_ptr), value :: ctx; character(kind=c_char), intent(in) :: id(128)
    end function
    integer(c_int) function cales_device_count(ndev) bind(C,name='cales_device_count')
      import; integer(c_int) :: ndev
    end function
    integer(c_int) function cales_set_device(dev) bind(C,name='cales_set_device')
      import; integer(c_int), value :: dev
    end function
    ! the rank's rows of the initial field: local haloed arrays (0:n1+1,0:n2/nranks+1,0:n3+1)
    integer(c_int) function cales_initflow_slab(c,inivel,is_wallturb,u,v,w,p) bind(C,name='cales_initflow_slab')
      import; type(cales_case), intent(in) :: c; character(kind=c_char) :: inivel(*); integer(c_int), value :: is_wallturb
      real(c_rp) :: u(*),v(*),w(*),p(*)
    end function
  end interface
end module cales_c
