/* cales.h -- C-ABI of the MI355X-native CaLES hot path (libcales_hip.so).
 *
 * Drop-in boundary: one entry per routine the reference driver calls inside its time loop
 * (reference src/main.f90:417-507). A Fortran host binds these with ISO_C_BINDING
 * (cales_amd/fortran/cales_c.f90). Plain pointers and sizes only.
 *
 * Conventions
 *  - every entry returns 0 on success, non-zero on error (cales_last_error() gives the text);
 *  - host fields are Fortran-ordered doubles WITH one halo cell, a(0:n1+1,0:n2+1,0:n3+1),
 *    n = local sizes of the rank's y-slab (n = ng for one rank);
 *  - all device work is queued on ONE HIP stream (the reference's single `async(1)` queue,
 *    src/main.f90:368); entries that return scalars synchronise, the others do not.
 */
#ifndef CALES_H
#define CALES_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* rp of the reference (src/precision.f90:11-20). libcales_hip.so is the FP64 build; libcales_hip_sp.so, built from the same sources with
 * -DCALES_SINGLE, is the reference's -D_SINGLE_PRECISION: every real of this interface (fields, parameters, scalars) is then a float.
 * cales_real_size() tells a host which one it has loaded. */
#ifdef CALES_SINGLE
typedef float cales_real;
#else
typedef double cales_real;
#endif
int cales_real_size(void);      /* 8 or 4 */

/* The run-time image of input.nml (reference src/param.f90:37-76,95-119). Character and
 * multi-dimensional entries keep Fortran storage order, so a Fortran host can pass its
 * namelist variables unchanged: cbcvel(0:1,3,3) -> [side + 2*dir + 6*vel]. */
typedef struct cales_case {
  int32_t ng[3];          /* global grid */
  cales_real  l[3];
  int32_t gtype; cales_real gr;
  cales_real  visci;
  char    cbcvel[18], cbcpre[6], cbcsgs[6];
  cales_real  bcvel[18], bcpre[6], bcsgs[6];
  cales_real  bforce[3]; int32_t is_forced[3]; cales_real velf[3];
  int32_t sgstype;        /* 0 'none', 1 'smag', 2 'dsmag'  (src/sgs.f90:61-153) */
  int32_t lwm[6]; cales_real hwm;
  int32_t impdiff;        /* 0 explicit; 2 = _IMPDIFF + _IMPDIFF_1D (z-implicit); 1 = _IMPDIFF (3-D implicit; periodic or no-slip wall pairs in x and y) */
  int32_t nranks, rank;   /* y-slab decomposition: rank owns rows rank*ng2/nranks+1 ... */
} cales_case;

typedef struct cales_ctx cales_ctx;

enum cales_field { CALES_U = 0, CALES_V = 1, CALES_W = 2, CALES_P = 3, CALES_PP = 4, CALES_VISCT = 5,
                   CALES_DUDT = 6, CALES_DVDT = 7, CALES_DWDT = 8,      /* current RK r.h.s. (rk.f90 dudtrk)  */
                   CALES_DUDTO = 9, CALES_DVDTO = 10, CALES_DWDTO = 11, /* previous RK r.h.s. (dudtrko)       */
                   CALES_DUDTD = 12, CALES_DVDTD = 13, CALES_DWDTD = 14,/* implicit part (dudtrkd)            */
                   CALES_NFIELDS = 15 };

/* ---- host-only helpers (no GPU needed) -------------------------------------------------- */
/* src/initgrid.f90:15  initgrid(gtype,n,gr,lz,dzc,dzf,zc,zf); arrays (0:n+1) */
int cales_initgrid(int gtype, int n, cales_real gr, cales_real lz, cales_real *dzc, cales_real *dzf, cales_real *zc, cales_real *zf);
/* src/initflow.f90:17  initflow(inivel,...,u,v,w,p): zer uni cou poi iop pdc hdc hcp tgv tgw ant duc bit-compatible with the
 * reference; log hcl tbl with their +-5 % noise from a counter-based generator (the reference uses the Fortran run-time's
 * random_number: same distribution, other numbers); global haloed host arrays; returns 1 for an unknown kind */
int cales_initflow(const cales_case *c, const char *inivel, int is_wallturb, cales_real *u, cales_real *v, cales_real *w, cales_real *p);
/* src/sanity.f90:33-67 rules restated; returns 0 if the case is accepted */
int cales_check_case(const cales_case *c, char *msg, int msglen);

/* ---- device selection (one process per GPU: a multi-rank host picks its device before cales_create; the reference does this in
 * src/initmpi.f90:64-73 with acc_set_device_num(mod(local rank, ndev))) ----------------------------- */
int cales_device_count(int *ndev);
int cales_set_device(int dev);

/* ---- context ------------------------------------------------------------------------------ */
/* stream: a hipStream_t to queue on, or NULL to let the context create its own */
int  cales_create(const cales_case *c, void *stream, cales_ctx **out);
void cales_destroy(cales_ctx *ctx);
const char *cales_last_error(const cales_ctx *ctx);    /* ctx may be NULL for create-time errors */
int  cales_sync(cales_ctx *ctx);
int  cales_local_size(const cales_ctx *ctx, int32_t n[3], int32_t lo[3]);

/* host <-> device (src/main.f90:368 `enter data copyin(u,v,w,p)`, :576-607 `update self`) */
int cales_upload_state(cales_ctx *ctx, const cales_real *u, const cales_real *v, const cales_real *w, const cales_real *p);
int cales_download_state(cales_ctx *ctx, cales_real *u, cales_real *v, cales_real *w, cales_real *p, cales_real *visct);
int cales_set_field(cales_ctx *ctx, int field, const cales_real *host);
int cales_get_field(cales_ctx *ctx, int field, cales_real *host);
int cales_get_bcvel(cales_ctx *ctx, int ivel, cales_real *x, cales_real *y, cales_real *z);  /* bcu/bcv/bcw planes (typedef.f90:10) */

/* ---- operators, named after the reference routines they replace ------------------------- */
int cales_bounduvw(cales_ctx *ctx, int is_updt_wm, int is_correc);          /* src/bound.f90:18   */
int cales_boundp(cales_ctx *ctx, int field, int which);                     /* src/bound.f90:156; which 0 cbcpre/bcp, 1 cbcsgs/bcs */
int cales_mom(cales_ctx *ctx);                                              /* src/mom.f90:17 -> CALES_DUDT.. */
int cales_rk(cales_ctx *ctx, int irk, cales_real dt);                           /* src/rk.f90:17 (forcing f stays on the device). The ghost
                                                                              * cells of u,v,w keep the values of the last bounduvw (as in the reference) when a
                                                                              * wall model is active -- it may sample them -- and are undefined otherwise;
                                                                              * bounduvw follows in every caller (src/main.f90:493) */
/* the same with the caller's Runge-Kutta coefficients: rk(rkpar, ..., dt, ..., f) of src/rk.f90:17 as the reference declares it
 * (rkpar = rkcoeff(:,irk), src/param.f90:27-29); f_out may be NULL, otherwise the forcing f(1:3) is returned (sync) */
int cales_rk_par(cales_ctx *ctx, const cales_real rkpar[2], cales_real dt, cales_real f_out[3]);
int cales_bulk_forcing(cales_ctx *ctx);                                     /* src/mom.f90:311    */
int cales_get_forcing(cales_ctx *ctx, cales_real f[3]);                         /* f of the last cales_rk (sync)  */
int cales_bulk_mean(cales_ctx *ctx, int field, int c_or_f, cales_real *mean);   /* src/utils.f90:16 (sync) */
int cales_fillps(cales_ctx *ctx, cales_real dtrki);                             /* src/fillps.f90:14  */
int cales_updt_rhs_b(cales_ctx *ctx);                                       /* src/bound.f90:562, pressure r.h.s. */
int cales_solver(cales_ctx *ctx);                                           /* src/solver.f90:20 on CALES_PP */
int cales_helmholtz_z(cales_ctx *ctx, int ivel, cales_real alpha);              /* main.f90:425-445: updt_rhs_b + solver_gaussel_z */
int cales_helmholtz(cales_ctx *ctx, int ivel, cales_real alpha);                /* impdiff = 1, main.f90:423-491: cmpt_rhs_b + updt_rhs_b (x, y, z faces) + solver on a velocity component, any BC pair of find_fft */
int cales_correc(cales_ctx *ctx, cales_real dtrk);                              /* src/correc.f90:14  */
int cales_updatep(cales_ctx *ctx, cales_real alpha);                            /* src/updatep.f90:14 */
int cales_cmpt_sgs(cales_ctx *ctx);                                         /* src/sgs.f90:21     */
int cales_chkdt(cales_ctx *ctx, cales_real *dtmax);                             /* src/chkdt.f90:17 (sync) */
int cales_chkdiv(cales_ctx *ctx, cales_real *divtot, cales_real *divmax);           /* src/chkdiv.f90:16 (sync) */
/* plane statistics of the channel: first block of out1d_single_point_chan (src/output.f90:509-700, idir = 3), the 27 columns of
 * velstats_fld_*.out/.bin between the coordinates and the spacings; buf(27, n3) column-major on the host (sync). With several
 * ranks: the sums over this rank's rows -- the caller adds the ranks as the reference does (output.f90:691). */
#define CALES_NSTATS_CHAN 27
int cales_out1d_single_point_chan(cales_ctx *ctx, cales_real *buf);
/* second and third block of the same routine (src/output.f90:700-1055): budget(38, n3) = the columns of velstats_fld_*_reystr_budget.out/.bin,
 * leakage(6, n3) = those of velstats_fld_*_leakage.out/.bin; either pointer may be NULL (sync) */
#define CALES_NBUDGET_CHAN 38
#define CALES_NLEAKAGE_CHAN 6
int cales_out1d_chan_budgets(cales_ctx *ctx, cales_real *budget, cales_real *leakage);
/* the other statistics routines a case may call from its out1d.h90 (src/out1d.h90:25-37), all sync, all returning THIS rank's sums (the caller
 * adds / concatenates the ranks as the reference's MPI_ALLREDUCE does):
 *   out1d      (src/output.f90:50-163): profile of `field` along idir (1, 2, 3) averaged over the other two directions, weighted with dzf(k)
 *              (use_dzc = 0, cell-centred in z) or dzc(k) (use_dzc = 1: w) where z is averaged over; buf(n(idir))
 *   out1d_chan (src/output.f90:317-405, idir = 3): um, vm, wm, u2, v2, w2, uw per plane; buf(7, n3)
 *   out2d_duct (src/output.f90:406-507, streamwise x): um, vm, wm, u2, v2, w2, uv, uw, vw at cell centres for every (j, k); buf(9, n2, n3) */
int cales_out1d(cales_ctx *ctx, int field, int idir, int use_dzc, cales_real *buf);
#define CALES_NSTATS_OUT1D_CHAN 7
int cales_out1d_chan(cales_ctx *ctx, cales_real *buf);
#define CALES_NSTATS_DUCT 9
int cales_out2d_duct(cales_ctx *ctx, cales_real *buf);

/* one time step = 3 RK substeps in the order of src/main.f90:417-508; no host synchronisation.
 * Without subgrid model (explicit diffusion, periodic / no-slip directions) the LAST projection of the step may still be pending when this returns --
 * the next cales_step applies it in its first momentum pass; every other entry that reads or writes a field, cales_sync included, completes it first,
 * so what a caller can observe is always the projected state of src/main.f90:498-504 (CALES_EAGER_PROJECTION). ONE RANK ONLY: on several
 * slabs completing a projection moves slab rows, so cales_step always completes it itself and no entry is a hidden collective. Three entries do not
 * complete a pending projection because they read no field: cales_get_forcing, cales_get_dpdl (scalars accumulated by the step) and cales_get_bcvel
 * (boundary planes the projection does not touch).
 * With periodic x (explicit diffusion, no wall model) the kernels of a step wrap around and the x GHOST COLUMNS are likewise brought up to date by the first
 * entry other than the next cales_step (local copies, any number of slabs; the same mechanism, the same three exceptions). */
int cales_step(cales_ctx *ctx, cales_real dt);
int cales_get_dpdl(cales_ctx *ctx, cales_real dpdl[3]);                         /* main.f90:492,508 (sync) */
/* The path the NEXT cales_step takes through the fused / folded forms of its operators, as text "key=value;key=value;..." (projection, x ghost columns,
 * fillps, bulk forcing, wall model, ghost cells, sgs, solver kernels, ranks, exchanges): decided once from the case, the CALES_* switches and the
 * context's state (struct StepPlan, cales_amd/csrc/common.hpp), re-made only when one of those changes, and only READ by the step -- the sequence it
 * protects is src/main.f90:417-508. Reads no field (a pending projection stays pending). Returns 0, or 2 when buf was too short (text truncated). */
int cales_describe_plan(cales_ctx *ctx, char *buf, int buflen);

/* ---- multi-GPU: y-slab decomposition --------------------------------------
 * The reference exchanges halos with MPI_SENDRECV / cudecompUpdateHalos (src/bound.f90:619-723) and transposes
 * pencils with 2decomp / cudecompTranspose* (src/solver.f90:50-66, src/solver_gpu.f90:97-125). Here the rank owns
 * a y-slab; the library packs/unpacks on the device and calls back into the host for the three exchanges, which the
 * host performs on its own communicator (torch.distributed/RCCL in cales_amd/decomp.py). Offsets are in doubles,
 * relative to the staging buffers A and B registered below; every callback must enqueue its work on the
 * context's stream (or order it after that stream) and return 0.
 *   halo:      send A[off_send_lo..+count) to the lower y-neighbour and A[off_send_hi..) to the upper one; receive the
 *              lower neighbour's "hi" block into B[off_recv_lo..) and the upper neighbour's "lo" block into B[off_recv_hi..)
 *   alltoall:  dir 0: A -> B, dir 1: B -> A; `count` doubles per peer, peer blocks contiguous in rank order
 *   allreduce: in place on A[off..off+count), op 0 sum, 1 max, 2 min */
typedef int (*cales_halo_cb)(void *user, int64_t off_send_lo, int64_t off_send_hi, int64_t off_recv_lo, int64_t off_recv_hi, int64_t count);
typedef int (*cales_alltoall_cb)(void *user, int dir, int64_t count);
typedef int (*cales_allreduce_cb)(void *user, int64_t off, int64_t count, int op);
int cales_comm_buffer_doubles(const cales_ctx *ctx, int64_t *n);      /* required size of A and of B */
int cales_set_comm(cales_ctx *ctx, cales_halo_cb halo, cales_alltoall_cb a2a, cales_allreduce_cb allred, void *user,
                   cales_real *bufA, cales_real *bufB, int64_t nbuf);
/* Optional, after cales_set_comm: exchanges that may run BESIDE the kernels (the reference's non-blocking halos, src/bound.f90:619-696
 * `_ASYNC_HALO`, and cuDecomp's pipelined transposes, src/initmpi.f90:94-139). The library owns a second HIP stream and the events
 * between the two; it hands that stream to these callbacks, which must enqueue ALL their work on it (no host synchronisation):
 *   halo_s:        as `halo`, on `stream`
 *   alltoall_part: a slice of the all-to-all: for every peer p, send src[p*peer_stride + off, +count) to p and receive p's slice into
 *                  dst[p*peer_stride + off, +count); dir 0: src = A, dst = B; dir 1: src = B, dst = A. The Poisson solve then sends the
 *                  spectrum in k-chunks: the x transforms of chunk c+1 and the y transforms of chunk c-1 run while chunk c travels;
 *                  the y halos of the dynamic model's twelve scratch fields travel while the interior tiles of its last pass run.
 * Without this call (or with NULL entries) every exchange is issued in order on the context's stream, as before. */
typedef int (*cales_halo_s_cb)(void *user, int64_t off_send_lo, int64_t off_send_hi, int64_t off_recv_lo, int64_t off_recv_hi, int64_t count, void *stream);
typedef int (*cales_alltoall_part_cb)(void *user, int dir, int64_t peer_stride, int64_t off, int64_t count, void *stream);
int cales_set_comm_overlap(cales_ctx *ctx, cales_halo_s_cb halo_s, cales_alltoall_part_cb a2a_part);
/* Native alternative to cales_set_comm: the library performs the three exchanges itself with RCCL (xGMI) on the context's
 * stream -- grouped ncclSend/ncclRecv for the halo rows, ncclAllToAll for the transposition of the Poisson solve (one pair
 * per solve where src/solver.f90:50-66 needs four pencil transposes), ncclAllReduce for the reductions. Rank 0 obtains the
 * rendezvous token, the host distributes it (MPI_Bcast, torch.distributed, a file ...), then every rank calls
 * cales_comm_init_rccl, which is collective. RCCL is opened at run time; without it these calls fail and cales_set_comm remains. */
#define CALES_COMM_ID_BYTES 128
int cales_comm_unique_id(void *id_out);                       /* rank 0 only; writes CALES_COMM_ID_BYTES bytes */
int cales_comm_init_rccl(cales_ctx *ctx, const void *id);     /* all ranks */
int cales_comm_selftest(void);                                /* one-rank communicator: halo order, all-to-all, all-reduce; 0 = ok */
/* initial field of the rank's slab only (local haloed arrays); the volume mean is summed in the global order */
int cales_initflow_slab(const cales_case *c, const char *inivel, int is_wallturb, cales_real *u, cales_real *v, cales_real *w, cales_real *p);

/* ---- measurement -------------------------------------------------------------------------- */
/* When enabled, every kernel launch is bracketed by HIP events on the context's stream. */
int cales_profile_enable(cales_ctx *ctx, int on);
int cales_profile_reset(cales_ctx *ctx);
int cales_profile_count(cales_ctx *ctx);
int cales_profile_get(cales_ctx *ctx, int idx, char *name, int namelen, int64_t *calls, cales_real *total_ms);
/* algorithmic words (8 B) per cell per call of the named kernel group, for the roofline line */
int cales_device_info(cales_ctx *ctx, char *name, int namelen, int64_t *hbm_bytes);
/* Same-box calibration for the roofline fractions: read-only, write-only and copy streams over the rows of the context's own fields in
 * the library's layout (8 B per lane), `reps` launches each after one warm-up, timed with HIP events on the context's stream; gbps[0..2] = read, write,
 * copy in GB/s (copy counts bytes read + written), *bytes_per_stream = bytes one launch reads (or writes). Overwrites a scratch field only. (sync) */
int cales_calibrate(cales_ctx *ctx, int reps, cales_real gbps[3], int64_t *bytes_per_stream);

#ifdef __cplusplus
}
#endif
#endif
