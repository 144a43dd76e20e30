/* TEST INFRASTRUCTURE -- CPU restatement (plain C, FP64) of CaLES's per-step hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * It is the checker, never the product: libcales_hip.so does not link or call it.
 *
 * Pinning: every operator below that exists in a reference module buildable in this
 * image is checked against golden vectors produced by the reference itself
 * (oracle/ref -> oracle/_ref, tests/golden/gen_golden.py, tests/test_oracle_golden.py).
 * PARITY UNPINNED at one boundary: the r2r transforms (FFTW 3.x, not vendored,
 * src/fft.f90:70-85,179-190) and the call order of the solver driver around them
 * (src/solver.f90:20-80, needs 2decomp-fft). The transforms are restated from the FFTW
 * manual's definitions, checked against scipy.fft (same conventions) and by the discrete
 * identity L_h(solve(r)) = r built from the reference's own fillps/correc stencils. The
 * plain-arithmetic parts of the solver ARE pinned to the reference's own routines, compiled
 * from their lines (oracle/ref/Makefile): eigenvalues, tridmatrix (initsolver.f90:66-169),
 * gaussel, gaussel_periodic, dgtsv_homebrewed (solver.f90:82-179), and with them all of
 * solver_gaussel_z (a transposition around gaussel).
 *
 * All 3-D arrays are Fortran-ordered with one halo cell: a(0:n1+1,0:n2+1,0:n3+1).
 */
#ifndef CALES_ORACLE_H
#define CALES_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int    ng[3];
  double l[3];
  int    gtype; double gr;
  double visci;
  char   cbcvel[18];  /* (side,dir,vel) Fortran order: side + 2*dir + 6*vel */
  char   cbcpre[6];   /* (side,dir) */
  char   cbcsgs[6];
  double bcvel[18], bcpre[6], bcsgs[6];
  double bforce[3]; int is_forced[3]; double velf[3];
  int    sgstype;     /* 0 none, 1 smag, 2 dsmag */
  int    lwm[6];      /* (side,dir) */
  double hwm;
  int    impdiff;     /* 0 explicit, 1 3-D implicit (_IMPDIFF), 2 z-implicit (_IMPDIFF_1D) */
  int    nthreads;    /* OpenMP threads for the cell loops (reductions stay serial) */
} oparams;

typedef struct ostate ostate;

ostate *o_create(const oparams *p);
void    o_destroy(ostate *s);
/* sums over all cells (bulk mean, total divergence): 0 (default) = cell by cell in the reference's order on one rank, what the golden vectors pin;
   1 = plane by plane over the OpenMP team (bench.py's timed CPU baseline: a serial sum over 1.3e8 cells three times a step would idle 127 cores) */
void    o_set_team_sums(ostate *s, int on);

/* set-up products (src/initgrid.f90, src/bound.f90:726-867, src/initsolver.f90) */
void o_first_touch(const ostate *s, double *a);
void o_get_grid(const ostate *s, double *dzc, double *dzf, double *zc, double *zf);
void o_get_index_wm(const ostate *s, int *index_wm);
void o_get_cbcvel(const ostate *s, char *cbcvel);
void o_get_rhsbp(const ostate *s, double *x, double *y, double *z);
void o_get_bcvel(const ostate *s, int ivel, double *x, double *y, double *z);
void o_get_solver(const ostate *s, int which, double *lambdaxy, double *a, double *b, double *c, double *normfft);
void o_initgrid(int gtype, int n, double gr, double lz, double *dzc, double *dzf, double *zc, double *zf);
int  o_initflow(ostate *s, const char *inivel, int is_wallturb, double *u, double *v, double *w, double *p);

/* operators, same order of arguments as the reference where it helps reading the tests */
void o_bounduvw(ostate *s, int is_updt_wm, int is_correc, double *u, double *v, double *w);
void o_boundp(ostate *s, int which, double *p);                /* 0: cbcpre/bcp, 1: cbcsgs/bcs */
void o_mom(ostate *s, const double *u, const double *v, const double *w, const double *visct,
           double *dudt, double *dvdt, double *dwdt, double *dudtd, double *dvdtd, double *dwdtd);
void o_rk(ostate *s, int irk, double dt, const double *p, const double *visct,
          double *u, double *v, double *w, double *f);
void o_bulk_forcing(ostate *s, const double *f, double *u, double *v, double *w);
double o_bulk_mean(ostate *s, int c_or_f, const double *p);
void o_fillps(ostate *s, double dtrki, const double *u, const double *v, const double *w, double *pp);
void o_updt_rhs_b_p(ostate *s, double *pp);
void o_updt_rhs_b_vel(ostate *s, int ivel, double alpha, double *q);
void o_updt_rhs_b_velz(ostate *s, int ivel, double alpha, double *q);
void o_solver(ostate *s, double *pp);                          /* Poisson, cbcpre, 'c','c','c' */
void o_solver_zsweep(ostate *s, double *pp);                   /* its tridiagonal sweep alone (solver.f90:56-61,82-151) */
void o_solver_gaussel_z(ostate *s, int ivel, double alpha, double *q);
int o_solver_helmholtz(ostate *s, int ivel, double alpha, double *q);   /* 3-D implicit diffusion, x and y periodic */
void o_correc(ostate *s, double dtrk, const double *pp, double *u, double *v, double *w);
void o_updatep(ostate *s, double alpha, const double *pp, double *p);
void o_cmpt_sgs(ostate *s, const double *u, const double *v, const double *w, double *visct);
double o_chkdt(ostate *s, const double *visct, const double *u, const double *v, const double *w);
void o_chkdiv(ostate *s, const double *u, const double *v, const double *w, double *divtot, double *divmax);
void o_stats_chan(ostate *s, const double *u, const double *v, const double *w, const double *p, const double *visct, double *buf);
void o_out1d(ostate *s, int idir, int use_dzc, const double *p, double *buf);
void o_out1d_chan(ostate *s, const double *u, const double *v, const double *w, double *buf);
void o_out2d_duct(ostate *s, const double *u, const double *v, const double *w, double *buf);

/* one full time step = 3 RK substeps in the order of src/main.f90:417-507; f accumulates dpdl */
void o_step(ostate *s, double dt, double *u, double *v, double *w, double *p, double *pp, double *visct,
            double *dpdl);

/* r2r transforms with FFTW's definitions (unnormalised), in place, one line */
enum { O_R2HC = 0, O_HC2R = 1, O_REDFT00 = 3, O_REDFT01 = 4, O_REDFT10 = 5, O_REDFT11 = 6,
       O_RODFT00 = 7, O_RODFT01 = 8, O_RODFT10 = 9, O_RODFT11 = 10 };
void o_r2r(int kind, int n, double *x, int stride);

#ifdef __cplusplus
}
#endif
#endif
