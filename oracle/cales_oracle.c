/* TEST INFRASTRUCTURE -- see cales_oracle.h. CPU restatement of the CaLES hot path.
 * Every function cites the reference lines it follows (paths relative to /root/reference).
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 * Expression order follows the Fortran so results are (nearly always) bit-identical
 * with the amdflang-compiled reference; Fortran default-real (single) literals and
 * integer arithmetic are reproduced where they change the value (initgrid, initflow).
 */
#include "cales_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PI (acos(-1.0))
#define EPS DBL_EPSILON            /* src/param.f90:20 */
#define BIG DBL_MAX                /* src/param.f90:25 */
#define C_SMAG 0.11                /* src/param.f90:33 */
#define KAP_LOG 0.41
#define B_LOG 5.20
static const double RKCOEFF[3][2] = {{32./60., 0.}, {25./60., -17./60.}, {45./60., -25./60.}}; /* param.f90:27-29 */

typedef struct { double *x, *y, *z; } obound;   /* src/typedef.f90:10-14; planes (0:na+1,0:nb+1,0:1) */

struct ostate {
  oparams P;
  int n[3];
  size_t s1, s2, ntot;              /* strides of haloed arrays */
  double dl[3], dli[3], visc;
  double *dzc, *dzf, *zc, *zf, *dzci, *dzfi, *gvr_c, *gvr_f;
  char cbcvel[18];                  /* after initbc's wall-model rewrite */
  int is_bound[6], index_wm[6];
  obound bcu, bcv, bcw, bcp, bcs, bcuf, bcvf, bcwf, bcu_mag, bcv_mag, bcw_mag;
  double *rhsbp[3];                 /* (na,nb,0:1) no halo */
  /* Poisson solver operands (src/initsolver.f90) */
  double *lambdaxy, *a, *b, *c; double normfft;
  int kind_fwd[2], kind_bwd[2];
  /* z-implicit Helmholtz operands per velocity component */
  double *av[3], *bv[3], *cv[3];
  /* rk buffers (src/rk.f90:36-41) */
  double *dudtrk[3], *dudtrko[3], *dudtrkd[3];
  /* sgs scratch (src/sgs.f90:51-53) */
  double *s0, *uc, *vc, *wc, *uf, *vf, *wf, *alph2, *wk[6], *sij[6], *mij[6];
  double is_wall[6];
  int sgs_first;
  int nthreads;
  int team_sums;                    /* o_set_team_sums: sums over all cells by planes (timing runs); 0 = the reference's cell-by-cell order */
};

#define IX(i,j,k) ((size_t)(i) + s1*((size_t)(j) + s2*(size_t)(k)))
#define CBV(side,dir,vel) (s->cbcvel[(side) + 2*((dir)-1) + 6*((vel)-1)])
#define CBP(side,dir) (s->P.cbcpre[(side) + 2*((dir)-1)])
#define CBS(side,dir) (s->P.cbcsgs[(side) + 2*((dir)-1)])
#define LWM(side,dir) (s->P.lwm[(side) + 2*((dir)-1)])
#define ISB(side,dir) (s->is_bound[(side) + 2*((dir)-1)])
#define IWM(side,dir) (s->index_wm[(side) + 2*((dir)-1)])

static double *dalloc(size_t n) { double *p = (double *)calloc(n ? n : 1, sizeof(double)); return p; }

/* ------------------------------------------------------------------ grid: src/initgrid.f90:15-197 */
static double gridpoint(int gtype, int kg, int nzg, double alpha, double z0) {
  double z = z0;
  switch (gtype) {
  default:
  case 1: /* two-end tanh, initgrid.f90:88-101 */
    if (alpha != 0.) z = 0.5*(1. + tanh((z0 - 0.5)*alpha)/tanh(alpha/2.));
    break;
  case 2: /* one end, :102-115 */
    if (alpha != 0.) z = 1.0*(1. + tanh((z0 - 1.0)*alpha)/tanh(alpha/1.));
    break;
  case 3: /* one end reversed, :116-129 */
    if (alpha != 0.) z = 1. - 1.0*(1. + tanh((1. - z0 - 1.0)*alpha)/tanh(alpha/1.));
    break;
  case 4: /* middle, :130-149 */
    if (alpha != 0.) {
      if (z0 <= 0.5) z = 0.5*(1. - 1. + tanh(2.*alpha*(z0 - 0.))/tanh(alpha));
      else           z = 0.5*(1. + 1. + tanh(2.*alpha*(z0 - 1.))/tanh(alpha));
    }
    break;
  case 6: { /* wall-model sine, :150-162; `0.1*32./nzg` is single precision in the reference */
    float dzcf = 0.1f*32.f/(float)nzg;
    double dzc = (double)dzcf;
    z = z0 - (dzc*nzg/2. - 1.)/(2.*PI)*sin(2.*PI*z0);
    break; }
  case 5: { /* 'natural' (Pirozzoli & Orlandi), :163-196 */
    const double kb = 32., alph = PI/1.5, c_eta = 0.8, dyp = 0.05;
    double nn = nzg/2.;
    double r1 = nn/kb;
    double retau = 1./(1. + r1*r1)*(dyp*nn + pow(3./4.*alph*c_eta*nn, 4./3.)*(r1*r1));
    int km = kg < (nzg - kg) ? kg : (nzg - kg);
    double k = 1.*km;
    double r2 = k/kb;
    z = 1./(1. + r2*r2)*(dyp*k + pow(3./4.*alph*c_eta*k, 4./3.)*(r2*r2))/(2.*retau);
    if (kg > nzg - kg) z = 1. - z;
    break; }
  }
  return z;
}

void o_initgrid(int gtype, int n, double gr, double lz, double *dzc, double *dzf, double *zc, double *zf) {
  /* src/initgrid.f90:15-81 */
  zf[0] = 0.;
  for (int k = 1; k <= n; k++) {
    float z0f = ((float)k - 0.f)/(1.f*(float)n);      /* `(k-0.)/(1.*n)` is default real */
    double z0 = (double)z0f;
    zf[k] = gridpoint(gtype, k, n, gr, z0);
    zf[k] = zf[k]*lz;
  }
  for (int k = 1; k <= n; k++) dzf[k] = zf[k] - zf[k-1];
  dzf[0] = dzf[1]; dzf[n+1] = dzf[n];
  for (int k = 0; k <= n; k++) dzc[k] = .5*(dzf[k] + dzf[k+1]);
  dzc[n+1] = dzc[n];
  zc[0] = -dzc[0]/2.; zf[0] = 0.;
  for (int k = 1; k <= n+1; k++) { zc[k] = zc[k-1] + dzc[k-1]; zf[k] = zf[k-1] + dzf[k]; }
}

/* ------------------------------------------------------------------ BC set-up: src/bound.f90:726-867 */
static void balloc(obound *b, const int *n) {
  b->x = dalloc((size_t)(n[1]+2)*(n[2]+2)*2);
  b->y = dalloc((size_t)(n[0]+2)*(n[2]+2)*2);
  b->z = dalloc((size_t)(n[0]+2)*(n[1]+2)*2);
}
static void bfree(obound *b) { free(b->x); free(b->y); free(b->z); }
static void bfill(obound *b, const int *n, const double *v6 /* (side,dir) */) {
  size_t nx = (size_t)(n[1]+2)*(n[2]+2), ny = (size_t)(n[0]+2)*(n[2]+2), nz = (size_t)(n[0]+2)*(n[1]+2);
  for (int sd = 0; sd < 2; sd++) {
    for (size_t q = 0; q < nx; q++) b->x[q + sd*nx] = v6[sd + 0];
    for (size_t q = 0; q < ny; q++) b->y[q + sd*ny] = v6[sd + 2];
    for (size_t q = 0; q < nz; q++) b->z[q + sd*nz] = v6[sd + 4];
  }
}
static void bnd_copy(obound *d, const obound *s_, const int *n) {
  memcpy(d->x, s_->x, sizeof(double)*(size_t)(n[1]+2)*(n[2]+2)*2);
  memcpy(d->y, s_->y, sizeof(double)*(size_t)(n[0]+2)*(n[2]+2)*2);
  memcpy(d->z, s_->z, sizeof(double)*(size_t)(n[0]+2)*(n[1]+2)*2);
}

static void initbc(ostate *s) {
  const int *n = s->n;
  memcpy(s->cbcvel, s->P.cbcvel, 18);
  for (int idir = 1; idir <= 3; idir++)           /* bound.f90:746-758 */
    for (int i = 0; i <= 1; i++)
      if (LWM(i,idir) != 0)
        for (int ivel = 1; ivel <= 3; ivel++) CBV(i,idir,ivel) = (ivel == idir) ? 'D' : 'N';
  bfill(&s->bcu, n, &s->P.bcvel[0]); bfill(&s->bcv, n, &s->P.bcvel[6]); bfill(&s->bcw, n, &s->P.bcvel[12]);
  bfill(&s->bcp, n, s->P.bcpre); bfill(&s->bcs, n, s->P.bcsgs);
  bnd_copy(&s->bcu_mag, &s->bcu, n); bnd_copy(&s->bcv_mag, &s->bcv, n); bnd_copy(&s->bcw_mag, &s->bcw, n);
  bnd_copy(&s->bcuf, &s->bcu, n); bnd_copy(&s->bcvf, &s->bcv, n); bnd_copy(&s->bcwf, &s->bcw, n);
  double h = s->P.hwm; const double *dl = s->dl, *zc = s->zc; double l3 = s->P.l[2];
  for (int q = 0; q < 6; q++) s->index_wm[q] = 0;
  if (ISB(0,1) && LWM(0,1) != 0) { int i = 1; while ((i - 0.5)*dl[0] < h) i++; IWM(0,1) = i; }
  if (ISB(1,1) && LWM(1,1) != 0) { int i = n[0]; while ((n[0] - i + 0.5)*dl[0] < h) i--; IWM(1,1) = i; }
  if (ISB(0,2) && LWM(0,2) != 0) { int j = 1; while ((j - 0.5)*dl[1] < h) j++; IWM(0,2) = j; }
  if (ISB(1,2) && LWM(1,2) != 0) { int j = n[1]; while ((n[1] - j + 0.5)*dl[1] < h) j--; IWM(1,2) = j; }
  if (ISB(0,3) && LWM(0,3) != 0) { int k = 1; while (zc[k] < h) k++; IWM(0,3) = k; }
  if (ISB(1,3) && LWM(1,3) != 0) { int k = n[2]; while (l3 - zc[k] < h) k--; IWM(1,3) = k; }
}

/* boundary r.h.s.: src/bound.f90:447-560 */
static void bc_rhs(const char *cbc2, const double *bc, int na, int nb, const double *dlc, const double *dlf,
                   char c_or_f, double *rhs) {
  size_t pl = (size_t)(na+2)*(nb+2), rl = (size_t)na*nb;
  for (int ib = 0; ib <= 1; ib++) {
    double sgn = ib == 0 ? 1. : -1.;
    for (int b_ = 1; b_ <= nb; b_++) for (int a_ = 1; a_ <= na; a_++) {
      double bcv = bc[a_ + (size_t)(na+2)*b_ + ib*pl], r = 0.;
      if (c_or_f == 'c') {
        if (cbc2[ib] == 'D') r = -2.*bcv/dlc[ib]/dlf[ib];
        else if (cbc2[ib] == 'N') r = sgn*bcv/dlf[ib];
      } else {
        if (cbc2[ib] == 'D') r = -bcv/dlc[ib]/dlf[ib];
        else if (cbc2[ib] == 'N') r = sgn*bcv/dlc[ib];
      }
      rhs[(a_-1) + (size_t)na*(b_-1) + ib*rl] = r;
    }
  }
}
static void cmpt_rhs_b(ostate *s, const char *cbc6, const obound *bc, const char *cf, double *rx, double *ry, double *rz) {
  const int *n = s->n; int n3 = n[2];
  double dx01[2] = {s->dl[0], s->dl[0]}, dy01[2] = {s->dl[1], s->dl[1]};
  double dzc01_c[2] = {s->dzc[0], s->dzc[n3]}, dzf01_c[2] = {s->dzf[1], s->dzf[n3]};
  double dzc01_f[2] = {s->dzc[1], s->dzc[n3-1]}, dzf01_f[2] = {s->dzf[1], s->dzf[n3]};
  if (rx) bc_rhs(cbc6 + 0, bc->x, n[1], n[2], dx01, dx01, cf[0], rx);
  if (ry) bc_rhs(cbc6 + 2, bc->y, n[0], n[2], dy01, dy01, cf[1], ry);
  if (rz) { if (cf[2] == 'c') bc_rhs(cbc6 + 4, bc->z, n[0], n[1], dzc01_c, dzf01_c, 'c', rz);
            else              bc_rhs(cbc6 + 4, bc->z, n[0], n[1], dzc01_f, dzf01_f, 'f', rz); }
}
static void updt_rhs_b(ostate *s, const char *cf, const char *cbc6, const double *rx, const double *ry,
                       const double *rz, double *p) {          /* bound.f90:562-617 */
  const int *n = s->n; size_t s1 = s->s1, s2 = s->s2; int q[3] = {0,0,0};
  for (int d = 0; d < 3; d++) if (cf[d] == 'f' && cbc6[1 + 2*d] == 'D') q[d] = 1;
  if (rx) for (int ib = 0; ib <= 1; ib++) if (ISB(ib,1)) { int ii = ib ? n[0]-q[0] : 1;
    for (int k = 1; k <= n[2]; k++) for (int j = 1; j <= n[1]; j++)
      p[IX(ii,j,k)] += rx[(j-1) + (size_t)n[1]*(k-1) + ib*(size_t)n[1]*n[2]]; }
  if (ry) for (int ib = 0; ib <= 1; ib++) if (ISB(ib,2)) { int jj = ib ? n[1]-q[1] : 1;
    for (int k = 1; k <= n[2]; k++) for (int i = 1; i <= n[0]; i++)
      p[IX(i,jj,k)] += ry[(i-1) + (size_t)n[0]*(k-1) + ib*(size_t)n[0]*n[2]]; }
  if (rz) for (int ib = 0; ib <= 1; ib++) if (ISB(ib,3)) { int kk = ib ? n[2]-q[2] : 1;
    for (int j = 1; j <= n[1]; j++) for (int i = 1; i <= n[0]; i++)
      p[IX(i,j,kk)] += rz[(i-1) + (size_t)n[0]*(j-1) + ib*(size_t)n[0]*n[1]]; }
}
void o_updt_rhs_b_p(ostate *s, double *pp) { updt_rhs_b(s, "ccc", s->P.cbcpre, s->rhsbp[0], s->rhsbp[1], s->rhsbp[2], pp); }
void o_updt_rhs_b_velz(ostate *s, int ivel, double alpha, double *q) {   /* main.f90:425-433 (z part) */
  const int *n = s->n; char cf[3] = {'c','c','c'}; cf[ivel-1] = 'f';
  size_t nz = (size_t)n[0]*n[1]*2; double *rz = dalloc(nz);
  const obound *bc = ivel == 1 ? &s->bcu : ivel == 2 ? &s->bcv : &s->bcw;
  cmpt_rhs_b(s, &s->cbcvel[6*(ivel-1)], bc, cf, NULL, NULL, rz);
  for (size_t i = 0; i < nz; i++) rz[i] = rz[i]*alpha;
  updt_rhs_b(s, cf, &s->cbcvel[6*(ivel-1)], NULL, NULL, rz, q);
  free(rz);
}

/* ------------------------------------------------------------------ solver set-up: src/initsolver.f90:66-169, src/fft.f90:192-245 */
static void eigenvalues(int n, const char *cbc2, char c_or_f, double *lambda) {
  double pi = PI;
  if (cbc2[0] == 'P' && cbc2[1] == 'P') {
    for (int l = 1; l <= n; l++) lambda[l-1] = -2.*(1. - cos((2*(l-1))*pi/(1.*n)));
  } else if (cbc2[0] == 'N' && cbc2[1] == 'N') {
    for (int l = 1; l <= n; l++) lambda[l-1] = -2.*(1. - cos((l-1)*pi/(1.*n)));   /* 'c' and 'f' (n-1+1) coincide */
  } else if (cbc2[0] == 'D' && cbc2[1] == 'D') {
    if (c_or_f == 'c') for (int l = 1; l <= n; l++) lambda[l-1] = -2.*(1. - cos(l*pi/(1.*n)));
    else { for (int l = 1; l <= n-1; l++) lambda[l-1] = -2.*(1. - cos(l*pi/(1.*(n+1-1)))); lambda[n-1] = 0.; }
  } else {
    for (int l = 1; l <= n; l++) lambda[l-1] = -2.*(1. - cos((2*l-1)*pi/(2.*n)));
  }
}
static void tridmatrix(const char *cbc2, int n, const double *dzci, const double *dzfi, char c_or_f,
                       double *a, double *b, double *c) {
  for (int k = 1; k <= n; k++) {
    if (c_or_f == 'c') { a[k-1] = dzfi[k]*dzci[k-1]; c[k-1] = dzfi[k]*dzci[k]; }
    else               { a[k-1] = dzfi[k]*dzci[k];   c[k-1] = dzfi[k+1]*dzci[k]; }
  }
  for (int k = 0; k < n; k++) b[k] = -(a[k] + c[k]);
  double factor[2];
  for (int ib = 0; ib <= 1; ib++) factor[ib] = cbc2[ib] == 'P' ? 0. : cbc2[ib] == 'D' ? -1. : 1.;
  if (c_or_f == 'c') { b[0] = b[0] + factor[0]*a[0]; b[n-1] = b[n-1] + factor[1]*c[n-1]; }
  else { if (cbc2[0] == 'N') b[0] = b[0] + factor[0]*a[0]; if (cbc2[1] == 'N') b[n-1] = b[n-1] + factor[1]*c[n-1]; }
}
static void find_fft(const char *bc2, char c_or_f, int *kf, int *kb, double *norm) {
  norm[0] = 2.; norm[1] = 0.;
  int pp = bc2[0]=='P', nn = bc2[0]=='N'&&bc2[1]=='N', dd = bc2[0]=='D'&&bc2[1]=='D', nd = bc2[0]=='N'&&bc2[1]=='D';
  if (pp) { *kf = O_R2HC; *kb = O_HC2R; norm[0] = 1.; return; }
  if (c_or_f == 'c') {
    if (nn) { *kf = O_REDFT10; *kb = O_REDFT01; }
    else if (dd) { *kf = O_RODFT10; *kb = O_RODFT01; }
    else if (nd) { *kf = O_REDFT11; *kb = O_REDFT11; }
    else { *kf = O_RODFT11; *kb = O_RODFT11; }
  } else {
    if (nn) { *kf = O_REDFT00; *kb = O_REDFT00; norm[1] = -1.; }
    else if (dd) { *kf = O_RODFT00; *kb = O_RODFT00; norm[1] = 1.; }
    else if (nd) { *kf = O_REDFT10; *kb = O_REDFT01; }
    else { *kf = O_RODFT01; *kb = O_RODFT10; }
  }
}

/* ------------------------------------------------------------------ r2r transforms (FFTW manual, "What FFTW Really Computes") */
typedef struct { double re, im; } cpx;
static void fft_rec(int n, const cpx *in, int istride, cpx *out, const cpx *tw, int tstride) {
  /* decimation in time, mixed radix; tw[q*tstride] = exp(-2 pi i q / n) */
  if (n == 1) { out[0] = in[0]; return; }
  int r = (n % 4 == 0) ? 4 : (n % 2 == 0) ? 2 : (n % 3 == 0) ? 3 : (n % 5 == 0) ? 5 : 0;
  if (r == 0) { for (r = 7; r*r <= n; r += 2) if (n % r == 0) break; if (r*r > n) r = n; }
  int m = n/r;
  for (int q = 0; q < r; q++) fft_rec(m, in + (size_t)q*istride, istride*r, out + (size_t)q*m, tw, tstride*r);
  cpx tstack[64]; cpx *t = r <= 64 ? tstack : (cpx *)malloc(sizeof(cpx)*r);      /* (no allocator call per recursion level for the usual radices) */
  for (int k = 0; k < m; k++) {
    for (int q = 0; q < r; q++) {
      cpx x = out[(size_t)q*m + k], w = tw[(size_t)((long)q*k)*tstride];      /* q k < r m = n: no wrap */
      t[q].re = x.re*w.re - x.im*w.im; t[q].im = x.re*w.im + x.im*w.re;
    }
    for (int p = 0; p < r; p++) {
      double sr = 0., si = 0.;
      for (int q = 0; q < r; q++) {
        cpx w = tw[(size_t)((long)((p*q) % r)*m)*tstride];      /* (p q m) mod n = ((p q) mod r) m */
        sr += t[q].re*w.re - t[q].im*w.im; si += t[q].re*w.im + t[q].im*w.re;
      }
      out[(size_t)p*m + k].re = sr; out[(size_t)p*m + k].im = si;
    }
  }
  if (t != tstack) free(t);
}
typedef struct { int n; cpx *tw; } twtab;
static twtab g_tw[16]; static volatile int g_ntw = 0;
static const cpx *twiddles(int n) {
  /* read-mostly table: the common case (the size is there) takes no lock -- a critical section around every line's look-up serialised the
     transforms of all threads (the team of 64 ran slower than the team of 32). Entries are published by the counter, after they are complete. */
  const int have = g_ntw;
  for (int i = 0; i < have; i++) if (g_tw[i].n == n) return g_tw[i].tw;
  const cpx *res = NULL;
  #pragma omp critical(o_twid)
  {
    for (int i = 0; i < g_ntw; i++) if (g_tw[i].n == n) res = g_tw[i].tw;
    if (!res) {
      cpx *t = (cpx *)malloc(sizeof(cpx)*n);
      for (int q = 0; q < n; q++) { double a = -2.*PI*q/n; t[q].re = cos(a); t[q].im = sin(a); }
      if (g_ntw < 16) { g_tw[g_ntw].n = n; g_tw[g_ntw].tw = t;
        #pragma omp flush
        g_ntw = g_ntw + 1; }
      res = t;
    }
  }
  return res;
}
/* scratch of the calling thread, grown on demand (three allocator calls per line -- 1.6e6 lines per solve at 512^3 -- were the other thing besides
   the twiddle look-up that every thread of a large team queued for) */
static void *scratch(int slot, size_t bytes) {
  static __thread void *buf[4]; static __thread size_t cap[4];
  if (cap[slot] < bytes) { free(buf[slot]); buf[slot] = malloc(bytes); cap[slot] = bytes; }
  return buf[slot];
}
static void cfft(int n, cpx *x) { /* forward, unnormalised */
  cpx *y = (cpx *)scratch(0, sizeof(cpx)*n);
  fft_rec(n, x, 1, y, twiddles(n), 1);
  memcpy(x, y, sizeof(cpx)*n);
}
static void r2r_impl(int kind, int n, double *x, int st, int depth) {
  /* (RODFT10/01 call REDFT10/01 on their own y: the nested call takes the second pair of scratch slots) */
  double *y = (double *)scratch(1 + 2*depth, sizeof(double)*(n > 0 ? n : 1)); cpx *z; const double pi = PI;
  switch (kind) {
  case O_R2HC: /* Y_k = sum x_j e^{-2 pi i jk/n}; out r0..r_{n/2}, i_{(n+1)/2-1}..i_1 */
    z = (cpx *)scratch(2, sizeof(cpx)*n);
    for (int j = 0; j < n; j++) { z[j].re = x[(size_t)j*st]; z[j].im = 0.; }
    cfft(n, z);
    for (int k = 0; k <= n/2; k++) y[k] = z[k].re;
    for (int k = 1; k < (n+1)/2; k++) y[n-k] = z[k].im;
    break;
  case O_HC2R: /* inverse of the above, unnormalised: x_j = sum_k Y_k e^{+2 pi i jk/n} */
    z = (cpx *)scratch(2, sizeof(cpx)*n);
    for (int k = 0; k <= n/2; k++) { z[k].re = x[(size_t)k*st]; z[k].im = 0.; }
    for (int k = 1; k < (n+1)/2; k++) { z[k].im = x[(size_t)(n-k)*st]; z[n-k].re = z[k].re; z[n-k].im = -z[k].im; }
    for (int k = 0; k < n; k++) z[k].im = -z[k].im;       /* conj -> forward FFT -> conj */
    cfft(n, z);
    for (int j = 0; j < n; j++) y[j] = z[j].re;
    break;
  case O_REDFT10: /* DCT-II: Y_k = 2 sum x_j cos(pi (j+1/2) k / n), via Makhoul's length-n FFT */
    z = (cpx *)scratch(2, sizeof(cpx)*n);
    for (int j = 0; j < (n+1)/2; j++) { z[j].re = x[(size_t)(2*j)*st]; z[j].im = 0.; }
    for (int j = 0; j < n/2; j++) { z[n-1-j].re = x[(size_t)(2*j+1)*st]; z[n-1-j].im = 0.; }
    cfft(n, z);
    for (int k = 0; k < n; k++) { double a = -pi*k/(2.*n); y[k] = 2.*(z[k].re*cos(a) - z[k].im*sin(a)); }
    break;
  case O_REDFT01: /* DCT-III: Y_k = x_0 + 2 sum_{j>=1} x_j cos(pi j (k+1/2)/n) */
    z = (cpx *)scratch(2, sizeof(cpx)*n);
    for (int k = 0; k < n; k++) { /* V_k = (x_k - i x_{n-k}) e^{+i pi k/2n}, x_n := 0; then inverse FFT */
      double xr = x[(size_t)k*st], xi = k == 0 ? 0. : -x[(size_t)(n-k)*st], a = pi*k/(2.*n);
      z[k].re = xr*cos(a) - xi*sin(a); z[k].im = -(xr*sin(a) + xi*cos(a)); /* conj for inverse via forward */
    }
    cfft(n, z);
    for (int j = 0; j < (n+1)/2; j++) y[2*j] = z[j].re;
    for (int j = 0; j < n/2; j++) y[2*j+1] = z[n-1-j].re;
    break;
  case O_RODFT10: /* DST-II: Y_k = 2 sum x_j sin(pi (j+1/2)(k+1)/n) = DCT-II of (-1)^j x_j, reversed */
    for (int j = 0; j < n; j++) y[j] = (j & 1) ? -x[(size_t)j*st] : x[(size_t)j*st];
    r2r_impl(O_REDFT10, n, y, 1, 1);
    for (int k = 0; k < n/2; k++) { double t = y[k]; y[k] = y[n-1-k]; y[n-1-k] = t; }
    break;
  case O_RODFT01: /* DST-III: Y_k = (-1)^k x_{n-1} + 2 sum_{j<n-1} x_j sin(pi (j+1)(k+1/2)/n) */
    for (int j = 0; j < n; j++) y[j] = x[(size_t)(n-1-j)*st];
    r2r_impl(O_REDFT01, n, y, 1, 1);
    for (int k = 0; k < n; k++) if (k & 1) y[k] = -y[k];
    break;
  case O_REDFT11: for (int k = 0; k < n; k++) { double a = 0.;
      for (int j = 0; j < n; j++) a += x[(size_t)j*st]*cos(pi*(j+0.5)*(k+0.5)/n); y[k] = 2.*a; } break;
  case O_RODFT11: for (int k = 0; k < n; k++) { double a = 0.;
      for (int j = 0; j < n; j++) a += x[(size_t)j*st]*sin(pi*(j+0.5)*(k+0.5)/n); y[k] = 2.*a; } break;
  case O_REDFT00: for (int k = 0; k < n; k++) { double a = x[0] + ((k & 1) ? -1. : 1.)*x[(size_t)(n-1)*st];
      for (int j = 1; j < n-1; j++) a += 2.*x[(size_t)j*st]*cos(pi*j*k/(n-1.)); y[k] = a; } break;
  case O_RODFT00: for (int k = 0; k < n; k++) { double a = 0.;
      for (int j = 0; j < n; j++) a += x[(size_t)j*st]*sin(pi*(j+1.)*(k+1.)/(n+1.)); y[k] = 2.*a; } break;
  default: for (int k = 0; k < n; k++) y[k] = x[(size_t)k*st];
  }
  for (int k = 0; k < n; k++) x[(size_t)k*st] = y[k];
}
void o_r2r(int kind, int n, double *x, int st) { r2r_impl(kind, n, x, st, 0); }

/* ------------------------------------------------------------------ create / destroy */
ostate *o_create(const oparams *p) {
  ostate *s = (ostate *)calloc(1, sizeof(ostate));
  s->P = *p;
  for (int d = 0; d < 3; d++) {
    s->n[d] = p->ng[d];
    s->dl[d] = p->l[d]/(1.f*(float)p->ng[d]);     /* param.f90:153 `l(:)/(1.*ng(:))` */
    s->dli[d] = 1./s->dl[d];                      /* dl**(-1) */
  }
  s->visc = 1./p->visci;
  const int *n = s->n;
  s->s1 = n[0]+2; s->s2 = n[1]+2; s->ntot = s->s1*s->s2*(size_t)(n[2]+2);
  s->nthreads = p->nthreads > 0 ? p->nthreads : 1;
  int n3 = n[2];
  s->dzc = dalloc(n3+2); s->dzf = dalloc(n3+2); s->zc = dalloc(n3+2); s->zf = dalloc(n3+2);
  s->dzci = dalloc(n3+2); s->dzfi = dalloc(n3+2); s->gvr_c = dalloc(n3+2); s->gvr_f = dalloc(n3+2);
  o_initgrid(p->gtype, n3, p->gr, p->l[2], s->dzc, s->dzf, s->zc, s->zf);
  for (int k = 0; k <= n3+1; k++) {               /* main.f90:273-274,282-283 */
    s->dzci[k] = 1./s->dzc[k]; s->dzfi[k] = 1./s->dzf[k];
    s->gvr_c[k] = s->dl[0]*s->dl[1]*s->dzc[k]/(p->l[0]*p->l[1]*p->l[2]);
    s->gvr_f[k] = s->dl[0]*s->dl[1]*s->dzf[k]/(p->l[0]*p->l[1]*p->l[2]);
  }
  /* one rank, x-pencils: is_bound true along x and along non-periodic y,z (initmpi.f90:201-204) */
  s->is_bound[0] = s->is_bound[1] = 1;
  for (int d = 2; d <= 3; d++) { int per = CBP(0,d) == 'P' && CBP(1,d) == 'P'; ISB(0,d) = ISB(1,d) = !per; }
  balloc(&s->bcu, n); balloc(&s->bcv, n); balloc(&s->bcw, n); balloc(&s->bcp, n); balloc(&s->bcs, n);
  balloc(&s->bcuf, n); balloc(&s->bcvf, n); balloc(&s->bcwf, n);
  balloc(&s->bcu_mag, n); balloc(&s->bcv_mag, n); balloc(&s->bcw_mag, n);
  initbc(s);
  s->rhsbp[0] = dalloc((size_t)n[1]*n[2]*2); s->rhsbp[1] = dalloc((size_t)n[0]*n[2]*2); s->rhsbp[2] = dalloc((size_t)n[0]*n[1]*2);
  cmpt_rhs_b(s, p->cbcpre, &s->bcp, "ccc", s->rhsbp[0], s->rhsbp[1], s->rhsbp[2]);
  /* Poisson solver: initsolver.f90:17-64 */
  double *lx = dalloc(n[0]), *ly = dalloc(n[1]);
  eigenvalues(n[0], &p->cbcpre[0], 'c', lx); for (int i = 0; i < n[0]; i++) lx[i] = lx[i]*(s->dli[0]*s->dli[0]);
  eigenvalues(n[1], &p->cbcpre[2], 'c', ly); for (int j = 0; j < n[1]; j++) ly[j] = ly[j]*(s->dli[1]*s->dli[1]);
  s->lambdaxy = dalloc((size_t)n[0]*n[1]);
  for (int j = 0; j < n[1]; j++) for (int i = 0; i < n[0]; i++) s->lambdaxy[i + (size_t)n[0]*j] = lx[i] + ly[j];
  free(lx); free(ly);
  s->a = dalloc(n3); s->b = dalloc(n3); s->c = dalloc(n3);
  tridmatrix(&p->cbcpre[4], n3, s->dzci, s->dzfi, 'c', s->a, s->b, s->c);
  double norm[2], nf = 1.;
  find_fft(&p->cbcpre[0], 'c', &s->kind_fwd[0], &s->kind_bwd[0], norm); nf = nf*norm[0]*(n[0] + norm[1]);
  find_fft(&p->cbcpre[2], 'c', &s->kind_fwd[1], &s->kind_bwd[1], norm); nf = nf*norm[0]*(n[1] + norm[1]);
  s->normfft = 1./nf;
  if (p->impdiff) {  /* main.f90:322-327: a,b,c per component; cbcvel AFTER initbc's rewrite */
    for (int iv = 0; iv < 3; iv++) {
      s->av[iv] = dalloc(n3); s->bv[iv] = dalloc(n3); s->cv[iv] = dalloc(n3);
      tridmatrix(&s->cbcvel[6*iv + 4], n3, s->dzci, s->dzfi, iv == 2 ? 'f' : 'c', s->av[iv], s->bv[iv], s->cv[iv]);
    }
  }
  size_t ni = (size_t)n[0]*n[1]*n[2];
  for (int c = 0; c < 3; c++) { s->dudtrk[c] = dalloc(ni); s->dudtrko[c] = dalloc(ni); if (p->impdiff) s->dudtrkd[c] = dalloc(ni); }
  s->sgs_first = 1;
  return s;
}
void o_destroy(ostate *s) {
  if (!s) return;
  free(s->dzc); free(s->dzf); free(s->zc); free(s->zf); free(s->dzci); free(s->dzfi); free(s->gvr_c); free(s->gvr_f);
  bfree(&s->bcu); bfree(&s->bcv); bfree(&s->bcw); bfree(&s->bcp); bfree(&s->bcs); bfree(&s->bcuf); bfree(&s->bcvf);
  bfree(&s->bcwf); bfree(&s->bcu_mag); bfree(&s->bcv_mag); bfree(&s->bcw_mag);
  for (int d = 0; d < 3; d++) { free(s->rhsbp[d]); free(s->dudtrk[d]); free(s->dudtrko[d]); free(s->dudtrkd[d]);
    free(s->av[d]); free(s->bv[d]); free(s->cv[d]); }
  free(s->lambdaxy); free(s->a); free(s->b); free(s->c);
  free(s->s0); free(s->uc); free(s->vc); free(s->wc); free(s->uf); free(s->vf); free(s->wf); free(s->alph2);
  for (int m = 0; m < 6; m++) { free(s->wk[m]); free(s->sij[m]); free(s->mij[m]); }
  free(s);
}
/* first touch of a haloed field by the threads that will work on its planes (timing aid of bench.py's cpu_baseline: a field allocated by the
   caller and first written by one thread lives on that thread's NUMA node; the loops below are split over (k, j) like this one). No arithmetic. */
void o_set_team_sums(ostate *s, int on) { s->team_sums = on != 0; }
void o_first_touch(const ostate *s, double *a) {
  const int *n = s->n; const size_t s1 = s->s1, s2 = s->s2;
  #pragma omp parallel for collapse(2) schedule(static) num_threads(s->nthreads)
  for (int k = 0; k <= n[2] + 1; k++) for (int j = 0; j <= n[1] + 1; j++) memset(&a[(size_t)0 + s1*((size_t)j + s2*(size_t)k)], 0, sizeof(double)*s1);
}
void o_get_grid(const ostate *s, double *dzc, double *dzf, double *zc, double *zf) {
  size_t b = sizeof(double)*(s->n[2]+2);
  memcpy(dzc, s->dzc, b); memcpy(dzf, s->dzf, b); memcpy(zc, s->zc, b); memcpy(zf, s->zf, b);
}
void o_get_index_wm(const ostate *s, int *iw) { memcpy(iw, s->index_wm, sizeof(int)*6); }
void o_get_cbcvel(const ostate *s, char *c) { memcpy(c, s->cbcvel, 18); }
void o_get_rhsbp(const ostate *s, double *x, double *y, double *z) {
  const int *n = s->n;
  memcpy(x, s->rhsbp[0], sizeof(double)*(size_t)n[1]*n[2]*2);
  memcpy(y, s->rhsbp[1], sizeof(double)*(size_t)n[0]*n[2]*2);
  memcpy(z, s->rhsbp[2], sizeof(double)*(size_t)n[0]*n[1]*2);
}
void o_get_bcvel(const ostate *s, int ivel, double *x, double *y, double *z) {
  const obound *b = ivel == 1 ? &s->bcu : ivel == 2 ? &s->bcv : &s->bcw; const int *n = s->n;
  memcpy(x, b->x, sizeof(double)*(size_t)(n[1]+2)*(n[2]+2)*2);
  memcpy(y, b->y, sizeof(double)*(size_t)(n[0]+2)*(n[2]+2)*2);
  memcpy(z, b->z, sizeof(double)*(size_t)(n[0]+2)*(n[1]+2)*2);
}
void o_get_solver(const ostate *s, int which, double *lambdaxy, double *a, double *b, double *c, double *normfft) {
  const int *n = s->n; size_t nb = sizeof(double)*n[2];
  if (which == 0) {
    memcpy(lambdaxy, s->lambdaxy, sizeof(double)*(size_t)n[0]*n[1]);
    memcpy(a, s->a, nb); memcpy(b, s->b, nb); memcpy(c, s->c, nb); *normfft = s->normfft;
  } else if (s->av[which-1]) {
    memcpy(a, s->av[which-1], nb); memcpy(b, s->bv[which-1], nb); memcpy(c, s->cv[which-1], nb);
  }
}

/* ------------------------------------------------------------------ ghost cells: src/bound.f90:202-399 */
static void set_bc(ostate *s, char ctype, int ibound, int idir, int centered, const double *bc, double dr, double *p) {
  const int *nn = s->n; size_t s1 = s->s1, s2 = s->s2;
  int n = nn[idir-1];
  /* plane extents (all points incl. ghosts of the other two directions) */
  int na = idir == 1 ? nn[1] : nn[0], nb = idir == 3 ? nn[1] : nn[2];
  size_t pl = (size_t)(na+2)*(nb+2);
  double sgn = (ctype == 'D' && centered) ? -1. : 1.;
  #define PA(m,a_,b_) (idir == 1 ? p[IX(m,a_,b_)] : idir == 2 ? p[IX(a_,m,b_)] : p[IX(a_,b_,m)])
  #define SETP(m,a_,b_,val) do { if (idir == 1) p[IX(m,a_,b_)] = (val); else if (idir == 2) p[IX(a_,m,b_)] = (val); else p[IX(a_,b_,m)] = (val); } while (0)
  #pragma omp parallel for schedule(static) num_threads(s->nthreads)
  for (int b_ = 0; b_ <= nb+1; b_++) for (int a_ = 0; a_ <= na+1; a_++) {
    double bcv = bc ? bc[a_ + (size_t)(na+2)*b_ + ibound*pl] : 0.;
    switch (ctype) {
    case 'P': { double lo_ = PA(n,a_,b_), hi_ = PA(1,a_,b_); SETP(0,a_,b_,lo_); SETP(n+1,a_,b_,hi_); break; }
    case 'D':
      if (centered) { if (ibound == 0) SETP(0,a_,b_, 2.*bcv + sgn*PA(1,a_,b_)); else SETP(n+1,a_,b_, 2.*bcv + sgn*PA(n,a_,b_)); }
      else { if (ibound == 0) SETP(0,a_,b_, bcv); else { SETP(n+1,a_,b_, PA(n-1,a_,b_)); SETP(n,a_,b_, bcv); } }
      break;
    case 'N':
      if (centered) { if (ibound == 0) SETP(0,a_,b_, -dr*bcv + sgn*PA(1,a_,b_)); else SETP(n+1,a_,b_, dr*bcv + sgn*PA(n,a_,b_)); }
      else { if (ibound == 0) SETP(0,a_,b_, -dr*bcv + PA(1,a_,b_)); else { SETP(n+1,a_,b_, PA(n,a_,b_)); SETP(n,a_,b_, dr*bcv + PA(n-1,a_,b_)); } }
      break;
    }
  }
  #undef PA
  #undef SETP
}
/* one rank: the MPI self-exchange of src/bound.f90:619-696 for periodic, non-pencil directions */
static void updthalo_self(ostate *s, int idir, double *p) {
  if (idir == 1) return;                       /* pencil axis */
  if (ISB(0,idir)) return;                     /* neighbours are MPI_PROC_NULL */
  const int *n = s->n; size_t s1 = s->s1, s2 = s->s2;
  if (idir == 2) {
    #pragma omp parallel for schedule(static) num_threads(s->nthreads)
    for (int k = 0; k <= n[2]+1; k++) for (int i = 0; i <= n[0]+1; i++) {
      p[IX(i,n[1]+1,k)] = p[IX(i,1,k)]; p[IX(i,0,k)] = p[IX(i,n[1],k)]; } }
  else {
    #pragma omp parallel for schedule(static) num_threads(s->nthreads)
    for (int j = 0; j <= n[1]+1; j++) for (int i = 0; i <= n[0]+1; i++) {
      p[IX(i,j,n[2]+1)] = p[IX(i,j,1)]; p[IX(i,j,0)] = p[IX(i,j,n[2])]; } }
}
void o_boundp(ostate *s, int which, double *p) {   /* bound.f90:156-200 */
  const char *cbc = which == 0 ? s->P.cbcpre : s->P.cbcsgs; const obound *bc = which == 0 ? &s->bcp : &s->bcs;
  const double *dl = s->dl; int n3 = s->n[2];
  for (int d = 1; d <= 3; d++) updthalo_self(s, d, p);
  if (ISB(0,1)) set_bc(s, cbc[0], 0, 1, 1, bc->x, dl[0], p);
  if (ISB(1,1)) set_bc(s, cbc[1], 1, 1, 1, bc->x, dl[0], p);
  if (ISB(0,2)) set_bc(s, cbc[2], 0, 2, 1, bc->y, dl[1], p);
  if (ISB(1,2)) set_bc(s, cbc[3], 1, 2, 1, bc->y, dl[1], p);
  if (ISB(0,3)) set_bc(s, cbc[4], 0, 3, 1, bc->z, s->dzc[0], p);
  if (ISB(1,3)) set_bc(s, cbc[5], 1, 3, 1, bc->z, s->dzc[n3], p);
}

/* ------------------------------------------------------------------ wall model: src/wmodel.f90:65-335 */
static double vel_relative(double v1, double v2, double coef, double mag) {
  double r = (1. - coef)*v1 + coef*v2; r = r - mag; return r;
}
static void wallmodel(int mtype, double uh, double vh, double h, double l1d, double visc, double *tauw) {
  double upar, utau, f, fp, conv, utau_old, tauw_tot;
  if (mtype == 1) {
    conv = 1.; upar = sqrt(uh*uh + vh*vh);
    utau = fmax(sqrt(upar/h*visc), visc/h*exp(-KAP_LOG*B_LOG));
    while (conv > 0.5e-4) {
      utau_old = utau;
      f = upar/utau - 1./KAP_LOG*log(h*utau/visc) - B_LOG;
      fp = -1./utau*(upar/utau + 1./KAP_LOG);
      utau = fabs(utau - f/fp);
      conv = fabs(utau/utau_old - 1.);
    }
    tauw_tot = utau*utau;
  } else {
    upar = sqrt(uh*uh + vh*vh);
    double del = 0.5*l1d, umax = upar/(h/del*(2. - h/del));
    tauw_tot = 2./del*umax*visc;
  }
  tauw[0] = tauw_tot*uh/(upar + EPS); tauw[1] = tauw_tot*vh/(upar + EPS);
}
static void cmpt_wallmodelbc(ostate *s, int ibound, int idir, const double *u, const double *v, const double *w,
                             obound *bcu, obound *bcv, obound *bcw) {
  const int *n = s->n; size_t s1 = s->s1, s2 = s->s2; const double *dl = s->dl, *l = s->P.l, *zc = s->zc, *zf = s->zf, *dzc = s->dzc;
  double h = s->P.hwm, visc = s->visc, visci = 1./visc, coef, sgn, tauw[2];
  int mtype = LWM(ibound,idir), index = IWM(ibound,idir);
  #define BX(b,j,k) ((b)[(j) + (size_t)(n[1]+2)*(k) + (size_t)ibound*(n[1]+2)*(n[2]+2)])
  #define BY(b,i,k) ((b)[(i) + (size_t)(n[0]+2)*(k) + (size_t)ibound*(n[0]+2)*(n[2]+2)])
  #define BZ(b,i,j) ((b)[(i) + (size_t)(n[0]+2)*(j) + (size_t)ibound*(n[0]+2)*(n[1]+2)])
  if (idir == 1) {
    int i2 = index, i1 = ibound == 0 ? index-1 : index+1;
    if (ibound == 0) { coef = (h - (i1 - 0.5)*dl[0])/dl[0]; sgn = 1.; } else { coef = (h - (n[0] - i1 + 0.5)*dl[0])/dl[0]; sgn = -1.; }
    for (int k = 1; k <= n[2]; k++) for (int j = 0; j <= n[1]; j++) {
      double v1 = v[IX(i1,j,k)], v2 = v[IX(i2,j,k)];
      double w1 = 0.25*(w[IX(i1,j,k)] + w[IX(i1,j+1,k)] + w[IX(i1,j,k-1)] + w[IX(i1,j+1,k-1)]);
      double w2 = 0.25*(w[IX(i2,j,k)] + w[IX(i2,j+1,k)] + w[IX(i2,j,k-1)] + w[IX(i2,j+1,k-1)]);
      double v_mag = BX(s->bcv_mag.x,j,k);
      double w_mag = 0.25*(BX(s->bcw_mag.x,j,k) + BX(s->bcw_mag.x,j+1,k) + BX(s->bcw_mag.x,j,k-1) + BX(s->bcw_mag.x,j+1,k-1));
      double vh = vel_relative(v1,v2,coef,v_mag), wh = vel_relative(w1,w2,coef,w_mag);
      wallmodel(mtype, vh, wh, h, l[0], visc, tauw); BX(bcv->x,j,k) = sgn*visci*tauw[0];
    }
    for (int k = 0; k <= n[2]; k++) for (int j = 1; j <= n[1]; j++) {
      double wei = (zf[k] - zc[k])/dzc[k];
      double v1 = 0.5*((1.-wei)*(v[IX(i1,j-1,k)] + v[IX(i1,j,k)]) + wei*(v[IX(i1,j-1,k+1)] + v[IX(i1,j,k+1)]));
      double v2 = 0.5*((1.-wei)*(v[IX(i2,j-1,k)] + v[IX(i2,j,k)]) + wei*(v[IX(i2,j-1,k+1)] + v[IX(i2,j,k+1)]));
      double w1 = w[IX(i1,j,k)], w2 = w[IX(i2,j,k)];
      double v_mag = 0.5*((1.-wei)*(BX(s->bcv_mag.x,j-1,k) + BX(s->bcv_mag.x,j,k)) + wei*(BX(s->bcv_mag.x,j-1,k+1) + BX(s->bcv_mag.x,j,k+1)));
      double w_mag = BX(s->bcw_mag.x,j,k);
      double vh = vel_relative(v1,v2,coef,v_mag), wh = vel_relative(w1,w2,coef,w_mag);
      wallmodel(mtype, vh, wh, h, l[0], visc, tauw); BX(bcw->x,j,k) = sgn*visci*tauw[1];
    }
  } else if (idir == 2) {
    int j2 = index, j1 = ibound == 0 ? index-1 : index+1;
    /* `(j1-0.5)` is default real in the reference (wmodel.f90:175,180): exact for these magnitudes */
    if (ibound == 0) { coef = (h - (double)((float)j1 - 0.5f)*dl[1])/dl[1]; sgn = 1.; }
    else { coef = (h - (double)((float)(n[1] - j1) + 0.5f)*dl[1])/dl[1]; sgn = -1.; }
    for (int k = 1; k <= n[2]; k++) for (int i = 0; i <= n[0]; i++) {
      double u1 = u[IX(i,j1,k)], u2 = u[IX(i,j2,k)];
      double w1 = 0.25*(w[IX(i,j1,k)] + w[IX(i+1,j1,k)] + w[IX(i,j1,k-1)] + w[IX(i+1,j1,k-1)]);
      double w2 = 0.25*(w[IX(i,j2,k)] + w[IX(i+1,j2,k)] + w[IX(i,j2,k-1)] + w[IX(i+1,j2,k-1)]);
      double u_mag = BY(s->bcu_mag.y,i,k);
      double w_mag = 0.25*(BY(s->bcw_mag.y,i,k) + BY(s->bcw_mag.y,i+1,k) + BY(s->bcw_mag.y,i,k-1) + BY(s->bcw_mag.y,i+1,k-1));
      double uh = vel_relative(u1,u2,coef,u_mag), wh = vel_relative(w1,w2,coef,w_mag);
      wallmodel(mtype, uh, wh, h, l[1], visc, tauw); BY(bcu->y,i,k) = sgn*visci*tauw[0];
    }
    for (int k = 0; k <= n[2]; k++) for (int i = 1; i <= n[0]; i++) {
      double wei = (zf[k] - zc[k])/dzc[k];
      double u1 = 0.5*((1.-wei)*(u[IX(i-1,j1,k)] + u[IX(i,j1,k)]) + wei*(u[IX(i-1,j1,k+1)] + u[IX(i,j1,k+1)]));
      double u2 = 0.5*((1.-wei)*(u[IX(i-1,j2,k)] + u[IX(i,j2,k)]) + wei*(u[IX(i-1,j2,k+1)] + u[IX(i,j2,k+1)]));
      double w1 = w[IX(i,j1,k)], w2 = w[IX(i,j2,k)];
      double u_mag = 0.5*((1.-wei)*(BY(s->bcu_mag.y,i-1,k) + BY(s->bcu_mag.y,i,k)) + wei*(BY(s->bcu_mag.y,i-1,k+1) + BY(s->bcu_mag.y,i,k+1)));
      double w_mag = BY(s->bcw_mag.y,i,k);
      double uh = vel_relative(u1,u2,coef,u_mag), wh = vel_relative(w1,w2,coef,w_mag);
      wallmodel(mtype, uh, wh, h, l[1], visc, tauw); BY(bcw->y,i,k) = sgn*visci*tauw[1];
    }
  } else {
    int k2 = index, k1 = ibound == 0 ? index-1 : index+1;
    if (ibound == 0) { coef = (h - zc[k1])/dzc[k1]; sgn = 1.; } else { coef = (h - (l[2] - zc[k1]))/(dzc[k2]); sgn = -1.; }
    for (int j = 1; j <= n[1]; j++) for (int i = 0; i <= n[0]; i++) {
      double u1 = u[IX(i,j,k1)], u2 = u[IX(i,j,k2)];
      double v1 = 0.25*(v[IX(i,j,k1)] + v[IX(i+1,j,k1)] + v[IX(i,j-1,k1)] + v[IX(i+1,j-1,k1)]);
      double v2 = 0.25*(v[IX(i,j,k2)] + v[IX(i+1,j,k2)] + v[IX(i,j-1,k2)] + v[IX(i+1,j-1,k2)]);
      double u_mag = BZ(s->bcu_mag.z,i,j);
      double v_mag = 0.25*(BZ(s->bcv_mag.z,i,j) + BZ(s->bcv_mag.z,i+1,j) + BZ(s->bcv_mag.z,i,j-1) + BZ(s->bcv_mag.z,i+1,j-1));
      double uh = vel_relative(u1,u2,coef,u_mag), vh = vel_relative(v1,v2,coef,v_mag);
      wallmodel(mtype, uh, vh, h, l[2], visc, tauw); BZ(bcu->z,i,j) = sgn*visci*tauw[0];
    }
    for (int j = 0; j <= n[1]; j++) for (int i = 1; i <= n[0]; i++) {
      double u1 = 0.25*(u[IX(i-1,j,k1)] + u[IX(i,j,k1)] + u[IX(i-1,j+1,k1)] + u[IX(i,j+1,k1)]);
      double u2 = 0.25*(u[IX(i-1,j,k2)] + u[IX(i,j,k2)] + u[IX(i-1,j+1,k2)] + u[IX(i,j+1,k2)]);
      double v1 = v[IX(i,j,k1)], v2 = v[IX(i,j,k2)];
      double u_mag = 0.25*(BZ(s->bcu_mag.z,i-1,j) + BZ(s->bcu_mag.z,i,j) + BZ(s->bcu_mag.z,i-1,j+1) + BZ(s->bcu_mag.z,i,j+1));
      double v_mag = BZ(s->bcv_mag.z,i,j);
      double uh = vel_relative(u1,u2,coef,u_mag), vh = vel_relative(v1,v2,coef,v_mag);
      wallmodel(mtype, uh, vh, h, l[2], visc, tauw); BZ(bcv->z,i,j) = sgn*visci*tauw[1];
    }
  }
  #undef BX
  #undef BY
  #undef BZ
}

static void bounduvw_bc(ostate *s, obound *bcu, obound *bcv, obound *bcw, int is_updt_wm, int is_correc,
                        double *u, double *v, double *w) {     /* bound.f90:18-154 */
  const double *dl = s->dl; int n3 = s->n[2];
  for (int d = 1; d <= 3; d++) { updthalo_self(s, d, u); updthalo_self(s, d, v); updthalo_self(s, d, w); }
  int inb;
  inb = (!is_correc) || (CBV(0,1,1) == 'P' && CBV(1,1,1) == 'P');
  for (int ib = 0; ib <= 1; ib++) if (ISB(ib,1)) {
    if (inb) set_bc(s, CBV(ib,1,1), ib, 1, 0, bcu->x, dl[0], u);
    if (LWM(ib,1) == 0) { set_bc(s, CBV(ib,1,2), ib, 1, 1, bcv->x, dl[0], v); set_bc(s, CBV(ib,1,3), ib, 1, 1, bcw->x, dl[0], w); }
  }
  inb = (!is_correc) || (CBV(0,2,2) == 'P' && CBV(1,2,2) == 'P');
  for (int ib = 0; ib <= 1; ib++) if (ISB(ib,2)) {
    if (inb) set_bc(s, CBV(ib,2,2), ib, 2, 0, bcv->y, dl[1], v);
    if (LWM(ib,2) == 0) { set_bc(s, CBV(ib,2,1), ib, 2, 1, bcu->y, dl[1], u); set_bc(s, CBV(ib,2,3), ib, 2, 1, bcw->y, dl[1], w); }
  }
  inb = (!is_correc) || (CBV(0,3,3) == 'P' && CBV(1,3,3) == 'P');
  for (int ib = 0; ib <= 1; ib++) if (ISB(ib,3)) {
    double drf = ib ? s->dzf[n3] : s->dzf[0], drc = ib ? s->dzc[n3] : s->dzc[0];
    if (inb) set_bc(s, CBV(ib,3,3), ib, 3, 0, bcw->z, drf, w);
    if (LWM(ib,3) == 0) { set_bc(s, CBV(ib,3,1), ib, 3, 1, bcu->z, drc, u); set_bc(s, CBV(ib,3,2), ib, 3, 1, bcv->z, drc, v); }
  }
  if (is_updt_wm)                                  /* wmodel.f90:19-63 */
    for (int d = 1; d <= 3; d++) for (int ib = 0; ib <= 1; ib++)
      if (ISB(ib,d) && LWM(ib,d) != 0) cmpt_wallmodelbc(s, ib, d, u, v, w, bcu, bcv, bcw);
  for (int ib = 0; ib <= 1; ib++) if (ISB(ib,1) && LWM(ib,1) != 0) {
    set_bc(s, CBV(ib,1,2), ib, 1, 1, bcv->x, dl[0], v); set_bc(s, CBV(ib,1,3), ib, 1, 1, bcw->x, dl[0], w); }
  for (int ib = 0; ib <= 1; ib++) if (ISB(ib,2) && LWM(ib,2) != 0) {
    set_bc(s, CBV(ib,2,1), ib, 2, 1, bcu->y, dl[1], u); set_bc(s, CBV(ib,2,3), ib, 2, 1, bcw->y, dl[1], w); }
  for (int ib = 0; ib <= 1; ib++) if (ISB(ib,3) && LWM(ib,3) != 0) { double drc = ib ? s->dzc[n3] : s->dzc[0];
    set_bc(s, CBV(ib,3,1), ib, 3, 1, bcu->z, drc, u); set_bc(s, CBV(ib,3,2), ib, 3, 1, bcv->z, drc, v); }
}
void o_bounduvw(ostate *s, int is_updt_wm, int is_correc, double *u, double *v, double *w) {
  bounduvw_bc(s, &s->bcu, &s->bcv, &s->bcw, is_updt_wm, is_correc, u, v, w);
}

/* ------------------------------------------------------------------ momentum r.h.s.: src/mom.f90:17-309 */
void o_mom(ostate *s, const double *u, const double *v, const double *w, const double *visct,
           double *dudt, double *dvdt, double *dwdt, double *dudtd, double *dvdtd, double *dwdtd) {
  const int nx = s->n[0], ny = s->n[1], nz = s->n[2]; const size_t s1 = s->s1, s2 = s->s2;
  const double dxi = s->dli[0], dyi = s->dli[1], visc = s->visc; const double *dzci = s->dzci, *dzfi = s->dzfi;
  const int imp = s->P.impdiff;
  #pragma omp parallel for collapse(2) num_threads(s->nthreads)
  for (int k = 1; k <= nz; k++) for (int j = 1; j <= ny; j++) for (int i = 1; i <= nx; i++) {
    #define LD(a,di,dj,dk) a[IX(i+(di),j+(dj),k+(dk))]
    double u_ccm=LD(u,0,0,-1),u_pcm=LD(u,1,0,-1),u_cpm=LD(u,0,1,-1),u_cmc=LD(u,0,-1,0),u_pmc=LD(u,1,-1,0),u_mcc=LD(u,-1,0,0),
           u_ccc=LD(u,0,0,0),u_pcc=LD(u,1,0,0),u_mpc=LD(u,-1,1,0),u_cpc=LD(u,0,1,0),u_cmp=LD(u,0,-1,1),u_mcp=LD(u,-1,0,1),u_ccp=LD(u,0,0,1);
    double v_ccm=LD(v,0,0,-1),v_pcm=LD(v,1,0,-1),v_cpm=LD(v,0,1,-1),v_cmc=LD(v,0,-1,0),v_pmc=LD(v,1,-1,0),v_mcc=LD(v,-1,0,0),
           v_ccc=LD(v,0,0,0),v_pcc=LD(v,1,0,0),v_mpc=LD(v,-1,1,0),v_cpc=LD(v,0,1,0),v_cmp=LD(v,0,-1,1),v_mcp=LD(v,-1,0,1),v_ccp=LD(v,0,0,1);
    double w_ccm=LD(w,0,0,-1),w_pcm=LD(w,1,0,-1),w_cpm=LD(w,0,1,-1),w_cmc=LD(w,0,-1,0),w_pmc=LD(w,1,-1,0),w_mcc=LD(w,-1,0,0),
           w_ccc=LD(w,0,0,0),w_pcc=LD(w,1,0,0),w_mpc=LD(w,-1,1,0),w_cpc=LD(w,0,1,0),w_cmp=LD(w,0,-1,1),w_mcp=LD(w,-1,0,1),w_ccp=LD(w,0,0,1);
    double s_ccm=LD(visct,0,0,-1),s_pcm=LD(visct,1,0,-1),s_cpm=LD(visct,0,1,-1),s_cmc=LD(visct,0,-1,0),s_pmc=LD(visct,1,-1,0),
           s_mcc=LD(visct,-1,0,0),s_ccc=LD(visct,0,0,0),s_pcc=LD(visct,1,0,0),s_mpc=LD(visct,-1,1,0),s_cpc=LD(visct,0,1,0),
           s_cmp=LD(visct,0,-1,1),s_mcp=LD(visct,-1,0,1),s_ccp=LD(visct,0,0,1),s_ppc=LD(visct,1,1,0),s_pcp=LD(visct,1,0,1),s_cpp=LD(visct,0,1,1);
    #undef LD
    (void)u_cmp; (void)v_mcp; (void)w_mpc; (void)w_cmp; (void)w_mcp; (void)u_pmc; (void)v_mpc;
    double visc_ip,visc_im,visc_jp,visc_jm,visc_kp,visc_km;
    /* x */
    visc_ip = s_pcc; visc_im = s_ccc;
    visc_jp = 0.25*(s_ccc+s_pcc+s_cpc+s_ppc); visc_jm = 0.25*(s_ccc+s_pcc+s_cmc+s_pmc);
    visc_kp = 0.25*(s_ccc+s_pcc+s_ccp+s_pcp); visc_km = 0.25*(s_ccc+s_pcc+s_ccm+s_pcm);
    double dudx_ip=(u_pcc-u_ccc)*dxi, dudx_im=(u_ccc-u_mcc)*dxi, dudy_jp=(u_cpc-u_ccc)*dyi, dudy_jm=(u_ccc-u_cmc)*dyi,
           dudz_kp=(u_ccp-u_ccc)*dzci[k], dudz_km=(u_ccc-u_ccm)*dzci[k-1];
    double dvdx_jp=(v_pcc-v_ccc)*dxi, dvdx_jm=(v_pmc-v_cmc)*dxi, dwdx_kp=(w_pcc-w_ccc)*dxi, dwdx_km=(w_pcm-w_ccm)*dxi;
    double uu_ip=0.25*(u_pcc+u_ccc)*(u_ccc+u_pcc), uu_im=0.25*(u_mcc+u_ccc)*(u_ccc+u_mcc),
           vu_jp=0.25*(v_pcc+v_ccc)*(u_ccc+u_cpc), vu_jm=0.25*(v_pmc+v_cmc)*(u_ccc+u_cmc),
           wu_kp=0.25*(w_pcc+w_ccc)*(u_ccc+u_ccp), wu_km=0.25*(w_pcm+w_ccm)*(u_ccc+u_ccm);
    double dudtd_xy_s = visc*(dudx_ip-dudx_im)*dxi + visc*(dudy_jp-dudy_jm)*dyi;
    double dudtd_z_s  = visc*(dudz_kp-dudz_km)*dzfi[k];
    double dudt_s = -(uu_ip-uu_im)*dxi - (vu_jp-vu_jm)*dyi - (wu_kp-wu_km)*dzfi[k]
                    +(visc_ip*(dudx_ip+dudx_ip)-visc_im*(dudx_im+dudx_im))*dxi +
                     (visc_jp*(dudy_jp+dvdx_jp)-visc_jm*(dudy_jm+dvdx_jm))*dyi +
                     (visc_kp*(dudz_kp+dwdx_kp)-visc_km*(dudz_km+dwdx_km))*dzfi[k];
    /* y */
    visc_ip = 0.25*(s_ccc+s_cpc+s_pcc+s_ppc); visc_im = 0.25*(s_ccc+s_cpc+s_mcc+s_mpc);
    visc_jp = s_cpc; visc_jm = s_ccc;
    visc_kp = 0.25*(s_ccc+s_cpc+s_ccp+s_cpp); visc_km = 0.25*(s_ccc+s_cpc+s_ccm+s_cpm);
    double dvdx_ip=(v_pcc-v_ccc)*dxi, dvdx_im=(v_ccc-v_mcc)*dxi, dvdy_jp=(v_cpc-v_ccc)*dyi, dvdy_jm=(v_ccc-v_cmc)*dyi,
           dvdz_kp=(v_ccp-v_ccc)*dzci[k], dvdz_km=(v_ccc-v_ccm)*dzci[k-1];
    double dudy_ip=(u_cpc-u_ccc)*dyi, dudy_im=(u_mpc-u_mcc)*dyi, dwdy_kp=(w_cpc-w_ccc)*dyi, dwdy_km=(w_cpm-w_ccm)*dyi;
    double uv_ip=0.25*(u_ccc+u_cpc)*(v_ccc+v_pcc), uv_im=0.25*(u_mcc+u_mpc)*(v_ccc+v_mcc),
           vv_jp=0.25*(v_ccc+v_cpc)*(v_ccc+v_cpc), vv_jm=0.25*(v_ccc+v_cmc)*(v_ccc+v_cmc),
           wv_kp=0.25*(w_ccc+w_cpc)*(v_ccc+v_ccp), wv_km=0.25*(w_ccm+w_cpm)*(v_ccc+v_ccm);
    double dvdtd_xy_s = visc*(dvdx_ip-dvdx_im)*dxi + visc*(dvdy_jp-dvdy_jm)*dyi;
    double dvdtd_z_s  = visc*(dvdz_kp-dvdz_km)*dzfi[k];
    double dvdt_s = -(uv_ip-uv_im)*dxi - (vv_jp-vv_jm)*dyi - (wv_kp-wv_km)*dzfi[k]
                    +(visc_ip*(dvdx_ip+dudy_ip)-visc_im*(dvdx_im+dudy_im))*dxi +
                     (visc_jp*(dvdy_jp+dvdy_jp)-visc_jm*(dvdy_jm+dvdy_jm))*dyi +
                     (visc_kp*(dvdz_kp+dwdy_kp)-visc_km*(dvdz_km+dwdy_km))*dzfi[k];
    /* z */
    visc_ip = 0.25*(s_ccc+s_ccp+s_pcc+s_pcp); visc_im = 0.25*(s_ccc+s_ccp+s_mcc+s_mcp);
    visc_jp = 0.25*(s_ccc+s_ccp+s_cpc+s_cpp); visc_jm = 0.25*(s_ccc+s_ccp+s_cmc+s_cmp);
    visc_kp = s_ccp; visc_km = s_ccc;
    double dwdx_ip=(w_pcc-w_ccc)*dxi, dwdx_im=(w_ccc-w_mcc)*dxi, dwdy_jp=(w_cpc-w_ccc)*dyi, dwdy_jm=(w_ccc-w_cmc)*dyi,
           dwdz_kp=(w_ccp-w_ccc)*dzfi[k+1], dwdz_km=(w_ccc-w_ccm)*dzfi[k];
    double dudz_ip=(u_ccp-u_ccc)*dzci[k], dudz_im=(u_mcp-u_mcc)*dzci[k], dvdz_jp=(v_ccp-v_ccc)*dzci[k], dvdz_jm=(v_cmp-v_cmc)*dzci[k];
    double uw_ip=0.25*(u_ccc+u_ccp)*(w_ccc+w_pcc), uw_im=0.25*(u_mcc+u_mcp)*(w_ccc+w_mcc),
           vw_jp=0.25*(v_ccc+v_ccp)*(w_ccc+w_cpc), vw_jm=0.25*(v_cmc+v_cmp)*(w_ccc+w_cmc),
           ww_kp=0.25*(w_ccc+w_ccp)*(w_ccc+w_ccp), ww_km=0.25*(w_ccc+w_ccm)*(w_ccc+w_ccm);
    double dwdtd_xy_s = visc*(dwdx_ip-dwdx_im)*dxi + visc*(dwdy_jp-dwdy_jm)*dyi;
    double dwdtd_z_s  = visc*(dwdz_kp-dwdz_km)*dzci[k];
    double dwdt_s = -(uw_ip-uw_im)*dxi - (vw_jp-vw_jm)*dyi - (ww_kp-ww_km)*dzci[k]
                    +(visc_ip*(dwdx_ip+dudz_ip)-visc_im*(dwdx_im+dudz_im))*dxi +
                     (visc_jp*(dwdy_jp+dvdz_jp)-visc_jm*(dwdy_jm+dvdz_jm))*dyi +
                     (visc_kp*(dwdz_kp+dwdz_kp)-visc_km*(dwdz_km+dwdz_km))*dzci[k];
    size_t q = (size_t)(i-1) + (size_t)nx*((size_t)(j-1) + (size_t)ny*(k-1));
    if (imp == 2) {
      dudt[q] = dudt_s + dudtd_xy_s; dvdt[q] = dvdt_s + dvdtd_xy_s; dwdt[q] = dwdt_s + dwdtd_xy_s;
      dudtd[q] = dudtd_z_s; dvdtd[q] = dvdtd_z_s; dwdtd[q] = dwdtd_z_s;
    } else if (imp == 1) {
      dudt[q] = dudt_s; dvdt[q] = dvdt_s; dwdt[q] = dwdt_s;
      dudtd[q] = dudtd_xy_s + dudtd_z_s; dvdtd[q] = dvdtd_xy_s + dvdtd_z_s; dwdtd[q] = dwdtd_xy_s + dwdtd_z_s;
    } else {
      dudt[q] = dudt_s + dudtd_xy_s + dudtd_z_s; dvdt[q] = dvdt_s + dvdtd_xy_s + dvdtd_z_s; dwdt[q] = dwdt_s + dwdtd_xy_s + dwdtd_z_s;
    }
  }
}

/* ------------------------------------------------------------------ bulk mean / forcing: utils.f90:16-47, mom.f90:311-335 */
double o_bulk_mean(ostate *s, int c_or_f, const double *p) {
  const int *n = s->n; size_t s1 = s->s1, s2 = s->s2; const double *g = c_or_f ? s->gvr_f : s->gvr_c;
  double mean = 0.;
  if (!s->team_sums) {      /* the reference's order on one rank (utils.f90:35-46): cell by cell -- what the golden vectors were made with */
    for (int k = 1; k <= n[2]; k++) for (int j = 1; j <= n[1]; j++) for (int i = 1; i <= n[0]; i++) mean = mean + p[IX(i,j,k)]*g[k];
    return mean;
  }
  /* o_set_team_sums (the timed CPU baseline): every plane is summed in the reference's (i, j) order by one thread and the planes are added in order --
     one value for every team size, equal to the sequential sum to a few units in the last place (as the reference's own value is across rank counts:
     local sums + MPI_ALLREDUCE) */
  double *part = (double *)malloc(sizeof(double)*(size_t)(n[2]+2));
  #pragma omp parallel for schedule(static) num_threads(s->nthreads)
  for (int k = 1; k <= n[2]; k++) { double a = 0.;
    for (int j = 1; j <= n[1]; j++) for (int i = 1; i <= n[0]; i++) a = a + p[IX(i,j,k)]*g[k];
    part[k] = a; }
  for (int k = 1; k <= n[2]; k++) mean = mean + part[k];
  free(part);
  return mean;
}
void o_bulk_forcing(ostate *s, const double *f, double *u, double *v, double *w) {
  const int *n = s->n; size_t s1 = s->s1, s2 = s->s2; double *q[3] = {u, v, w};
  for (int c = 0; c < 3; c++) if (s->P.is_forced[c]) { double ff = f[c];
    #pragma omp parallel for collapse(2) num_threads(s->nthreads)
    for (int k = 1; k <= n[2]; k++) for (int j = 1; j <= n[1]; j++) for (int i = 1; i <= n[0]; i++) q[c][IX(i,j,k)] += ff; }
}

/* ------------------------------------------------------------------ RK substep: src/rk.f90:17-121,197-222 */
void o_rk(ostate *s, int irk, double dt, const double *p, const double *visct, double *u, double *v, double *w, double *f) {
  const int *n = s->n; const size_t s1 = s->s1, s2 = s->s2; const double *dli = s->dli, *dzci = s->dzci, *bforce = s->P.bforce;
  double factor1 = RKCOEFF[irk-1][0]*dt, factor2 = RKCOEFF[irk-1][1]*dt, factor12 = factor1 + factor2;
  const int imp = s->P.impdiff;
  o_mom(s, u, v, w, visct, s->dudtrk[0], s->dudtrk[1], s->dudtrk[2], s->dudtrkd[0], s->dudtrkd[1], s->dudtrkd[2]);
  const double *du = s->dudtrk[0], *dv = s->dudtrk[1], *dw = s->dudtrk[2], *duo = s->dudtrko[0], *dvo = s->dudtrko[1], *dwo = s->dudtrko[2];
  const double *dud = s->dudtrkd[0], *dvd = s->dudtrkd[1], *dwd = s->dudtrkd[2];
  #pragma omp parallel for collapse(2) num_threads(s->nthreads)
  for (int k = 1; k <= n[2]; k++) for (int j = 1; j <= n[1]; j++) for (int i = 1; i <= n[0]; i++) {
    size_t q = (size_t)(i-1) + (size_t)n[0]*((size_t)(j-1) + (size_t)n[1]*(k-1)), c = IX(i,j,k);
    u[c] = u[c] + factor1*du[q] + factor2*duo[q] + factor12*(bforce[0] - dli[0]*(p[IX(i+1,j,k)] - p[c]));
    v[c] = v[c] + factor1*dv[q] + factor2*dvo[q] + factor12*(bforce[1] - dli[1]*(p[IX(i,j+1,k)] - p[c]));
    w[c] = w[c] + factor1*dw[q] + factor2*dwo[q] + factor12*(bforce[2] - dzci[k]*(p[IX(i,j,k+1)] - p[c]));
    if (imp) { u[c] = u[c] + factor12*dud[q]; v[c] = v[c] + factor12*dvd[q]; w[c] = w[c] + factor12*dwd[q]; }
  }
  for (int c = 0; c < 3; c++) { double *t = s->dudtrk[c]; s->dudtrk[c] = s->dudtrko[c]; s->dudtrko[c] = t; }
  f[0] = f[1] = f[2] = 0.;
  if (s->P.is_forced[0]) f[0] = s->P.velf[0] - o_bulk_mean(s, 1, u);
  if (s->P.is_forced[1]) f[1] = s->P.velf[1] - o_bulk_mean(s, 1, v);
  if (s->P.is_forced[2]) f[2] = s->P.velf[2] - o_bulk_mean(s, 0, w);
  if (imp) {
    #pragma omp parallel for collapse(2) num_threads(s->nthreads)
    for (int k = 1; k <= n[2]; k++) for (int j = 1; j <= n[1]; j++) for (int i = 1; i <= n[0]; i++) {
      size_t q = (size_t)(i-1) + (size_t)n[0]*((size_t)(j-1) + (size_t)n[1]*(k-1)), c = IX(i,j,k);
      u[c] = u[c] - .5*factor12*dud[q]; v[c] = v[c] - .5*factor12*dvd[q]; w[c] = w[c] - .5*factor12*dwd[q];
    }
  }
}

/* ------------------------------------------------------------------ projection pieces */
void o_fillps(ostate *s, double dti, const double *u, const double *v, const double *w, double *p) { /* fillps.f90:14-48 */
  const int *n = s->n; const size_t s1 = s->s1, s2 = s->s2; const double *dzfi = s->dzfi;
  double dtidxi = dti*s->dli[0], dtidyi = dti*s->dli[1];
  #pragma omp parallel for collapse(2) num_threads(s->nthreads)
  for (int k = 1; k <= n[2]; k++) for (int j = 1; j <= n[1]; j++) for (int i = 1; i <= n[0]; i++)
    p[IX(i,j,k)] = ((w[IX(i,j,k)] - w[IX(i,j,k-1)])*dti*dzfi[k] + (v[IX(i,j,k)] - v[IX(i,j-1,k)])*dtidyi + (u[IX(i,j,k)] - u[IX(i-1,j,k)])*dtidxi);
}
void o_correc(ostate *s, double dt, const double *p, double *u, double *v, double *w) {  /* correc.f90:14-68 */
  const int *n = s->n; const size_t s1 = s->s1, s2 = s->s2; const double *dzci = s->dzci;
  double factori = dt*s->dli[0], factorj = dt*s->dli[1];
  #pragma omp parallel for collapse(2) num_threads(s->nthreads)
  for (int k = 0; k <= n[2]+1; k++) for (int j = 0; j <= n[1]+1; j++) for (int i = 0; i <= n[0]; i++)
    u[IX(i,j,k)] = u[IX(i,j,k)] - factori*(p[IX(i+1,j,k)] - p[IX(i,j,k)]);
  #pragma omp parallel for collapse(2) num_threads(s->nthreads)
  for (int k = 0; k <= n[2]+1; k++) for (int j = 0; j <= n[1]; j++) for (int i = 0; i <= n[0]+1; i++)
    v[IX(i,j,k)] = v[IX(i,j,k)] - factorj*(p[IX(i,j+1,k)] - p[IX(i,j,k)]);
  #pragma omp parallel for collapse(2) num_threads(s->nthreads)
  for (int k = 0; k <= n[2]; k++) for (int j = 0; j <= n[1]+1; j++) for (int i = 0; i <= n[0]+1; i++)
    w[IX(i,j,k)] = w[IX(i,j,k)] - dt*dzci[k]*(p[IX(i,j,k+1)] - p[IX(i,j,k)]);
}
void o_updatep(ostate *s, double alpha, const double *pp, double *p) {  /* updatep.f90:14-49 */
  const int *n = s->n; const size_t s1 = s->s1, s2 = s->s2; const double *dzci = s->dzci, *dzfi = s->dzfi;
  const double dxi = s->dli[0], dyi = s->dli[1]; const int imp = s->P.impdiff;
  #pragma omp parallel for collapse(2) num_threads(s->nthreads)
  for (int k = 1; k <= n[2]; k++) for (int j = 1; j <= n[1]; j++) for (int i = 1; i <= n[0]; i++) {
    size_t c = IX(i,j,k);
    if (imp == 0) p[c] = p[c] + pp[c];
    else if (imp == 1)
      p[c] = p[c] + pp[c] + alpha*((pp[IX(i+1,j,k)] - 2.*pp[c] + pp[IX(i-1,j,k)])*(dxi*dxi) +
                                   (pp[IX(i,j+1,k)] - 2.*pp[c] + pp[IX(i,j-1,k)])*(dyi*dyi) +
                                   ((pp[IX(i,j,k+1)] - pp[c])*dzci[k] - (pp[c] - pp[IX(i,j,k-1)])*dzci[k-1])*dzfi[k]);
    else
      p[c] = p[c] + pp[c] + alpha*(((pp[IX(i,j,k+1)] - pp[c])*dzci[k] - (pp[c] - pp[IX(i,j,k-1)])*dzci[k-1])*dzfi[k]);
  }
}

/* ------------------------------------------------------------------ tridiagonal: src/solver.f90:82-179 */
static void dgtsv_homebrewed(int n, const double *a, const double *b, const double *c, double *p, int st, double *d) {
  double z = 1./(b[0] + EPS);
  d[0] = c[0]*z; p[0] = p[0]*z;
  for (int l = 1; l < n; l++) {
    z = 1./(b[l] - a[l]*d[l-1] + EPS);
    d[l] = c[l]*z;
    p[(size_t)l*st] = (p[(size_t)l*st] - a[l]*p[(size_t)(l-1)*st])*z;
  }
  for (int l = n-2; l >= 0; l--) p[(size_t)l*st] = p[(size_t)l*st] - d[l]*p[(size_t)(l+1)*st];
}
static void gaussel_line(int n, const double *a, const double *b, const double *c, double lam, int periodic,
                         double *p, int st, double *work /* 4n */) {
  double *bb = work, *d = work + n, *p1 = work + 2*n, *p2 = work + 3*n;
  for (int l = 0; l < n; l++) bb[l] = b[l] + lam;
  if (!periodic) { dgtsv_homebrewed(n, a, bb, c, p, st, d); return; }
  for (int l = 0; l < n-1; l++) p1[l] = p[(size_t)l*st];
  dgtsv_homebrewed(n-1, a, bb, c, p1, 1, d);
  for (int l = 0; l < n; l++) p2[l] = 0.;
  p2[0] = -a[0]; p2[n-2] = -c[n-2];
  dgtsv_homebrewed(n-1, a, bb, c, p2, 1, d);
  double pn = (p[(size_t)(n-1)*st] - c[n-1]*p1[0] - a[n-1]*p1[n-2]) / (bb[n-1] + c[n-1]*p2[0] + a[n-1]*p2[n-2] + EPS);
  p[(size_t)(n-1)*st] = pn;
  for (int l = 0; l < n-1; l++) p[(size_t)l*st] = p1[l] + p2[l]*pn;
}

/* ------------------------------------------------------------------ Poisson solve: src/solver.f90:20-80 (one rank) */
/* the z sweep of the solve alone: gaussel / gaussel_periodic with lambdaxy, solver.f90:56-61,82-151 (pinned: `sol_gz_*` golden vectors made by
   the reference's own routines) */
void o_solver_zsweep(ostate *s, double *p) {
  const int *n = s->n; const size_t s1 = s->s1, s2 = s->s2;
  int periodic_z = CBP(0,3) == 'P' && CBP(1,3) == 'P';
  #pragma omp parallel num_threads(s->nthreads)
  {
    double *work = (double *)malloc(sizeof(double)*4*n[2]);
    #pragma omp for collapse(2)
    for (int j = 1; j <= n[1]; j++) for (int i = 1; i <= n[0]; i++)
      gaussel_line(n[2], s->a, s->b, s->c, s->lambdaxy[(i-1) + (size_t)n[0]*(j-1)], periodic_z, &p[IX(i,j,1)], (int)(s1*s2), work);
    free(work);
  }
}
void o_solver(ostate *s, double *p) {
  const int *n = s->n; const size_t s1 = s->s1, s2 = s->s2;
  #pragma omp parallel for collapse(2) num_threads(s->nthreads)
  for (int k = 1; k <= n[2]; k++) for (int j = 1; j <= n[1]; j++) o_r2r(s->kind_fwd[0], n[0], &p[IX(1,j,k)], 1);
  #pragma omp parallel for collapse(2) num_threads(s->nthreads)
  for (int k = 1; k <= n[2]; k++) for (int i = 1; i <= n[0]; i++) o_r2r(s->kind_fwd[1], n[1], &p[IX(i,1,k)], (int)s1);
  o_solver_zsweep(s, p);
  #pragma omp parallel for collapse(2) num_threads(s->nthreads)
  for (int k = 1; k <= n[2]; k++) for (int i = 1; i <= n[0]; i++) o_r2r(s->kind_bwd[1], n[1], &p[IX(i,1,k)], (int)s1);
  #pragma omp parallel for collapse(2) num_threads(s->nthreads)
  for (int k = 1; k <= n[2]; k++) for (int j = 1; j <= n[1]; j++) {
    o_r2r(s->kind_bwd[0], n[0], &p[IX(1,j,k)], 1);
    for (int i = 1; i <= n[0]; i++) p[IX(i,j,k)] = p[IX(i,j,k)]*s->normfft;
  }
}
void o_solver_gaussel_z(ostate *s, int ivel, double alpha, double *q) {  /* solver.f90:182-233; main.f90:435-445 */
  const int *n = s->n; const size_t s1 = s->s1, s2 = s->s2; int n3 = n[2];
  double *aa = dalloc(n3), *bb = dalloc(n3), *cc = dalloc(n3);
  for (int k = 0; k < n3; k++) { aa[k] = s->av[ivel-1][k]*alpha; bb[k] = s->bv[ivel-1][k]*alpha + 1.; cc[k] = s->cv[ivel-1][k]*alpha; }
  const char *bcz = &s->cbcvel[6*(ivel-1) + 4];
  int qq = (ivel == 3 && bcz[1] == 'D') ? 1 : 0, periodic = bcz[0] == 'P' && bcz[1] == 'P';
  #pragma omp parallel num_threads(s->nthreads)
  {
    double *work = (double *)malloc(sizeof(double)*4*n3);
    #pragma omp for collapse(2)
    for (int j = 1; j <= n[1]; j++) for (int i = 1; i <= n[0]; i++)
      gaussel_line(n3 - qq, aa, bb, cc, 0., periodic, &q[IX(i,j,1)], (int)(s1*s2), work);
    free(work);
  }
  free(aa); free(bb); free(cc);
}

/* 3-D implicit diffusion (_IMPDIFF without _IMPDIFF_1D), main.f90:423-491: (1 + alpha L) q = q* for one velocity component
 * through solver.f90:20-80 with lambdaxy*alpha and aa,bb,cc = a*alpha, b*alpha+1, c*alpha, and the transform kinds, sizes,
 * eigenvalues and normalisation of the component: find_fft / eigenvalues with c_or_f = 'f' along its own direction
 * (initsolver.f90:66-98, fft.f90:63-143,192-245). */
int o_solver_helmholtz(ostate *s, int ivel, double alpha, double *q) {
  const int *n = s->n; const size_t s1 = s->s1, s2 = s->s2; int n3 = n[2];
  const char *bcv = &s->cbcvel[6*(ivel-1)];
  char cf[3] = {'c','c','c'}; cf[ivel-1] = 'f';
  /* transform kinds, sizes, eigenvalues and normalisation of THIS component (initsolver.f90:66-98 with c_or_f, fft.f90:63-143) */
  int kf[2], kb[2], cut[2]; double nrm[2][2], normfft = 1.;
  double *lam[2];
  for (int d = 0; d < 2; d++) {
    find_fft(bcv + 2*d, cf[d], &kf[d], &kb[d], nrm[d]);
    cut[d] = (bcv[2*d] == 'D' && bcv[2*d+1] == 'D' && cf[d] == 'f') ? 1 : 0;      /* one point less with Dirichlet at the faces */
    lam[d] = dalloc(n[d]); eigenvalues(n[d], bcv + 2*d, cf[d], lam[d]);
    for (int l = 0; l < n[d]; l++) lam[d][l] = lam[d][l]*(s->dli[d]*s->dli[d]);
    normfft = normfft*nrm[d][0]*(s->P.ng[d] + nrm[d][1] - cut[d]);
  }
  normfft = 1./normfft;
  double *aa = dalloc(n3), *bb = dalloc(n3), *cc = dalloc(n3);
  for (int k = 0; k < n3; k++) { aa[k] = s->av[ivel-1][k]*alpha; bb[k] = s->bv[ivel-1][k]*alpha + 1.; cc[k] = s->cv[ivel-1][k]*alpha; }
  int qq = (ivel == 3 && bcv[5] == 'D') ? 1 : 0, periodic = bcv[4] == 'P' && bcv[5] == 'P';
  #pragma omp parallel for collapse(2) num_threads(s->nthreads)
  for (int k = 1; k <= n[2]; k++) for (int j = 1; j <= n[1]; j++) o_r2r(kf[0], n[0] - cut[0], &q[IX(1,j,k)], 1);
  #pragma omp parallel for collapse(2) num_threads(s->nthreads)
  for (int k = 1; k <= n[2]; k++) for (int i = 1; i <= n[0]; i++) o_r2r(kf[1], n[1] - cut[1], &q[IX(i,1,k)], (int)s1);
  #pragma omp parallel num_threads(s->nthreads)
  {
    double *work = (double *)malloc(sizeof(double)*4*n3);
    #pragma omp for collapse(2)
    for (int j = 1; j <= n[1]; j++) for (int i = 1; i <= n[0]; i++)
      gaussel_line(n3 - qq, aa, bb, cc, (lam[0][i-1] + lam[1][j-1])*alpha, periodic, &q[IX(i,j,1)], (int)(s1*s2), work);
    free(work);
  }
  #pragma omp parallel for collapse(2) num_threads(s->nthreads)
  for (int k = 1; k <= n[2]; k++) for (int i = 1; i <= n[0]; i++) o_r2r(kb[1], n[1] - cut[1], &q[IX(i,1,k)], (int)s1);
  #pragma omp parallel for collapse(2) num_threads(s->nthreads)
  for (int k = 1; k <= n[2]; k++) for (int j = 1; j <= n[1]; j++) {
    o_r2r(kb[0], n[0] - cut[0], &q[IX(1,j,k)], 1);
    for (int i = 1; i <= n[0]; i++) q[IX(i,j,k)] = q[IX(i,j,k)]*normfft;
  }
  free(aa); free(bb); free(cc); free(lam[0]); free(lam[1]);
  return 0;
}
/* boundary r.h.s. of all three directions for the 3-D implicit step (main.f90:424-431) */
void o_updt_rhs_b_vel(ostate *s, int ivel, double alpha, double *q) {
  const int *n = s->n; char cf[3] = {'c','c','c'}; cf[ivel-1] = 'f';
  size_t nx = (size_t)n[1]*n[2]*2, ny = (size_t)n[0]*n[2]*2, nz = (size_t)n[0]*n[1]*2;
  double *rx = dalloc(nx), *ry = dalloc(ny), *rz = dalloc(nz);
  const obound *bc = ivel == 1 ? &s->bcu : ivel == 2 ? &s->bcv : &s->bcw;
  cmpt_rhs_b(s, &s->cbcvel[6*(ivel-1)], bc, cf, rx, ry, rz);
  for (size_t i = 0; i < nx; i++) rx[i] = rx[i]*alpha;
  for (size_t i = 0; i < ny; i++) ry[i] = ry[i]*alpha;
  for (size_t i = 0; i < nz; i++) rz[i] = rz[i]*alpha;
  updt_rhs_b(s, cf, &s->cbcvel[6*(ivel-1)], rx, ry, rz, q);
  free(rx); free(ry); free(rz);
}

/* ------------------------------------------------------------------ diagnostics */
double o_chkdt(ostate *s, const double *visct, const double *u, const double *v, const double *w) { /* chkdt.f90:17-99 */
  const int *n = s->n; const size_t s1 = s->s1, s2 = s->s2; const double *dzci = s->dzci, *dzfi = s->dzfi;
  double dxi = 1./s->dl[0], dyi = 1./s->dl[1], dl2i = dxi*dxi + dyi*dyi, visc = s->visc;
  double dti = 0., dtid = 0.; const int imp = s->P.impdiff;
  #pragma omp parallel for collapse(2) schedule(static) reduction(max:dti,dtid) num_threads(s->nthreads)
  for (int k = 1; k <= n[2]; k++) for (int j = 1; j <= n[1]; j++) for (int i = 1; i <= n[0]; i++) {
    double ux = fabs(u[IX(i,j,k)]);
    double vx = 0.25*fabs(v[IX(i,j,k)] + v[IX(i,j-1,k)] + v[IX(i+1,j,k)] + v[IX(i+1,j-1,k)]);
    double wx = 0.25*fabs(w[IX(i,j,k)] + w[IX(i,j,k-1)] + w[IX(i+1,j,k)] + w[IX(i+1,j,k-1)]);
    double uy = 0.25*fabs(u[IX(i,j,k)] + u[IX(i,j+1,k)] + u[IX(i-1,j+1,k)] + u[IX(i-1,j,k)]);
    double vy = fabs(v[IX(i,j,k)]);
    double wy = 0.25*fabs(w[IX(i,j,k)] + w[IX(i,j+1,k)] + w[IX(i,j+1,k-1)] + w[IX(i,j,k-1)]);
    double uz = 0.25*fabs(u[IX(i,j,k)] + u[IX(i-1,j,k)] + u[IX(i-1,j,k+1)] + u[IX(i,j,k+1)]);
    double vz = 0.25*fabs(v[IX(i,j,k)] + v[IX(i,j-1,k)] + v[IX(i,j-1,k+1)] + v[IX(i,j,k+1)]);
    double wz = fabs(w[IX(i,j,k)]);
    double dtix = ux*dxi + vx*dyi + wx*dzfi[k], dtiy = uy*dxi + vy*dyi + wy*dzfi[k], dtiz = uz*dxi + vz*dyi + wz*dzci[k];
    dti = fmax(fmax(fmax(dti, dtix), dtiy), dtiz);
    double viscx = 0.5*(visct[IX(i,j,k)] + visct[IX(i+1,j,k)]), viscy = 0.5*(visct[IX(i,j,k)] + visct[IX(i,j+1,k)]),
           viscz = 0.5*(visct[IX(i,j,k)] + visct[IX(i,j,k+1)]);
    double dtidx = viscx*(dl2i + dzfi[k]*dzfi[k]), dtidy = viscy*(dl2i + dzfi[k]*dzfi[k]), dtidz = viscz*(dl2i + dzci[k]*dzci[k]);
    if (imp != 1) {
      dtidx = dtidx + visc*dl2i; dtidy = dtidy + visc*dl2i; dtidz = dtidz + visc*dl2i;
      if (imp != 2) { dtidx = dtidx + visc*(dzfi[k]*dzfi[k]); dtidy = dtidy + visc*(dzfi[k]*dzfi[k]); dtidz = dtidz + visc*(dzci[k]*dzci[k]); }
    }
    dtid = fmax(fmax(fmax(dtid, dtidx), dtidy), dtidz);
  }
  if (dti == 0.) dti = 1.;
  if (dtid == 0.) dtid = EPS;
  return fmin(0.4125/dtid, 1.732/dti);
}
void o_chkdiv(ostate *s, const double *u, const double *v, const double *w, double *divtot, double *divmax) { /* chkdiv.f90:16-52 */
  const int *n = s->n; const size_t s1 = s->s1, s2 = s->s2; double dxi = s->dli[0], dyi = s->dli[1];
  double dt_ = 0., dm = 0.;
  if (!s->team_sums) {
    for (int k = 1; k <= n[2]; k++) for (int j = 1; j <= n[1]; j++) for (int i = 1; i <= n[0]; i++) {
      double div = (w[IX(i,j,k)] - w[IX(i,j,k-1)])*s->dzfi[k] + (v[IX(i,j,k)] - v[IX(i,j-1,k)])*dyi + (u[IX(i,j,k)] - u[IX(i-1,j,k)])*dxi;
      dm = fmax(dm, fabs(div)); dt_ = dt_ + div;
    }
    *divtot = dt_; *divmax = dm; return;
  }
  double *part = (double *)malloc(sizeof(double)*(size_t)(n[2]+2));
  /* (the maximum is exact in any order; the total is summed plane by plane in the reference's (i, j) order, planes added in order: one value for every team size) */
  #pragma omp parallel for schedule(static) reduction(max:dm) num_threads(s->nthreads)
  for (int k = 1; k <= n[2]; k++) { double a = 0.;
    for (int j = 1; j <= n[1]; j++) for (int i = 1; i <= n[0]; i++) {
      double div = (w[IX(i,j,k)] - w[IX(i,j,k-1)])*s->dzfi[k] + (v[IX(i,j,k)] - v[IX(i,j-1,k)])*dyi + (u[IX(i,j,k)] - u[IX(i-1,j,k)])*dxi;
      dm = fmax(dm, fabs(div)); a = a + div; }
    part[k] = a; }
  for (int k = 1; k <= n[2]; k++) dt_ = dt_ + part[k];
  free(part);
  *divtot = dt_; *divmax = dm;
}

/* ------------------------------------------------------------------ plane statistics: first block of out1d_single_point_chan
   (output.f90:509-700, idir = 3): 27 sums per z plane times dx dy/(lx ly); buf is (27, n3) in Fortran order.
   TEST INFRASTRUCTURE; pinned: tests/test_oracle_golden.py::test_plane_statistics compares it with the output of the reference's own
   routine (compiled from its lines by oracle/ref/Makefile) on the 14 golden end-of-step states, <= 4e-16 of each column's largest entry. */
void o_stats_chan(ostate *s, const double *u, const double *v, const double *w, const double *p, const double *visct, double *buf) {
  const int *n = s->n; const size_t s1 = s->s1, s2 = s->s2;
  const double dx = s->dl[0], dy = s->dl[1], ratio = dx*dy/(s->P.l[0]*s->P.l[1]);
  for (int k = 1; k <= n[2]; k++) {
    double b[27]; for (int q = 0; q < 27; q++) b[q] = 0.;
    for (int j = 1; j <= n[1]; j++) for (int i = 1; i <= n[0]; i++) {
      const double uc = u[IX(i,j,k)], vc = v[IX(i,j,k)], wc = w[IX(i,j,k)], pc = p[IX(i,j,k)];
      b[0] += uc; b[1] += vc; b[2] += wc;
      b[3] += uc*uc; b[4] += vc*vc; b[5] += wc*wc;
      b[6] += 0.25*(u[IX(i,j,k+1)] + uc)*(wc + w[IX(i+1,j,k)]);
      b[7] += uc*uc*uc; b[8] += vc*vc*vc; b[9] += wc*wc*wc;
      b[10] += (uc*uc)*(uc*uc); b[11] += (vc*vc)*(vc*vc); b[12] += (wc*wc)*(wc*wc);
      b[13] += pc; b[14] += pc*pc;
      const double ox = (w[IX(i,j+1,k)] - wc)/dy - (v[IX(i,j,k+1)] - vc)/s->dzc[k];
      const double oy = (u[IX(i,j,k+1)] - uc)/s->dzc[k] - (w[IX(i+1,j,k)] - wc)/dx;
      const double oz = (v[IX(i+1,j,k)] - vc)/dx - (u[IX(i,j+1,k)] - uc)/dy;
      b[15] += ox; b[16] += oy; b[17] += oz; b[18] += ox*ox; b[19] += oy*oy; b[20] += oz*oz;
      const double s_ccc = visct[IX(i,j,k)], s_pcc = visct[IX(i+1,j,k)], s_cpc = visct[IX(i,j+1,k)], s_ccp = visct[IX(i,j,k+1)], s_pcp = visct[IX(i+1,j,k+1)];
      const double dudx_ip = (u[IX(i+1,j,k)] - uc)/dx, dudx_im = (uc - u[IX(i-1,j,k)])/dx;
      const double dvdy_jp = (v[IX(i,j+1,k)] - vc)/dy, dvdy_jm = (vc - v[IX(i,j-1,k)])/dy;
      const double dwdz_kp = (w[IX(i,j,k+1)] - wc)/s->dzf[k+1], dwdz_km = (wc - w[IX(i,j,k-1)])/s->dzf[k];
      const double dudz = (u[IX(i,j,k+1)] - uc)/s->dzc[k], dwdx = (w[IX(i+1,j,k)] - wc)/dx;
      b[21] -= 0.5*(s_pcc*(dudx_ip + dudx_ip) + s_ccc*(dudx_im + dudx_im));
      b[22] -= 0.5*(s_cpc*(dvdy_jp + dvdy_jp) + s_ccc*(dvdy_jm + dvdy_jm));
      b[23] -= 0.5*(s_ccp*(dwdz_kp + dwdz_kp) + s_ccc*(dwdz_km + dwdz_km));
      b[24] -= 0.25*(s_ccc + s_pcc + s_ccp + s_pcp)*(dudz + dwdx);
      b[25] += s_ccc;
      b[26] += dudz;
    }
    for (int q = 0; q < 27; q++) buf[q + 27*(size_t)(k-1)] = b[q]*ratio;
  }
}

/* out1d, out1d_chan (idir = 3), out2d_duct (streamwise x): src/output.f90:50-163, 317-405, 406-507. TEST INFRASTRUCTURE; pinned against the files the
   reference's own routines write (compiled from their lines by oracle/ref/Makefile; they print 8 significant digits) */
void o_out1d(ostate *s, int idir, int use_dzc, const double *p, double *buf) {
  const int *n = s->n; const size_t s1 = s->s1, s2 = s->s2; const double *dz = use_dzc ? s->dzc : s->dzf;
  const double dx = s->dl[0], dy = s->dl[1];
  if (idir == 3) { const double r = dx*dy/(s->P.l[0]*s->P.l[1]);
    for (int k = 1; k <= n[2]; k++) { double a = 0.; for (int j = 1; j <= n[1]; j++) for (int i = 1; i <= n[0]; i++) a += p[IX(i,j,k)]; buf[k-1] = a*r; }
  } else if (idir == 2) { const double r = dx/(s->P.l[0]*s->P.l[2]);
    for (int j = 1; j <= n[1]; j++) { double a = 0.; for (int k = 1; k <= n[2]; k++) for (int i = 1; i <= n[0]; i++) a += p[IX(i,j,k)]*dz[k]; buf[j-1] = a*r; }
  } else { const double r = dy/(s->P.l[1]*s->P.l[2]);
    for (int i = 1; i <= n[0]; i++) { double a = 0.; for (int k = 1; k <= n[2]; k++) for (int j = 1; j <= n[1]; j++) a += p[IX(i,j,k)]*dz[k]; buf[i-1] = a*r; }
  }
}
void o_out1d_chan(ostate *s, const double *u, const double *v, const double *w, double *buf) {
  const int *n = s->n; const size_t s1 = s->s1, s2 = s->s2; const double r = s->dl[0]*s->dl[1]/(s->P.l[0]*s->P.l[1]);
  for (int k = 1; k <= n[2]; k++) {
    double b[7] = {0.,0.,0.,0.,0.,0.,0.};
    for (int j = 1; j <= n[1]; j++) for (int i = 1; i <= n[0]; i++) {
      const double uc = u[IX(i,j,k)], vc = v[IX(i,j,k)], wc = w[IX(i,j,k)], wm = w[IX(i,j,k-1)];
      b[0] += uc; b[1] += vc; b[2] += 0.50*(wm + wc);
      b[3] += uc*uc; b[4] += vc*vc; b[5] += 0.50*(wc*wc + wm*wm);
      b[6] += 0.25*(u[IX(i-1,j,k)] + uc)*(wm + wc);
    }
    for (int q = 0; q < 7; q++) buf[q + 7*(size_t)(k-1)] = b[q]*r;
  }
}
void o_out2d_duct(ostate *s, const double *u, const double *v, const double *w, double *buf) {
  const int *n = s->n; const size_t s1 = s->s1, s2 = s->s2; const double r = s->dl[0]/s->P.l[0];
  for (int k = 1; k <= n[2]; k++) for (int j = 1; j <= n[1]; j++) {
    double b[9] = {0.,0.,0.,0.,0.,0.,0.,0.,0.};
    for (int i = 1; i <= n[0]; i++) {
      const double uc = u[IX(i,j,k)], um = u[IX(i-1,j,k)], vc = v[IX(i,j,k)], vm = v[IX(i,j-1,k)], wc = w[IX(i,j,k)], wm = w[IX(i,j,k-1)];
      b[0] += uc; b[1] += 0.5*(vm + vc); b[2] += 0.5*(wm + wc);
      b[3] += uc*uc; b[4] += 0.5*(vm*vm + vc*vc); b[5] += 0.5*(wm*wm + wc*wc);
      b[6] += 0.25*(um + uc)*(vm + vc); b[7] += 0.25*(um + uc)*(wm + wc); b[8] += 0.25*(vm + vc)*(wm + wc);
    }
    for (int q = 0; q < 9; q++) buf[q + 9*((size_t)(j-1) + (size_t)n[1]*(k-1))] = b[q]*r;
  }
}

/* ------------------------------------------------------------------ SGS: src/sgs.f90 */
static void extrapolate(ostate *s, double *p, int iface, int use_cbc) {   /* sgs.f90:682-767 */
  const int *n = s->n; const size_t s1 = s->s1, s2 = s->s2; int done[6]; double factor0, factor1;
  if (use_cbc) { factor0 = factor1 = 1.;
    for (int d = 1; d <= 3; d++) for (int ib = 0; ib <= 1; ib++) done[ib + 2*(d-1)] = ISB(ib,d) && CBV(ib,d,d) == 'D' && iface != d;
  } else { factor0 = (1./s->dzci[0])*s->dzci[1]; factor1 = (1./s->dzci[n[2]])*s->dzci[n[2]-1];
    for (int d = 1; d <= 3; d++) for (int ib = 0; ib <= 1; ib++) done[ib + 2*(d-1)] = ISB(ib,d) && LWM(ib,d) != 0 && iface != d; }
  if (done[0]) {
    #pragma omp parallel for schedule(static) num_threads(s->nthreads)
    for (int k = 0; k <= n[2]+1; k++) for (int j = 0; j <= n[1]+1; j++) p[IX(0,j,k)] = 2.*p[IX(1,j,k)] - p[IX(2,j,k)]; }
  if (done[1]) {
    #pragma omp parallel for schedule(static) num_threads(s->nthreads)
    for (int k = 0; k <= n[2]+1; k++) for (int j = 0; j <= n[1]+1; j++) p[IX(n[0]+1,j,k)] = 2.*p[IX(n[0],j,k)] - p[IX(n[0]-1,j,k)]; }
  if (done[2]) {
    #pragma omp parallel for schedule(static) num_threads(s->nthreads)
    for (int k = 0; k <= n[2]+1; k++) for (int i = 0; i <= n[0]+1; i++) p[IX(i,0,k)] = 2.*p[IX(i,1,k)] - p[IX(i,2,k)]; }
  if (done[3]) {
    #pragma omp parallel for schedule(static) num_threads(s->nthreads)
    for (int k = 0; k <= n[2]+1; k++) for (int i = 0; i <= n[0]+1; i++) p[IX(i,n[1]+1,k)] = 2.*p[IX(i,n[1],k)] - p[IX(i,n[1]-1,k)]; }
  if (done[4]) {
    #pragma omp parallel for schedule(static) num_threads(s->nthreads)
    for (int j = 0; j <= n[1]+1; j++) for (int i = 0; i <= n[0]+1; i++) p[IX(i,j,0)] = (1.+factor0)*p[IX(i,j,1)] - factor0*p[IX(i,j,2)]; }
  if (done[5]) {
    #pragma omp parallel for schedule(static) num_threads(s->nthreads)
    for (int j = 0; j <= n[1]+1; j++) for (int i = 0; i <= n[0]+1; i++) p[IX(i,j,n[2]+1)] = (1.+factor1)*p[IX(i,j,n[2])] - factor1*p[IX(i,j,n[2]-1)]; }
}
static void strain_rate(ostate *s, const double *u, const double *v, const double *w, double *s0, double **sij) { /* sgs.f90:1019-1110 */
  const int *n = s->n; const size_t s1 = s->s1, s2 = s->s2; const double dxi = s->dli[0], dyi = s->dli[1]; const double *dzci = s->dzci, *dzfi = s->dzfi;
  #pragma omp parallel for collapse(2) num_threads(s->nthreads)
  for (int k = 1; k <= n[2]; k++) for (int j = 1; j <= n[1]; j++) for (int i = 1; i <= n[0]; i++) {
    #define LD(a,di,dj,dk) a[IX(i+(di),j+(dj),k+(dk))]
    double u_mcm=LD(u,-1,0,-1),u_ccm=LD(u,0,0,-1),u_mmc=LD(u,-1,-1,0),u_cmc=LD(u,0,-1,0),u_mcc=LD(u,-1,0,0),u_ccc=LD(u,0,0,0),
           u_mpc=LD(u,-1,1,0),u_cpc=LD(u,0,1,0),u_mcp=LD(u,-1,0,1),u_ccp=LD(u,0,0,1);
    double v_cmm=LD(v,0,-1,-1),v_ccm=LD(v,0,0,-1),v_mmc=LD(v,-1,-1,0),v_cmc=LD(v,0,-1,0),v_pmc=LD(v,1,-1,0),v_mcc=LD(v,-1,0,0),
           v_ccc=LD(v,0,0,0),v_pcc=LD(v,1,0,0),v_cmp=LD(v,0,-1,1),v_ccp=LD(v,0,0,1);
    double w_cmm=LD(w,0,-1,-1),w_mcm=LD(w,-1,0,-1),w_ccm=LD(w,0,0,-1),w_pcm=LD(w,1,0,-1),w_cpm=LD(w,0,1,-1),w_cmc=LD(w,0,-1,0),
           w_mcc=LD(w,-1,0,0),w_ccc=LD(w,0,0,0),w_pcc=LD(w,1,0,0),w_cpc=LD(w,0,1,0);
    #undef LD
    double s11 = (u_ccc-u_mcc)*dxi, s22 = (v_ccc-v_cmc)*dyi, s33 = (w_ccc-w_ccm)*dzfi[k];
    double s12 = .125*((u_cpc-u_ccc)*dyi + (v_pcc-v_ccc)*dxi + (u_ccc-u_cmc)*dyi + (v_pmc-v_cmc)*dxi +
                       (u_mpc-u_mcc)*dyi + (v_ccc-v_mcc)*dxi + (u_mcc-u_mmc)*dyi + (v_cmc-v_mmc)*dxi);
    double s13 = .125*((u_ccp-u_ccc)*dzci[k] + (w_pcc-w_ccc)*dxi + (u_ccc-u_ccm)*dzci[k-1] + (w_pcm-w_ccm)*dxi +
                       (u_mcp-u_mcc)*dzci[k] + (w_ccc-w_mcc)*dxi + (u_mcc-u_mcm)*dzci[k-1] + (w_ccm-w_mcm)*dxi);
    double s23 = .125*((v_ccp-v_ccc)*dzci[k] + (w_cpc-w_ccc)*dyi + (v_ccc-v_ccm)*dzci[k-1] + (w_cpm-w_ccm)*dyi +
                       (v_cmp-v_cmc)*dzci[k] + (w_ccc-w_cmc)*dyi + (v_cmc-v_cmm)*dzci[k-1] + (w_ccm-w_cmm)*dyi);
    size_t c = IX(i,j,k);
    s0[c] = sqrt(2.*(s11*s11 + s22*s22 + s33*s33 + 2.*(s12*s12 + s13*s13 + s23*s23)));
    if (sij) { sij[0][c] = s11; sij[1][c] = s22; sij[2][c] = s33; sij[3][c] = s12; sij[4][c] = s13; sij[5][c] = s23; }
  }
}
static void filter3d(ostate *s, const double *p, double *pf) {   /* sgs.f90:616-680 */
  const int *n = s->n; const size_t s1 = s->s1, s2 = s->s2;
  #pragma omp parallel for collapse(2) num_threads(s->nthreads)
  for (int k = 1; k <= n[2]; k++) for (int j = 1; j <= n[1]; j++) for (int i = 1; i <= n[0]; i++) {
    #define Q(di,dj,dk) p[IX(i+(di),j+(dj),k+(dk))]
    pf[IX(i,j,k)] = (8.*(Q(0,0,0)) +
      4.*(Q(-1,0,0) + Q(0,-1,0) + Q(0,0,-1) + Q(1,0,0) + Q(0,1,0) + Q(0,0,1)) +
      2.*(Q(0,-1,-1) + Q(-1,0,-1) + Q(-1,-1,0) + Q(0,1,-1) + Q(1,0,-1) + Q(1,-1,0) +
          Q(0,-1,1) + Q(-1,0,1) + Q(-1,1,0) + Q(0,1,1) + Q(1,0,1) + Q(1,1,0)) +
      1.*(Q(-1,-1,-1) + Q(1,-1,-1) + Q(-1,1,-1) + Q(1,1,-1) + Q(-1,-1,1) + Q(1,-1,1) + Q(-1,1,1) + Q(1,1,1)))/64.;
    #undef Q
  }
}
/* copy of a haloed field by the team, split over (k, j) like the loops that use it (a memcpy by one thread would also first-touch a fresh scratch
   field on that thread's NUMA node) */
static void pcopy(ostate *s, double *d, const double *a) {
  const int *n = s->n; const size_t s1 = s->s1, s2 = s->s2;
  #pragma omp parallel for collapse(2) schedule(static) num_threads(s->nthreads)
  for (int k = 0; k <= n[2] + 1; k++) for (int j = 0; j <= n[1] + 1; j++) memcpy(&d[IX(0,j,k)], &a[IX(0,j,k)], sizeof(double)*s1);
}
static void ave1d_channel_z(ostate *s, double *p) {       /* sgs.f90:433-482, idir = 3 */
  const int *n = s->n; const size_t s1 = s->s1, s2 = s->s2; double gar = s->dl[0]*s->dl[1]/(s->P.l[0]*s->P.l[1]);
  #pragma omp parallel for schedule(static) num_threads(s->nthreads)
  for (int k = 1; k <= n[2]; k++) { double a = 0.;
    for (int j = 1; j <= n[1]; j++) for (int i = 1; i <= n[0]; i++) a = a + p[IX(i,j,k)];
    a = a*gar;
    for (int j = 0; j <= n[1]+1; j++) for (int i = 0; i <= n[0]+1; i++) p[IX(i,j,k)] = a; }
}
void o_cmpt_sgs(ostate *s, const double *u, const double *v, const double *w, double *visct) {   /* sgs.f90:21-386 */
  const int *n = s->n; const size_t s1 = s->s1, s2 = s->s2, nt = s->ntot; const double *dl = s->dl;
  if (s->P.sgstype == 0) { if (s->sgs_first) { s->sgs_first = 0; memset(visct, 0, sizeof(double)*nt); } return; }
  if (s->sgs_first) {
    s->sgs_first = 0; s->s0 = dalloc(nt);
    int nw = s->P.sgstype == 1 ? 3 : 6; for (int m = 0; m < nw; m++) s->wk[m] = dalloc(nt);
    for (int d = 1; d <= 3; d++) for (int ib = 0; ib <= 1; ib++) s->is_wall[ib + 2*(d-1)] = (ISB(ib,d) && CBV(ib,d,d) == 'D') ? 1. : 0.;
    if (s->P.sgstype == 2) {
      s->uc = dalloc(nt); s->vc = dalloc(nt); s->wc = dalloc(nt); s->uf = dalloc(nt); s->vf = dalloc(nt); s->wf = dalloc(nt);
      s->alph2 = dalloc(nt); for (int m = 0; m < 6; m++) { s->sij[m] = dalloc(nt); s->mij[m] = dalloc(nt); }
      for (size_t q = 0; q < nt; q++) s->alph2[q] = 4.00;      /* cmpt_alph2, sgs.f90:769-822 */
      for (int k = 0; k <= n[2]+1; k++) for (int j = 0; j <= n[1]+1; j++) for (int i = 0; i <= n[0]+1; i++) {
        int near = (s->is_wall[0] != 0. && i == 1) || (s->is_wall[1] != 0. && i == n[0]) || (s->is_wall[2] != 0. && j == 1) ||
                   (s->is_wall[3] != 0. && j == n[1]) || (s->is_wall[4] != 0. && k == 1) || (s->is_wall[5] != 0. && k == n[2]);
        if (near) s->alph2[IX(i,j,k)] = 2.52; }
    }
  }
  double *s0 = s->s0, **wk = s->wk;
  pcopy(s, wk[0], u); pcopy(s, wk[1], v); pcopy(s, wk[2], w);
  extrapolate(s, wk[0], 1, 0); extrapolate(s, wk[1], 2, 0); extrapolate(s, wk[2], 3, 0);
  if (s->P.sgstype == 1) {
    strain_rate(s, wk[0], wk[1], wk[2], s0, NULL);
    const double dxi = s->dli[0], dyi = s->dli[1], visc = s->visc, visci = 1./visc; const double *zc = s->zc, *dzci = s->dzci, *dzf = s->dzf;
    double sumw = 0.; for (int q = 0; q < 6; q++) sumw += s->is_wall[q];
    const double one_third = 1./3., l3 = s->P.l[2];
    #pragma omp parallel for collapse(2) num_threads(s->nthreads)
    for (int k = 1; k <= n[2]; k++) for (int j = 1; j <= n[1]; j++) for (int i = 1; i <= n[0]; i++) {
      double fd;
      if (sumw == 0.) fd = 1.;
      else {
        double dw[6], tauw[2], tauw_s = 0.;
        /* `(i-0.5)` is default real in the reference (sgs.f90:108-111) */
        dw[0] = dl[0]*(double)((float)i - 0.5f); dw[1] = dl[0]*(double)((float)(n[0]-i) + 0.5f);
        dw[2] = dl[1]*(double)((float)j - 0.5f); dw[3] = dl[1]*(double)((float)(n[1]-j) + 0.5f);
        dw[4] = zc[k]; dw[5] = l3 - zc[k];
        for (int q = 0; q < 6; q++) dw[q] = dw[q]*s->is_wall[q] + BIG*(1. - s->is_wall[q]);
        int loc = 0; for (int q = 1; q < 6; q++) if (dw[q] < dw[loc]) loc = q;
        double dw_min = dw[loc];
        switch (loc + 1) {
        case 1: tauw[0] = v[IX(1,j,k)]-v[IX(0,j,k)]+v[IX(1,j-1,k)]-v[IX(0,j-1,k)]; tauw[1] = w[IX(1,j,k)]-w[IX(0,j,k)]+w[IX(1,j,k-1)]-w[IX(0,j,k-1)];
                tauw_s = sqrt(tauw[0]*tauw[0]+tauw[1]*tauw[1])*dxi; break;
        case 2: tauw[0] = v[IX(n[0],j,k)]-v[IX(n[0]+1,j,k)]+v[IX(n[0],j-1,k)]-v[IX(n[0]+1,j-1,k)];
                tauw[1] = w[IX(n[0],j,k)]-w[IX(n[0]+1,j,k)]+w[IX(n[0],j,k-1)]-w[IX(n[0]+1,j,k-1)];
                tauw_s = sqrt(tauw[0]*tauw[0]+tauw[1]*tauw[1])*dxi; break;
        case 3: tauw[0] = u[IX(i,1,k)]-u[IX(i,0,k)]+u[IX(i-1,1,k)]-u[IX(i-1,0,k)]; tauw[1] = w[IX(i,1,k)]-w[IX(i,0,k)]+w[IX(i,1,k-1)]-w[IX(i,0,k-1)];
                tauw_s = sqrt(tauw[0]*tauw[0]+tauw[1]*tauw[1])*dyi; break;
        case 4: tauw[0] = u[IX(i,n[1],k)]-u[IX(i,n[1]+1,k)]+u[IX(i-1,n[1],k)]-u[IX(i-1,n[1]+1,k)];
                tauw[1] = w[IX(i,n[1],k)]-w[IX(i,n[1]+1,k)]+w[IX(i,n[1],k-1)]-w[IX(i,n[1]+1,k-1)];
                tauw_s = sqrt(tauw[0]*tauw[0]+tauw[1]*tauw[1])*dyi; break;
        case 5: tauw[0] = u[IX(i,j,1)]-u[IX(i,j,0)]+u[IX(i-1,j,1)]-u[IX(i-1,j,0)]; tauw[1] = v[IX(i,j,1)]-v[IX(i,j,0)]+v[IX(i,j-1,1)]-v[IX(i,j-1,0)];
                tauw_s = sqrt(tauw[0]*tauw[0]+tauw[1]*tauw[1])*dzci[0]; break;
        case 6: tauw[0] = u[IX(i,j,n[2])]-u[IX(i,j,n[2]+1)]+u[IX(i-1,j,n[2])]-u[IX(i-1,j,n[2]+1)];
                tauw[1] = v[IX(i,j,n[2])]-v[IX(i,j,n[2]+1)]+v[IX(i,j-1,n[2])]-v[IX(i,j-1,n[2]+1)];
                tauw_s = sqrt(tauw[0]*tauw[0]+tauw[1]*tauw[1])*dzci[n[2]]; break;
        }
        tauw_s = 0.5*visc*tauw_s;
        double dw_plus = dw_min*sqrt(tauw_s)*visci;
        fd = 1. - exp(-dw_plus/25.);
      }
      double del = pow(dl[0]*dl[1]*dzf[k], one_third);
      double t = C_SMAG*del*fd;
      visct[IX(i,j,k)] = (t*t)*s0[IX(i,j,k)];
    }
    return;
  }
  /* dynamic Smagorinsky, sgs.f90:153-380 */
  double **sij = s->sij, **mij = s->mij, **lij = s->sij;
  strain_rate(s, wk[0], wk[1], wk[2], s0, sij);
  pcopy(s, visct, s0);
  o_boundp(s, 1, s0); for (int m = 0; m < 6; m++) o_boundp(s, 1, sij[m]);
  for (int m = 0; m < 6; m++) {
    #pragma omp parallel for schedule(static) num_threads(s->nthreads)
    for (size_t q = 0; q < nt; q++) wk[m][q] = s0[q]*sij[m][q]; }
  for (int m = 0; m < 6; m++) extrapolate(s, wk[m], 0, 1);
  for (int m = 0; m < 6; m++) filter3d(s, wk[m], mij[m]);
  pcopy(s, wk[0], u); pcopy(s, wk[1], v); pcopy(s, wk[2], w);
  extrapolate(s, wk[0], 1, 1); extrapolate(s, wk[1], 2, 1); extrapolate(s, wk[2], 3, 1);
  filter3d(s, wk[0], s->uf); filter3d(s, wk[1], s->vf); filter3d(s, wk[2], s->wf);
  bounduvw_bc(s, &s->bcuf, &s->bcvf, &s->bcwf, 0, 0, s->uf, s->vf, s->wf);
  extrapolate(s, s->uf, 1, 0); extrapolate(s, s->vf, 2, 0); extrapolate(s, s->wf, 3, 0);
  strain_rate(s, s->uf, s->vf, s->wf, s0, sij);
  #pragma omp parallel for collapse(2) schedule(static) num_threads(s->nthreads)
  for (int k = 1; k <= n[2]; k++) for (int j = 1; j <= n[1]; j++) for (int i = 1; i <= n[0]; i++) { size_t c = IX(i,j,k);
    for (int m = 0; m < 6; m++) mij[m][c] = 2.*(mij[m][c] - s->alph2[c]*s0[c]*sij[m][c]); }
  #pragma omp parallel for collapse(2) schedule(static) num_threads(s->nthreads)
  for (int k = 1; k <= n[2]; k++) for (int j = 1; j <= n[1]; j++) for (int i = 1; i <= n[0]; i++) {   /* interpolate, sgs.f90:850-870 */
    size_t c = IX(i,j,k);
    s->uc[c] = 0.5*(u[c] + u[IX(i-1,j,k)]); s->vc[c] = 0.5*(v[c] + v[IX(i,j-1,k)]); s->wc[c] = 0.5*(w[c] + w[IX(i,j,k-1)]); }
  o_boundp(s, 1, s->uc); o_boundp(s, 1, s->vc); o_boundp(s, 1, s->wc);
  #pragma omp parallel for schedule(static) num_threads(s->nthreads)
  for (size_t q = 0; q < nt; q++) {
    wk[0][q] = s->uc[q]*s->uc[q]; wk[1][q] = s->vc[q]*s->vc[q]; wk[2][q] = s->wc[q]*s->wc[q];
    wk[3][q] = s->uc[q]*s->vc[q]; wk[4][q] = s->uc[q]*s->wc[q]; wk[5][q] = s->vc[q]*s->wc[q]; }
  for (int m = 0; m < 6; m++) extrapolate(s, wk[m], 0, 1);
  for (int m = 0; m < 6; m++) filter3d(s, wk[m], lij[m]);
  extrapolate(s, s->uc, 0, 1); extrapolate(s, s->vc, 0, 1); extrapolate(s, s->wc, 0, 1);
  filter3d(s, s->uc, s->uf); filter3d(s, s->vc, s->vf); filter3d(s, s->wc, s->wf);
  #pragma omp parallel for collapse(2) schedule(static) num_threads(s->nthreads)
  for (int k = 1; k <= n[2]; k++) for (int j = 1; j <= n[1]; j++) for (int i = 1; i <= n[0]; i++) { size_t c = IX(i,j,k);
    double m_[6], l_[6]; for (int m = 0; m < 6; m++) { m_[m] = mij[m][c]; l_[m] = lij[m][c]; }
    double uf = s->uf[c], vf = s->vf[c], wf = s->wf[c];
    l_[0] = l_[0] - uf*uf; l_[1] = l_[1] - vf*vf; l_[2] = l_[2] - wf*wf; l_[3] = l_[3] - uf*vf; l_[4] = l_[4] - uf*wf; l_[5] = l_[5] - vf*wf;
    wk[0][c] = m_[0]*l_[0] + m_[1]*l_[1] + m_[2]*l_[2] + (m_[3]*l_[3] + m_[4]*l_[4] + m_[5]*l_[5])*2.;
    wk[1][c] = m_[0]*m_[0] + m_[1]*m_[1] + m_[2]*m_[2] + (m_[3]*m_[3] + m_[4]*m_[4] + m_[5]*m_[5])*2.; }
  ave1d_channel_z(s, wk[0]); ave1d_channel_z(s, wk[1]);
  #pragma omp parallel for collapse(2) schedule(static) num_threads(s->nthreads)
  for (int k = 1; k <= n[2]; k++) for (int j = 1; j <= n[1]; j++) for (int i = 1; i <= n[0]; i++) { size_t c = IX(i,j,k);
    visct[c] = visct[c]*wk[0][c]/wk[1][c]; visct[c] = fmax(visct[c], 0.); }
}

/* ------------------------------------------------------------------ initial fields: src/initflow.f90:17-283 (deterministic kinds) */
int o_initflow(ostate *s, const char *inivel, int is_wallturb, double *u, double *v, double *w, double *p) {
  const int *n = s->n; const size_t s1 = s->s1, s2 = s->s2; const double *l = s->P.l, *dl = s->dl, *zc = s->zc, *zf = s->zf, *dzf = s->dzf;
  const double pi = PI, visc = s->visc; const double *bcvel = s->P.bcvel, *bforce = s->P.bforce;
  double uref = 1., ubulk = uref; int is_mean = 0, is3d = 0;
  if (s->P.is_forced[0]) ubulk = s->P.velf[0];
  int n3 = n[2]; double *u1d = dalloc(2*n3 + 2), *zz = dalloc(2*n3 + 2);
  #define BCV(side,dir,vel) bcvel[(side) + 2*((dir)-1) + 6*((vel)-1)]
  if (!strcmp(inivel, "cou")) { uref = BCV(0,3,1) - BCV(1,3,1);
    for (int k = 1; k <= n3; k++) { double z = zc[k]/l[2]; u1d[k] = .5*(1. - 2.*z)*uref; } uref = fabs(uref);
  } else if (!strcmp(inivel, "poi")) {
    for (int k = 1; k <= n3; k++) { double z = zc[k]/l[2]; u1d[k] = 6.*z*(1. - z)*ubulk; } is_mean = 1;
  } else if (!strcmp(inivel, "iop")) { ubulk = .5*fabs(BCV(0,3,1) + BCV(1,3,1));
    for (int k = 1; k <= n3; k++) { double z = zc[k]/l[2]; u1d[k] = 6.*z*(1. - z)*ubulk; u1d[k] = u1d[k] - ubulk; } is_mean = 1;
  } else if (!strcmp(inivel, "zer")) { for (int k = 1; k <= n3; k++) u1d[k] = 0.;
  } else if (!strcmp(inivel, "uni")) { for (int k = 1; k <= n3; k++) u1d[k] = uref;
  } else if (!strcmp(inivel, "hcp")) {      /* initflow.f90:93-102 */
    for (int k = 1; k <= n3; k++) { double z = zc[k]/(2*l[2]); u1d[k] = 6.*z*(1. - z)*ubulk; } is_mean = 1;
  } else if (!strcmp(inivel, "pdc") || !strcmp(inivel, "hdc")) { double lref = l[2]/2.;      /* initflow.f90:157-180 */
    if (strcmp(inivel, "pdc")) lref = 2.*lref;
    if (is_wallturb) { uref = pow(bforce[0]*lref, (double)0.5f); double retau = uref*lref/visc, reb = pow(retau/.09, 1./.88); ubulk = reb*visc/(2*lref); }
    else ubulk = (bforce[0]*(lref*lref)/(3.*visc));
    if (!strcmp(inivel, "pdc")) for (int k = 1; k <= n3; k++) { double z = zc[k]/l[2]; u1d[k] = 6.*z*(1. - z)*ubulk; }
    else for (int k = 1; k <= n3; k++) { double z = zc[k]/(2*l[2]); u1d[k] = 6.*z*(1. - z)*ubulk; }
    is_mean = 1;
  } else if (!strcmp(inivel, "tgv")) { is3d = 1;
    for (int k = 1; k <= n[2]; k++) { double zcc = zc[k]/l[2]*2.*pi;
      for (int j = 1; j <= n[1]; j++) { double yc = (j - .5)*dl[1]/l[1]*2.*pi, yf = (j - .0)*dl[1]/l[1]*2.*pi;
        for (int i = 1; i <= n[0]; i++) { double xc = (i - .5)*dl[0]/l[0]*2.*pi, xf = (i - .0)*dl[0]/l[0]*2.*pi; size_t c = IX(i,j,k);
          u[c] = sin(xf)*cos(yc)*cos(zcc)*uref; v[c] = -cos(xc)*sin(yf)*cos(zcc)*uref; w[c] = 0.; p[c] = 0.; } } }
  } else if (!strcmp(inivel, "tgw")) { is3d = 1;
    for (int k = 1; k <= n[2]; k++) for (int j = 1; j <= n[1]; j++) { double yc = (j - .5)*dl[1], yf = (j - .0)*dl[1];
      for (int i = 1; i <= n[0]; i++) { double xc = (i - .5)*dl[0], xf = (i - .0)*dl[0]; size_t c = IX(i,j,k);
        u[c] = cos(xf)*sin(yc)*uref; v[c] = -sin(xc)*cos(yf)*uref; w[c] = 0.; p[c] = -(cos(2.*xc) + cos(2.*yc))/4.*(uref*uref); } }
  } else if (!strcmp(inivel, "ant")) { is3d = 1;
    const double cf = (double)(4.f*sqrtf(2.f)/3.f/sqrtf(3.f));     /* default-real constant in the reference */
    for (int k = 1; k <= n[2]; k++) { double zcc = zc[k]/l[2]*2.*pi + 0.5*pi, zff = zf[k]/l[2]*2.*pi + 0.5*pi;
      for (int j = 1; j <= n[1]; j++) { double yc = (j - .5)*dl[1]/l[1]*2.*pi + 0.5*pi, yf = (j - .0)*dl[1]/l[1]*2.*pi + 0.5*pi;
        for (int i = 1; i <= n[0]; i++) { double xc = (i - .5)*dl[0]/l[0]*2.*pi + 0.5*pi, xf = (i - .0)*dl[0]/l[0]*2.*pi + 0.5*pi; size_t c = IX(i,j,k);
          u[c] = cf*(sin(xf-5.*pi/6.)*cos(yc-1.*pi/6.)*sin(zcc) - sin(xf-1.*pi/6.)*sin(yc)*cos(zcc-5.*pi/6.))*uref;
          v[c] = cf*(sin(xc)*sin(yf-5.*pi/6.)*sin(zcc-1.*pi/6.) - cos(xc-5.*pi/6.)*sin(yf-1.*pi/6.)*sin(zcc))*uref;
          w[c] = cf*(cos(xc-1.*pi/6.)*sin(yc)*sin(zff-5.*pi/6.) - sin(xc)*cos(yc-5.*pi/6.)*sin(zff-1.*pi/6.))*uref;
          p[c] = -(u[c]*u[c] + v[c]*v[c] + w[c]*w[c])/2.; } } }
  } else if (!strcmp(inivel, "duc")) { is3d = 1; is_mean = 1;
    for (int k = 1; k <= n[2]; k++) for (int j = 1; j <= n[1]; j++) {
      double sum_term = 0., ly = .5*l[1], lz = .5*l[2], xi = -1. + (j - 1.5 + 1.)*dl[1]/ly, eta = -1. + zc[k]/lz;
      xi = -1. + (j + 1 - 1.5)*dl[1]/ly;           /* (j+lo(2)-1.5_rp) with lo(2) = 1 */
      for (int m = 0; m <= 100; m++) {
        double cosh_term = cosh((2*m+1)*pi*ly/(2*lz)*xi)/cosh((2*m+1)*pi*ly/(2*lz));
        double cos_term = cos((2*m+1)*pi/2*eta);
        double den = (double)((2*m+1)*(2*m+1)*(2*m+1));
        double term = ((m & 1) ? -1. : 1.)/den*cosh_term*cos_term;
        sum_term = sum_term + term; }
      double tp = 2./pi; double val = .5*(lz*lz)*(1. - eta*eta - 4.*(tp*tp*tp)*sum_term);
      for (int i = 0; i <= n[0]+1; i++) { size_t c = IX(i,j,k); u[c] = val; v[c] = 0.; w[c] = 0.; p[c] = 0.; } }
  } else { free(u1d); free(zz); return 1; }
  #undef BCV
  if (!is3d) for (int k = 1; k <= n[2]; k++) for (int j = 1; j <= n[1]; j++) for (int i = 1; i <= n[0]; i++) {
    size_t c = IX(i,j,k); u[c] = u1d[k]; v[c] = 0.; w[c] = 0.; p[c] = 0.; }
  if (is_mean && strcmp(inivel, "iop")) {          /* set_mean, initflow.f90:317-338 */
    double meanold = 0.;
    for (int k = 1; k <= n[2]; k++) { double g = dzf[k]/l[2]*(dl[0]/l[0])*(dl[1]/l[1]);
      for (int j = 1; j <= n[1]; j++) for (int i = 1; i <= n[0]; i++) meanold = meanold + u[IX(i,j,k)]*g; }
    if (meanold != 0.) for (int k = 1; k <= n[2]; k++) for (int j = 1; j <= n[1]; j++) for (int i = 1; i <= n[0]; i++)
      u[IX(i,j,k)] = u[IX(i,j,k)]/meanold*ubulk;
  }
  if (is_wallturb) {                               /* vortex pair, initflow.f90:218-246 */
    for (int k = 1; k <= n[2]; k++) { double zcc = 2.*zc[k]/l[2] - 1., zff = 2.*(zc[k]/l[2] + .5*dzf[k]/l[2]) - 1.;
      for (int j = 1; j <= n[1]; j++) { double yc = ((j - 0.5)*dl[1] - .5*l[1])*2./l[2], yf = ((j - 0.0)*dl[1] - .5*l[1])*2./l[2];
        for (int i = 1; i <= n[0]; i++) { double xc = ((i - 0.5)*dl[0] - .5*l[0])*2./l[2]; size_t c = IX(i,j,k);
          double gxy = xc*exp(-4.*(4.*(yf*yf) + xc*xc));                 /* gxy(yf,xc) */
          double dfz = -4.*zcc*(1. - zcc*zcc);
          double fz = (1. - zff*zff)*(1. - zff*zff);
          double dgxy = exp(-4.*(4.*(yc*yc) + xc*xc))*(1. - 8.*(xc*xc));  /* dgxy(yc,xc) */
          v[c] = -1.*gxy*dfz*ubulk*1.5; w[c] = 1.*fz*dgxy*ubulk*1.5; p[c] = 0.; } } }
  }
  free(u1d); free(zz);
  return 0;
}

/* ------------------------------------------------------------------ full step: src/main.f90:417-508 */
void o_step(ostate *s, double dt, double *u, double *v, double *w, double *p, double *pp, double *visct, double *dpdl) {
  double f[3];
  dpdl[0] = dpdl[1] = dpdl[2] = 0.;
  for (int irk = 1; irk <= 3; irk++) {
    double dtrk = (RKCOEFF[irk-1][0] + RKCOEFF[irk-1][1])*dt, dtrki = 1./dtrk, alpha = 0.;
    o_rk(s, irk, dt, p, visct, u, v, w, f);
    o_bulk_forcing(s, f, u, v, w);
    if (s->P.impdiff == 2) {
      alpha = -.5*s->visc*dtrk;
      double *q[3] = {u, v, w};
      for (int iv = 1; iv <= 3; iv++) { o_updt_rhs_b_velz(s, iv, alpha, q[iv-1]); o_solver_gaussel_z(s, iv, alpha, q[iv-1]); }
    } else if (s->P.impdiff == 1) {      /* x and y periodic only: their boundary r.h.s. planes vanish */
      alpha = -.5*s->visc*dtrk;
      double *q[3] = {u, v, w};
      for (int iv = 1; iv <= 3; iv++) { o_updt_rhs_b_vel(s, iv, alpha, q[iv-1]); o_solver_helmholtz(s, iv, alpha, q[iv-1]); }
    }
    for (int c = 0; c < 3; c++) dpdl[c] = dpdl[c] + f[c];
    o_bounduvw(s, 1, 0, u, v, w);
    o_fillps(s, dtrki, u, v, w, pp);
    o_updt_rhs_b_p(s, pp);
    o_solver(s, pp);
    o_boundp(s, 0, pp);
    o_correc(s, dtrk, pp, u, v, w);
    o_bounduvw(s, 1, 1, u, v, w);
    o_updatep(s, alpha, pp, p);
    o_boundp(s, 0, p);
    o_cmpt_sgs(s, u, v, w, visct);
    o_boundp(s, 1, visct);
  }
  double dti = 1./dt;
  for (int c = 0; c < 3; c++) dpdl[c] = -dpdl[c]*dti;
}
