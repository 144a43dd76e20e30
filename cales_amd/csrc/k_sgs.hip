// Eddy-viscosity kernels: cmpt_sgs (reference src/sgs.f90:21-386) for 'smag' (van Driest damped) and
// 'dsmag' (dynamic, plane-averaged: the reference hard-wires _CHANNEL, sgs.f90:8,362-364), with
// strain_rate (sgs.f90:1019-1110), extrapolate (682-767), filter3d (616-680), interpolate (850-870),
// ave1d_channel (433-482) and cmpt_alph2 (769-822).
#include "common.hpp"
#include <type_traits>

#define BX 64
#define BY 4

// ------------------------------------------------------------------------------------------ extrapolate
struct ExJob { real *p; int idir, ibound; real factor; real *save; int restore; };   // save: plane buffer for the overwritten ghosts
struct ExJobs { int njobs; ExJob job[18]; };
__global__ __launch_bounds__(256) void k_extrapolate(Geom g, ExJobs J) {
  const ExJob jb = J.job[blockIdx.z];
  const int idir = jb.idir;
  const int na = idir == 1 ? g.n2 : g.n1, nb = idir == 3 ? g.n2 : g.n3, n = idir == 1 ? g.n1 : idir == 2 ? g.n2 : g.n3;
  const int a = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y * 4 + threadIdx.y;
  if (a > na + 1 || b > nb + 1) return;
  const long st = idir == 1 ? 1 : idir == 2 ? (long)g.s1 : g.s12;
  real *p = jb.p + (idir == 1 ? g.ix(0, a, b) : idir == 2 ? g.ix(a, 0, b) : g.ix(a, b, 0));
#define P(m) p[(long)(m)*st]
  const real f = jb.factor;
  const int gh = jb.ibound == 0 ? 0 : n + 1;
  if (jb.save) {                      // in-place use on a live field: keep / give back the ghost value
    const size_t q = (size_t)a + (size_t)(na + 2) * b;
    if (jb.restore) { P(gh) = jb.save[q]; return; }
    jb.save[q] = P(gh);
  }
  if (idir < 3) { if (jb.ibound == 0) P(0) = 2. * P(1) - P(2); else P(n + 1) = 2. * P(n) - P(n - 1); }
  else { if (jb.ibound == 0) P(0) = (1. + f) * P(1) - f * P(2); else P(n + 1) = (1. + f) * P(n) - f * P(n - 1); }
#undef P
}
// mode 0: `lwm` form (wall-model faces, z factors from the grid), 1: `cbc` form (all no-slip walls, factor 1)
// save != nullptr: the fields are live (u,v,w themselves): the ghosts they lose are kept in `save` (restore = 0) and given back
// afterwards (restore = 1, directions in reverse order so that edge cells see the same sequence backwards)
static int extrapolate(cales_ctx *c, int nf, real **p, const int *iface, int mode, real *save = nullptr, int restore = 0) {
  const int *n = c->n;
  // save slots: [direction][field][side], each the size of that direction's ghost plane
  const size_t psz[3] = {(size_t)(n[1] + 2) * (n[2] + 2), (size_t)(n[0] + 2) * (n[2] + 2), (size_t)(n[0] + 2) * (n[1] + 2)};
  const size_t poff[3] = {0, 6 * psz[0], 6 * (psz[0] + psz[1])};
  for (int step = 1; step <= 3; ++step) {      // one launch per direction, order x,y,z as the reference
    const int idir = restore ? 4 - step : step;
    ExJobs J; J.njobs = 0;
    for (int q = 0; q < nf; ++q) for (int ib = 0; ib <= 1; ++ib) {
      bool done;
      real factor = 1.;
      if (mode == 1) done = ISB(c, ib, idir) && CBV(c, ib, idir, idir) == 'D' && iface[q] != idir;
      else {
        done = ISB(c, ib, idir) && LWM(c, ib, idir) != 0 && iface[q] != idir;
        factor = ib == 0 ? (1. / c->dzci[0]) * c->dzci[1] : (1. / c->dzci[n[2]]) * c->dzci[n[2] - 1];
      }
      if (!done) continue;
      ExJob &j = J.job[J.njobs++]; j.p = p[q]; j.idir = idir; j.ibound = ib; j.factor = factor; j.save = nullptr; j.restore = restore;
      if (save) j.save = save + poff[idir - 1] + (size_t)(q * 2 + ib) * psz[idir - 1];
    }
    if (!J.njobs) continue;
    const int na = idir == 1 ? n[1] : n[0], nb = idir == 3 ? n[1] : n[2];
    LAUNCH(c, k_extrapolate, dim3((na + 2 + 63) / 64, (nb + 2 + 3) / 4, J.njobs), dim3(64, 4), 0, c->stream, c->g, J);
  }
  LAUNCHCHK(c);
  return 0;
}

// ------------------------------------------------------------------------------------------ strain_rate
template <int WITH_SIJ>
__global__ __launch_bounds__(BX *BY) void k_strain(Geom g, real dxi, real dyi, const real *__restrict__ dzci,
                                                    const real *__restrict__ dzfi, const real *__restrict__ u,
                                                    const real *__restrict__ v, const real *__restrict__ w, real *__restrict__ s0,
                                                    real *__restrict__ s11o, real *__restrict__ s22o, real *__restrict__ s33o,
                                                    real *__restrict__ s12o, real *__restrict__ s13o, real *__restrict__ s23o) {
  int bx_, by_, bz_; stencil_block(bx_, by_, bz_);
  const int i = bx_ * BX + threadIdx.x + 1, j = by_ * BY + threadIdx.y + 1, k = bz_ + 1;
  if (i > g.n1 || j > g.n2) return;
  const size_t c = g.ix(i, j, k);
  const long sj = g.s1, sk = g.s12;
#define LD(a, di, dj, dk) a[c + (di) + (dj)*sj + (dk)*sk]
  const real u_mcm = LD(u, -1, 0, -1), u_ccm = LD(u, 0, 0, -1), u_mmc = LD(u, -1, -1, 0), u_cmc = LD(u, 0, -1, 0), u_mcc = LD(u, -1, 0, 0),
               u_ccc = LD(u, 0, 0, 0), u_mpc = LD(u, -1, 1, 0), u_cpc = LD(u, 0, 1, 0), u_mcp = LD(u, -1, 0, 1), u_ccp = LD(u, 0, 0, 1);
  const real v_cmm = LD(v, 0, -1, -1), v_ccm = LD(v, 0, 0, -1), v_mmc = LD(v, -1, -1, 0), v_cmc = LD(v, 0, -1, 0), v_pmc = LD(v, 1, -1, 0),
               v_mcc = LD(v, -1, 0, 0), v_ccc = LD(v, 0, 0, 0), v_pcc = LD(v, 1, 0, 0), v_cmp = LD(v, 0, -1, 1), v_ccp = LD(v, 0, 0, 1);
  const real w_cmm = LD(w, 0, -1, -1), w_mcm = LD(w, -1, 0, -1), w_ccm = LD(w, 0, 0, -1), w_pcm = LD(w, 1, 0, -1), w_cpm = LD(w, 0, 1, -1),
               w_cmc = LD(w, 0, -1, 0), w_mcc = LD(w, -1, 0, 0), w_ccc = LD(w, 0, 0, 0), w_pcc = LD(w, 1, 0, 0), w_cpc = LD(w, 0, 1, 0);
#undef LD
  const real zc = dzci[k], zm = dzci[k - 1];
  const real s11 = (u_ccc - u_mcc) * dxi, s22 = (v_ccc - v_cmc) * dyi, s33 = (w_ccc - w_ccm) * dzfi[k];
  const real s12 = .125 * ((u_cpc - u_ccc) * dyi + (v_pcc - v_ccc) * dxi + (u_ccc - u_cmc) * dyi + (v_pmc - v_cmc) * dxi +
                             (u_mpc - u_mcc) * dyi + (v_ccc - v_mcc) * dxi + (u_mcc - u_mmc) * dyi + (v_cmc - v_mmc) * dxi);
  const real s13 = .125 * ((u_ccp - u_ccc) * zc + (w_pcc - w_ccc) * dxi + (u_ccc - u_ccm) * zm + (w_pcm - w_ccm) * dxi +
                             (u_mcp - u_mcc) * zc + (w_ccc - w_mcc) * dxi + (u_mcc - u_mcm) * zm + (w_ccm - w_mcm) * dxi);
  const real s23 = .125 * ((v_ccp - v_ccc) * zc + (w_cpc - w_ccc) * dyi + (v_ccc - v_ccm) * zm + (w_cpm - w_ccm) * dyi +
                             (v_cmp - v_cmc) * zc + (w_ccc - w_cmc) * dyi + (v_cmc - v_cmm) * zm + (w_ccm - w_cmm) * dyi);
  const real s0v = sqrt(2. * (s11 * s11 + s22 * s22 + s33 * s33 + 2. * (s12 * s12 + s13 * s13 + s23 * s23)));
  s0[c] = s0v;
  if (WITH_SIJ) { s11o[c] = s11; s22o[c] = s22; s33o[c] = s33; s12o[c] = s12; s13o[c] = s13; s23o[c] = s23; }
}
static int strain_rate(cales_ctx *c, const real *u, const real *v, const real *w, real *s0, real **sij) {
  ProfScope ps(c, "strain_rate");
  dim3 b(BX, BY, 1), gr = grid3(c->n[0], c->n[1], c->n[2], b);
  if (sij) LAUNCH(c, k_strain<1>, gr, b, 0, c->stream, c->g, c->dli[0], c->dli[1], c->d_dzci, c->d_dzfi, u, v, w, s0, sij[0], sij[1], sij[2], sij[3], sij[4], sij[5]);
  else LAUNCH(c, k_strain<0>, gr, b, 0, c->stream, c->g, c->dli[0], c->dli[1], c->d_dzci, c->d_dzfi, u, v, w, s0, (real *)nullptr,
                          (real *)nullptr, (real *)nullptr, (real *)nullptr, (real *)nullptr, (real *)nullptr);
  LAUNCHCHK(c);
  return 0;
}

// ------------------------------------------------------------------------------------------ static Smagorinsky (sgs.f90:98-152)
struct SmagArgs { real w0, w1, w2, w3, w4, w5, dl1, dl2, l3, dxi, dyi, visc, sumw; };
__global__ __launch_bounds__(BX *BY) void k_smag(Geom g, SmagArgs A, const real *__restrict__ zc, const real *__restrict__ dzci,
                                                  const real *__restrict__ dzf, const real *__restrict__ u, const real *__restrict__ v,
                                                  const real *__restrict__ w, const real *__restrict__ s0, real *__restrict__ visct,
                                                  const real *__restrict__ twy) {      // twy != null (several slabs): sqrt(tau_w) planes of the two y walls, see wall_shear_y_planes
  int bx_, by_, bz_; stencil_block(bx_, by_, bz_);
  const int i = bx_ * BX + threadIdx.x + 1, j = by_ * BY + threadIdx.y + 1, k = bz_ + 1;
  if (i > g.n1 || j > g.n2) return;
  real fd;
  if (A.sumw == 0.) fd = 1.;
  else {
    const int jg = j + g.jlo;                    // global row: distances to the y walls use global indices
    real dw[6];
    dw[0] = A.dl1 * (i - 0.5); dw[1] = A.dl1 * (g.n1 - i + 0.5);
    dw[2] = A.dl2 * (jg - 0.5); dw[3] = A.dl2 * (g.ng2 - jg + 0.5);
    dw[4] = zc[k]; dw[5] = A.l3 - zc[k];
    const real isw[6] = {A.w0, A.w1, A.w2, A.w3, A.w4, A.w5};
    int loc = 0;
#pragma unroll
    for (int q = 0; q < 6; ++q) dw[q] = dw[q] * isw[q] + CALES_BIG * (1. - isw[q]);
#pragma unroll
    for (int q = 1; q < 6; ++q) if (dw[q] < dw[loc]) loc = q;
    const real dw_min = dw[loc];
    real t1 = 0., t2 = 0., sc = 0.;
    const int n1 = g.n1, n2 = g.n2, n3 = g.n3, im1 = i - 1;
    if (twy && (loc == 2 || loc == 3)) {      // the y walls belong to the first and the last slab: their shear arrives as planes (same arithmetic, k_wall_shear_y)
      const real dw_plus = dw_min * twy[(size_t)(loc == 2 ? k : n3 + 2 + k) * g.s1 + i] * (1. / A.visc);
      fd = 1. - exp(-dw_plus / 25.);
    } else {
    switch (loc) {
    case 0: t1 = v[g.ix(1, j, k)] - v[g.ix(0, j, k)] + v[g.ix(1, j - 1, k)] - v[g.ix(0, j - 1, k)];
            t2 = w[g.ix(1, j, k)] - w[g.ix(0, j, k)] + w[g.ix(1, j, k - 1)] - w[g.ix(0, j, k - 1)]; sc = A.dxi; break;
    case 1: t1 = v[g.ix(n1, j, k)] - v[g.ix(n1 + 1, j, k)] + v[g.ix(n1, j - 1, k)] - v[g.ix(n1 + 1, j - 1, k)];
            t2 = w[g.ix(n1, j, k)] - w[g.ix(n1 + 1, j, k)] + w[g.ix(n1, j, k - 1)] - w[g.ix(n1 + 1, j, k - 1)]; sc = A.dxi; break;
    case 2: t1 = u[g.ix(i, 1, k)] - u[g.ix(i, 0, k)] + u[g.ix(i - 1, 1, k)] - u[g.ix(i - 1, 0, k)];
            t2 = w[g.ix(i, 1, k)] - w[g.ix(i, 0, k)] + w[g.ix(i, 1, k - 1)] - w[g.ix(i, 0, k - 1)]; sc = A.dyi; break;
    case 3: t1 = u[g.ix(i, n2, k)] - u[g.ix(i, n2 + 1, k)] + u[g.ix(i - 1, n2, k)] - u[g.ix(i - 1, n2 + 1, k)];
            t2 = w[g.ix(i, n2, k)] - w[g.ix(i, n2 + 1, k)] + w[g.ix(i, n2, k - 1)] - w[g.ix(i, n2 + 1, k - 1)]; sc = A.dyi; break;
    case 4: t1 = u[g.ix(i, j, 1)] - u[g.ix(i, j, 0)] + u[g.ix(im1, j, 1)] - u[g.ix(im1, j, 0)];
            t2 = v[g.ix(i, j, 1)] - v[g.ix(i, j, 0)] + v[g.ix(i, j - 1, 1)] - v[g.ix(i, j - 1, 0)]; sc = dzci[0]; break;
    default: t1 = u[g.ix(i, j, n3)] - u[g.ix(i, j, n3 + 1)] + u[g.ix(im1, j, n3)] - u[g.ix(im1, j, n3 + 1)];
             t2 = v[g.ix(i, j, n3)] - v[g.ix(i, j, n3 + 1)] + v[g.ix(i, j - 1, n3)] - v[g.ix(i, j - 1, n3 + 1)]; sc = dzci[n3]; break;
    }
    real tauw_s = sqrt(t1 * t1 + t2 * t2) * sc;
    tauw_s = 0.5 * A.visc * tauw_s;
    const real dw_plus = dw_min * sqrt(tauw_s) * (1. / A.visc);
    fd = 1. - exp(-dw_plus / 25.);
    }
  }
  const real t = 0.11 * dzf[k] * fd;        // dzf[] here is the table (dx dy dzf(k))^(1/3) of k_smag_del (sgs.f90:145); c_smag, src/param.f90:33
  const size_t c = g.ix(i, j, k);
  visct[c] = (t * t) * s0[c];
}

// ------------------------------------------------------------------------------------------ small dsmag kernels
__global__ __launch_bounds__(256) void k_copy3(size_t n, const real *__restrict__ a, const real *__restrict__ b, const real *__restrict__ c_,
                                               real *__restrict__ x, real *__restrict__ y, real *__restrict__ z) {
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (size_t)gridDim.x * 256) { x[q] = a[q]; y[q] = b[q]; z[q] = c_[q]; }
}
__global__ __launch_bounds__(256) void k_copy1(size_t n, const real *__restrict__ a, real *__restrict__ x) {
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (size_t)gridDim.x * 256) x[q] = a[q];
}
struct P6 { real *p[6]; };
struct CP6 { const real *p[6]; };
__global__ __launch_bounds__(256) void k_s0sij(size_t n, const real *__restrict__ s0, CP6 sij, P6 wk) {   // sgs.f90:198-210 (all cells)
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (size_t)gridDim.x * 256) {
    const real s = s0[q];
#pragma unroll
    for (int m = 0; m < 6; ++m) wk.p[m][q] = s * sij.p[m][q];
  }
}
__global__ __launch_bounds__(256) void k_uiuj(size_t n, const real *__restrict__ uc, const real *__restrict__ vc, const real *__restrict__ wc, P6 wk) {
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (size_t)gridDim.x * 256) {   // sgs.f90:283-295
    const real a = uc[q], b = vc[q], c_ = wc[q];
    wk.p[0][q] = a * a; wk.p[1][q] = b * b; wk.p[2][q] = c_ * c_; wk.p[3][q] = a * b; wk.p[4][q] = a * c_; wk.p[5][q] = b * c_;
  }
}
// filter3d (sgs.f90:632-679): 27-point top-hat, weights 8/4/2/1 over 64
__global__ __launch_bounds__(BX *BY) void k_filter3d(Geom g, const real *__restrict__ p, real *__restrict__ pf) {
  int bx_, by_, bz_; stencil_block(bx_, by_, bz_);
  const int i = bx_ * BX + threadIdx.x + 1, j = by_ * BY + threadIdx.y + 1, k = bz_ + 1;
  if (i > g.n1 || j > g.n2) return;
  const size_t c = g.ix(i, j, k);
  const long sj = g.s1, sk = g.s12;
#define Q(di, dj, dk) p[c + (di) + (dj)*sj + (dk)*sk]
  pf[c] = (8. * (Q(0, 0, 0)) + 4. * (Q(-1, 0, 0) + Q(0, -1, 0) + Q(0, 0, -1) + Q(1, 0, 0) + Q(0, 1, 0) + Q(0, 0, 1)) +
           2. * (Q(0, -1, -1) + Q(-1, 0, -1) + Q(-1, -1, 0) + Q(0, 1, -1) + Q(1, 0, -1) + Q(1, -1, 0) + Q(0, -1, 1) + Q(-1, 0, 1) +
                 Q(-1, 1, 0) + Q(0, 1, 1) + Q(1, 0, 1) + Q(1, 1, 0)) +
           1. * (Q(-1, -1, -1) + Q(1, -1, -1) + Q(-1, 1, -1) + Q(1, 1, -1) + Q(-1, -1, 1) + Q(1, -1, 1) + Q(-1, 1, 1) + Q(1, 1, 1))) / 64.;
#undef Q
}
__global__ __launch_bounds__(BX *BY) void k_mij(Geom g, P6 mij, const real *__restrict__ alph2, const real *__restrict__ s0, CP6 sij) {
  const int i = blockIdx.x * BX + threadIdx.x + 1, j = blockIdx.y * BY + threadIdx.y + 1, k = blockIdx.z + 1;   // sgs.f90:262-272
  if (i > g.n1 || j > g.n2) return;
  const size_t c = g.ix(i, j, k);
  const real a = alph2[c], s = s0[c];
#pragma unroll
  for (int m = 0; m < 6; ++m) mij.p[m][c] = 2. * (mij.p[m][c] - a * s * sij.p[m][c]);
}
__global__ __launch_bounds__(BX *BY) void k_interp(Geom g, const real *__restrict__ u, const real *__restrict__ v, const real *__restrict__ w,
                                                    real *__restrict__ uc, real *__restrict__ vc, real *__restrict__ wc) {
  const int i = blockIdx.x * BX + threadIdx.x + 1, j = blockIdx.y * BY + threadIdx.y + 1, k = blockIdx.z + 1;   // sgs.f90:860-869
  if (i > g.n1 || j > g.n2) return;
  const size_t c = g.ix(i, j, k);
  uc[c] = 0.5 * (u[c] + u[c - 1]); vc[c] = 0.5 * (v[c] + v[c - g.s1]); wc[c] = 0.5 * (w[c] + w[c - g.s12]);
}
__global__ __launch_bounds__(BX *BY) void k_contract(Geom g, CP6 mij, CP6 lij, const real *__restrict__ uf, const real *__restrict__ vf,
                                                      const real *__restrict__ wf, real *__restrict__ lm, real *__restrict__ mm) {
  const int i = blockIdx.x * BX + threadIdx.x + 1, j = blockIdx.y * BY + threadIdx.y + 1, k = blockIdx.z + 1;   // sgs.f90:328-358
  if (i > g.n1 || j > g.n2) return;
  const size_t c = g.ix(i, j, k);
  real m_[6], l_[6];
#pragma unroll
  for (int m = 0; m < 6; ++m) { m_[m] = mij.p[m][c]; l_[m] = lij.p[m][c]; }
  const real a = uf[c], b = vf[c], d = wf[c];
  l_[0] -= a * a; l_[1] -= b * b; l_[2] -= d * d; l_[3] -= a * b; l_[4] -= a * d; l_[5] -= b * d;
  lm[c] = m_[0] * l_[0] + m_[1] * l_[1] + m_[2] * l_[2] + (m_[3] * l_[3] + m_[4] * l_[4] + m_[5] * l_[5]) * 2.;
  mm[c] = m_[0] * m_[0] + m_[1] * m_[1] + m_[2] * m_[2] + (m_[3] * m_[3] + m_[4] * m_[4] + m_[5] * m_[5]) * 2.;
}
// ave1d_channel (sgs.f90:462-480): plane sums of two fields -> p1d(2, n3); one block per (k, field)
__global__ __launch_bounds__(256) void k_plane_sum(Geom g, const real *__restrict__ a, const real *__restrict__ b, real *__restrict__ p1d) {
  __shared__ real sh[4];
  const int k = blockIdx.x + 1; const real *p = blockIdx.y ? b : a;
  real acc = 0.;
  const long np = (long)g.n1 * g.n2;
  for (long q = threadIdx.x; q < np; q += 256) acc += p[g.ix((int)(q % g.n1) + 1, (int)(q / g.n1) + 1, k)];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) p1d[blockIdx.y * g.n3 + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
// visct = max(visct*<LM>/<MM>,0) (sgs.f90:372-380); the plane averages replace the broadcast arrays
__global__ __launch_bounds__(BX *BY) void k_dsmag_final(Geom g, real gar, const real *__restrict__ p1d, const real *s0, real *visct) {
  const int i = blockIdx.x * BX + threadIdx.x + 1, j = blockIdx.y * BY + threadIdx.y + 1, k = blockIdx.z + 1;
  if (i > g.n1 || j > g.n2) return;
  const size_t c = g.ix(i, j, k);
  const real lm = p1d[k - 1] * gar, mm = p1d[g.n3 + k - 1] * gar;
  real vt = s0[c] * lm / mm;
  visct[c] = fmax(vt, 0.);
}
// plane coefficients of the lazy form: cs(k) = max(<LM>/<MM>, 0) for k = 1..n3 (NaN -> 0 as fmax does in k_dsmag_final), ghost planes from
// the z boundary type of the field (periodic: wrap; wall: the neighbour's, the ghost value of |S| carries the sign)
__global__ void k_dsmag_coef(int n3, real gar, const real *__restrict__ p1d, real *__restrict__ cs, int periodic_z) {
  for (int k = threadIdx.x + 1; k <= n3; k += blockDim.x) { const real lm = p1d[k - 1] * gar, mm = p1d[n3 + k - 1] * gar; cs[k] = fmax(lm / mm, 0.); }
  __syncthreads();
  if (threadIdx.x == 0) { cs[0] = periodic_z ? cs[n3] : cs[1]; cs[n3 + 1] = periodic_z ? cs[1] : cs[n3]; }
}
__global__ __launch_bounds__(256) void k_scale_planes(Geom g, const real *__restrict__ cs, real *__restrict__ f) {
  const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
  if (i > g.n1 + 1 || j > g.n2 + 1) return;
  const size_t c = g.ix(i, j, k);
  f[c] = f[c] * cs[k];
}
int materialize_visct(cales_ctx *c) {
  if (!c->visct_lazy) return 0;
  c->visct_lazy = false;
  const int *n = c->n;
  LAUNCH(c, k_scale_planes, dim3((n[0] + 2 + 63) / 64, (n[1] + 2 + 3) / 4, n[2] + 2), dim3(64, 4, 1), 0, c->stream, c->g, c->d_cs, c->f[CALES_VISCT]);
  LAUNCHCHK(c);
  return 0;
}
__global__ __launch_bounds__(256) void k_alph2(Geom g, real w0, real w1, real w2, real w3, real w4, real w5, real *__restrict__ alph2) {
  const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;    // sgs.f90:783-816
  if (i > g.n1 + 1 || j > g.n2 + 1) return;
  const int jg = j + g.jlo;
  const bool near = (w0 != 0. && i == 1) || (w1 != 0. && i == g.n1) || (w2 != 0. && jg == 1) || (w3 != 0. && jg == g.ng2) ||
                    (w4 != 0. && k == 1) || (w5 != 0. && k == g.n3);
  alph2[g.ix(i, j, k)] = near ? 2.52 : 4.00;
}

// ================================================================================================
// Fast path of the dynamic model for cases whose only walls are in z (channels) and without wall model.
// Same arithmetic as the reference sequence (sgs.f90:153-380) re-associated for the hardware, three passes:
//  K_AC  u,v,w -> |S|, |S|Sij, cell-centred velocity, test-filtered velocity        (k_strain_tile / k_corr_strain_tile)
//  K_B + K_DF  filter(|S|Sij) on the fly, filtered velocity, cell-centred velocity -> plane sums of Mij Lij and Mij Mij (k_lmf_tile; the
//        two-pass form k_filter6_tile + k_lij_mij_tile of rounds 1-2 went in round 5)
//  then <LM>/<MM> per plane and visct = max(|S| <LM>/<MM>, 0).
//  * the 27-point top-hat (sgs.f90:632-679) is separable, (1,2,1)^3/64, combined z first, then x, then y;
//  * the wall extrapolation of filtered fields (extrapolate(...,cbc), sgs.f90:705-710,751-766, factor 1) is the rule
//    Q(0) = 2 Q(1) - Q(2) applied to the z combination (filter and extrapolation are linear and commute);
//  * sij, Mij, Lij, LM, MM and the products ui uj are never stored.
// Traffic: ~45 words/cell instead of ~166 measured for the kernel-per-loop sequence (profiles/r01a_*).
// ================================================================================================
// Tile kernels: blocks of 64 x (TY+2) threads march in k; every thread loads its own cell of each input per plane (full
// coalesced rows, next plane prefetched), x neighbours come from DPP lane moves or LDS rows, y neighbours through LDS. The
// kernels that write fields (K_AC, K_B) use tiles of 64 x TY outputs starting at i = 1 + 64 bx so that rows are read and
// written as whole 128-B lines; the read-only K_DF uses 62 x TY outputs with the x halo inside the wave. With the wall rule
// Q(0) = 2Q(1)-Q(2) ghost planes of extrapolated quantities are never read.
__device__ inline void uiuj(const real *s, real *q) {
  q[0] = s[0]; q[1] = s[1]; q[2] = s[2]; q[3] = s[0] * s[0]; q[4] = s[1] * s[1]; q[5] = s[2] * s[2];
  q[6] = s[0] * s[1]; q[7] = s[0] * s[2]; q[8] = s[1] * s[2];
}
struct LijMijArgs {
  const real *uc[3], *uf[3], *mf[6];
  real *part;
  const real *dzci, *dzfi;
  real dxi, dyi;
  int kchunk, nblk, zlo, zhi;
  int wmlo, wmhi; real flo, fhi;      // wall-model z faces: ghost planes of uf,vf by extrapolate(...,lwm), sgs.f90:683-748
  int perx;                             // x ghost columns of uc.., uf.. are not stored: wrap around
  // y walls owned by this rank (k_lmf_tile only): wall rule of the filters along y, alph2 of the wall rows; wall-model y faces: ghost rows of u_f, w_f
  int wylo, wyhi, wmylo, wmyhi;
  // k_lmf_tile<.., UCF = 1>: uc[] are the velocities u, v, w themselves and the cell-centred velocity (sgs.f90:860-869) is formed while loading;
  // vcg = the field whose ghost row 0 holds v_c of the row below the slab (periodic copy or the neighbour's), the one value v(-1) would be needed for
  const real *vcg; int perz, xwrap;      // xwrap: inside cales_step with stale x ghost columns, u(0) is read as u(n1)
};
// K_B + K_D + K_F in one pass (the default): as k_lij_mij_tile, but the test filter of the six products |S|Sij is formed on the fly
// instead of being read back from a pass of its own -- 12 words per cell less (k_filter6_tile's 6 in + 6 out), 12 in and nothing
// but partial sums out. The kernel is bound by its vector instructions (profiles/r02a_sq.md: k_lij_mij_tile VALU busy 0.61, LDS 0.31),
// not by LDS or HBM, and LDS has no room for six more double-buffered filter sums, so the y combination of |S|Sij uses neither:
// every output row loads its own row and the two rows beside it (the neighbours' rows are L1/L2 hits: the same block loads them in
// the same iteration) and combines them in registers; x by DPP, z from two rolling x/y-combined planes. Tiles of 62 x TYL outputs
// with TYL = 8: ten waves per block need <= 168 VGPRs, which holds (without spills in the plane loop) because the 18 loads are issued
// after the register-hungry strain-rate part and folded into six values right after the next barrier; 62 x 6 tiles with eight waves
// measured 4.47 ms against 4.09, variants that spill in the loop 13 ms (profiles/r02d_sq.md: VALU 0.44, waves waiting 0.50).
#ifndef TYL
#define TYL 8
#endif
#ifndef TYLF
#define TYLF 8      // tile height of the instantiations without y walls (their LDS leaves room for twelve rows: six filtered quantities instead of nine)
#endif
#define LMF_TY(YW) ((YW) ? TYL : TYLF)
struct LmfArgs { LijMijArgs L; const real *ss[6]; const real2 *ss2[3]; int by0; BandMap bm; int gx; };      // ss2: |S|Sij as three fields of pairs (PAIR = 1), ss: six fields      // bm: block map of this launch (bm.gx = 0: plain 3-D grid); gx: x tiles of the whole field      // by0: first y tile of this launch (interior and edge tiles of a slab are launched apart)
// Every global access of the plane loop is UNCONDITIONAL (out-of-range lanes, rows and planes are clamped onto valid cells whose values are
// never used): a load inside a divergent branch makes the compiler wait with s_waitcnt vmcnt(0) at the next use of ANY loaded value -- it
// cannot count the operations in flight across the branch -- and that drained, right behind the barrier of every plane, the six loads of
// plane k+2 issued a hundred instructions earlier (HBM latency, once per plane, for all ten waves). With straight-line loads the waits are
// counted: the 18 |S|Sij loads are waited for with the 6 velocity loads of the next plane still in flight, and those with the next 18.
// For the same reason the block sums of a chunk's planes are collected in LDS and stored after the loop (a store inside the loop is a
// vector-memory operation in a branch as well), and the per-plane grid coefficients come through the scalar cache one plane ahead.
constexpr int LMF_KMAX = 256;      // longest k chunk (block sums of a chunk in LDS: 2 x 256 x 8 B of the 9.9 KB the tile leaves)
// UCF = 1: the cell-centred velocity u_c = (u(i) + u(i-1))/2, v_c = (v(j) + v(j-1))/2, w_c = (w(k) + w(k-1))/2 (sgs.f90:860-869, the same expressions
// K_AC stored) is formed here from u, v, w: five loads per plane instead of three, and K_AC writes three fields less (3 of its 16 words; the pass is
// bound by its writes) -- x periodic, z walls or periodic, y periodic / slab neighbours / walls; otherwise UCF = 0 reads the stored fields.
// PAIR = 1: K_AC stored |S|Sij as three fields of PAIRS (S11,S22), (S33,S12), (S13,S23) per cell: nine 16-byte loads per plane instead of eighteen
// 8-byte ones. The pass is bound by what its ten waves issue, not by bytes: 3.95 -> 3.45 ms at 512^3 (K_AC's paired stores cost 0.28 of the 0.5 back).
template <typename OFF, int YW, int UCF, int PAIR = 0>      // YW = 1: walls or wall-model faces in y (ducts); 0: the channel instantiation carries none of that logic
__global__ __launch_bounds__(64 * (LMF_TY(YW) + 2)) void k_lmf_tile(Geom g, LmfArgs B) {
  const LijMijArgs &A = B.L;
  constexpr int TL = LMF_TY(YW);      // tile height
  // FUC (no walls in y): the test-filtered CELL-CENTRED velocity is not filtered here -- the filter is linear and shift-invariant, so F(u_c) =
  // F((u(i) + u(i-1))/2) = (u_f(i) + u_f(i-1))/2, and the filtered face velocity u_f (K_AC's output) is in this pass's LDS ring anyway for the strain rate
  // of the filtered field; likewise v_f along y and w_f along z. Three of the nine quantities leave the z/x/y filter (a third of its LDS traffic, 12 %
  // of the pass's vector instructions). The identity holds wherever the two filters treat the walls alike: everywhere for the tangential components
  // (K_AC extrapolates u, v through a z wall exactly as the wall rule Q(0) = 2Q(1) - Q(2) does), not for w in the two planes next to a z wall, where the
  // reference extrapolates the cell-centred w_c (sgs.f90:751-766) and the face values are what they are: those planes keep the filter of w_c (slot 0).
  constexpr bool FUC = !YW;
  constexpr int NSH = FUC ? 6 : 9, Q0 = FUC ? 3 : 0;      // filtered through LDS: the six products, or all nine
  __shared__ real sh[2][NSH][TL + 2][64];
  __shared__ real shw[FUC ? TL + 2 : 1][64];      // FUC: w_c in the plane next to a z wall (one buffer: the two wall planes are never consecutive, n3 >= 3)
  __shared__ real ring[4][3][TL + 2][64];
  // plane sums: every working wave adds its lanes in fours (two DPP steps) and leaves sixteen partial sums per quantity; the first halo wave, idle
  // behind the barrier, adds the 128 partials of the previous plane (the full six-step wave sum ran in all ten waves before: 24 vector instructions
  // per plane and wave less in a pass that is bound by them)
  __shared__ real psum[2][2][TL][16];
  __shared__ real bsum[2][LMF_KMAX];
  const int tx = threadIdx.x, ty = threadIdx.y;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (B.bm.gx && !band_block(B.bm, bx, by, bz)) return;
  by += B.by0;
  const int i = bx * 62 + tx, j = by * TL + ty;
  const int kbeg = bz * A.kchunk + 1, kend = min(kbeg + A.kchunk - 1, g.n3);
  const bool outok = tx >= 1 && tx <= 62 && ty >= 1 && ty <= TL && i <= g.n1 && j <= g.n2;
  const int iw = A.perx ? (i == 0 ? g.n1 : (i == g.n1 + 1 ? 1 : i)) : i;
  const int ic = min(iw, g.n1 + 1), jc = min(j, g.n2 + 1);      // clamped: lanes / rows beyond the field read its last ghost column / row
  const OFF c0 = (OFF)g.ix(ic, jc, 0) * RSZ, sk = (OFF)g.s12 * RSZ, sj = (OFF)g.s1 * RSZ;      // byte offsets
  // rows whose |S|Sij this thread combines: its own and the two beside it; the two halo waves (and rows beyond n2) take a row of the tile
  // interior instead (cache hits, results unused)
  const int jo = max(by * TL + 1, min(min(j, by * TL + TL), g.n2));
  const OFF c0s = (OFF)g.ix(ic, jo, 0) * RSZ;
  real sm[3], sc[3], sp[3], sn[3], fn[3];
  // ghost rows of u_f and w_f at wall-model y faces: 2 Q(1) - Q(2) along y (extrapolate(...,lwm) after bounduvw, sgs.f90:683-748); v_f keeps its own.
  // Branch-free: every row loads two values, c1 Q(o1) - c2 Q(o2), with (c1, c2) = (1, 0) and o1 = o2 = its own cell away from such a face
  const int yex = !YW ? 0 : (A.wmylo && j == 0) ? 1 : (A.wmyhi && j == g.n2 + 1) ? -1 : 0;
  const OFF e1 = yex > 0 ? sj : 0, e2 = yex > 0 ? 2 * sj : 0, b1 = yex < 0 ? sj : 0, b2 = yex < 0 ? 2 * sj : 0;      // o + e - b (unsigned offsets: no negative steps)
  const real ec1 = yex != 0 ? 2. : 1., ec2 = yex != 0 ? 1. : 0.;
  auto ldf = [&](int q, OFF o) -> real {
    if (YW && q != 1) return ec1 * ldb(A.uf[q], o + e1 - b1) - ec2 * ldb(A.uf[q], o + e2 - b2);
    return ldb(A.uf[q], o);
  };
  const bool ylo = YW && A.wylo && j == 1, yhi = YW && A.wyhi && j == g.n2;      // rows next to a y wall: ghost row of every filtered quantity = 2 Q(1) - Q(2)
  // UCF: v(j) and v(j-1) of this thread's row, or -- ghost row 0, whose lower neighbour is not in the slab -- the stored v_c of that row with weight 1
  const real *pva = (UCF && jc == 0) ? A.vcg : A.uc[1];
  const OFF ovb = (UCF && jc > 0) ? sj : 0;
  const real cva = (UCF && jc == 0) ? 1. : .5, cvb = (UCF && jc == 0) ? 0. : .5;
  real wprev = 0.;
  const OFF c0m = (OFF)g.ix((A.xwrap && ic == 1) ? g.n1 : max(ic - 1, 0), jc, 0) * RSZ;      // u(i-1): the wrapped column left of the first cell of a row
  auto ucload = [&](int kk, real *o) {      // cell-centred velocity of plane kk (wprev = w of plane kk - 1 on entry, of plane kk on exit)
    const OFF a = c0 + (OFF)kk * sk;
    o[0] = .5 * (ldb(A.uc[0], a) + ldb(A.uc[0], c0m + (OFF)kk * sk));
    o[1] = cva * ldb(pva, a) + cvb * ldb(A.uc[1], a - ovb);
    const real wn = ldb(A.uc[2], a); o[2] = .5 * (wn + wprev); wprev = wn;
  };
  if (UCF) {
    const int k2 = kbeg - 2 >= 0 ? kbeg - 2 : (A.perz ? g.n3 - 1 : 0);      // w below the first plane (periodic z: wrapped; walls: never used)
    wprev = ldb(A.uc[2], c0 + (OFF)k2 * sk);
    ucload(kbeg - 1, sm); ucload(kbeg, sc); ucload(kbeg + 1, sp);
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    if (!UCF) {
      sm[q] = ldb(A.uc[q], c0 + (OFF)(kbeg - 1) * sk);
      sc[q] = ldb(A.uc[q], c0 + (OFF)kbeg * sk);
      sp[q] = ldb(A.uc[q], c0 + (OFF)(kbeg + 1) * sk);
    }
    ring[(kbeg - 1) & 3][q][ty][tx] = ldf(q, c0 + (OFF)(kbeg - 1) * sk);
    ring[kbeg & 3][q][ty][tx] = ldf(q, c0 + (OFF)kbeg * sk);
    fn[q] = ldf(q, c0 + (OFF)(kbeg + 1) * sk);
    if (A.wmlo && kbeg == 1 && q < 2) ring[0][q][ty][tx] = (1. + A.flo) * ring[1][q][ty][tx] - A.flo * fn[q];
  }
  // |S|Sij: y and x combination of one plane (three rows in, lanes beside by DPP)
  auto ssload = [&](int kk, real (*raw)[3]) {
    if (PAIR) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const OFF o = 2 * (c0s + (OFF)kk * sk), s2 = 2 * sj;      // a pair field is twice as wide: byte offsets double
        const real2 a = *(const real2 *)((const char *)B.ss2[q] + (o - s2)), b = *(const real2 *)((const char *)B.ss2[q] + o), c_ = *(const real2 *)((const char *)B.ss2[q] + (o + s2));
        raw[2 * q][0] = a.x; raw[2 * q + 1][0] = a.y; raw[2 * q][1] = b.x; raw[2 * q + 1][1] = b.y; raw[2 * q][2] = c_.x; raw[2 * q + 1][2] = c_.y;
      }
    } else
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const OFF o = c0s + (OFF)kk * sk;
      raw[q][0] = ldb(B.ss[q], o - sj); raw[q][1] = ldb(B.ss[q], o); raw[q][2] = ldb(B.ss[q], o + sj);
    }
  };
  auto sscomb = [&](const real (*raw)[3], real *X) {
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const real Y = (ylo ? 2. * raw[q][1] - raw[q][2] : raw[q][0]) + 2. * raw[q][1] + (yhi ? 2. * raw[q][1] - raw[q][0] : raw[q][2]);
      X[q] = lane_prev(Y) + 2. * Y + lane_next(Y);
    }
  };
  real xm[6], xc[6], xp[6], rw[6][3];
  // the ghost plane below a wall is never used (LO planes take 4 xc); it is read all the same when the chunk starts inside the field
  ssload(kbeg - 1, rw); sscomb(rw, xm);
  ssload(kbeg, rw); sscomb(rw, xc);
  ssload(kbeg + 1, rw);
  const int blk = by * B.gx + bx;
  auto fold = [&](int k, int b) {      // block sums of plane k by ONE whole wave, fixed order
    const real *p0 = &psum[b][0][0][0], *p1 = &psum[b][1][0][0];
    static_assert(TL * 16 >= 128 && TL * 16 <= 192, "two or three partials per lane");
    constexpr int NP3 = TL * 16 - 128;      // partials beyond the first 128
    const int t3 = min(tx, NP3 > 0 ? NP3 - 1 : 0) + 128; const real w3 = (NP3 > 0 && tx < NP3) ? 1. : 0.;
    const real a = wave_sum_lane63(NP3 > 0 ? p0[tx] + p0[tx + 64] + w3 * p0[t3] : p0[tx] + p0[tx + 64]),
               bs = wave_sum_lane63(NP3 > 0 ? p1[tx] + p1[tx + 64] + w3 * p1[t3] : p1[tx] + p1[tx + 64]);
    if (tx == 63) { bsum[0][k - kbeg] = a; bsum[1][k - kbeg] = bs; }
  };
  // grid coefficients of the plane, one plane ahead through the scalar cache (uniform: scalar registers)
  real zcm = ldc(A.dzci, kbeg - 1), zcc = ldc(A.dzci, kbeg), zfc = ldc(A.dzfi, kbeg);
  auto plane = [&](const int k, auto lo_c, auto hi_c) {
    constexpr bool LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
    const int km = (k - 1) & 3, kc = k & 3, kp = (k + 1) & 3, buf = k & 1;
    const OFF idx = c0 + (OFF)min(k + 2, g.n3 + 1) * sk;      // plane k+2 (clamped behind the last ghost plane: loaded again, never used)
    const real zcn = ldc(A.dzci, k + 1), zfn = ldc(A.dzfi, k + 1);
#pragma unroll
    for (int q = 0; q < 3; ++q)
      ring[kp][q][ty][tx] = (HI && A.wmhi && q < 2) ? (1. + A.fhi) * ring[kc][q][ty][tx] - A.fhi * ring[km][q][ty][tx] : fn[q];
    if (UCF) ucload(min(k + 2, g.n3 + 1), sn);
#pragma unroll
    for (int q = 0; q < 3; ++q) { if (!UCF) sn[q] = ldb(A.uc[q], idx); fn[q] = ldf(q, idx); }
    real qm[9], qc[9], qp[9], r[9];
    uiuj(sc, qc);
    if (!LO && !HI) { uiuj(sm, qm); uiuj(sp, qp); }
#pragma unroll
    for (int q = Q0; q < 9; ++q) {
      const real G = (LO || HI) ? 4. * qc[q] : qm[q] + 2. * qc[q] + qp[q];
      r[q] = lane_prev(G) + 2. * G + lane_next(G);
      sh[buf][q - Q0][ty][tx] = r[q];
    }
    if (FUC && (LO || HI)) { const real G = 4. * qc[2]; r[2] = lane_prev(G) + 2. * G + lane_next(G); shw[ty][tx] = r[2]; }      // w_c next to a z wall
    __syncthreads();
    if (k > kbeg && ty == 0) fold(k - 1, buf ^ 1);
    // plane k+1 of |S|Sij: its 18 loads were issued at the end of the previous plane and are folded into six values here, before
    // the register-hungry part; the next 18 are issued after it (keeps the kernel under the 168 VGPRs that ten waves per block need)
    sscomb(rw, xp);
    real lm = 0., mm = 0.;
    if (outok) {
      real F[9];
#pragma unroll
      for (int q = Q0; q < 9; ++q) {
        const real dn = sh[buf][q - Q0][ty - 1][tx], up = sh[buf][q - Q0][ty + 1][tx];
        F[q] = ((ylo ? 2. * r[q] - up : dn) + 2. * r[q] + (yhi ? 2. * r[q] - dn : up)) / 64.;
      }
      if (FUC) {
        F[0] = .5 * (ring[kc][0][ty][tx] + ring[kc][0][ty][tx - 1]); F[1] = .5 * (ring[kc][1][ty][tx] + ring[kc][1][ty - 1][tx]);
        if (LO || HI) F[2] = (shw[ty - 1][tx] + 2. * r[2] + shw[ty + 1][tx]) / 64.;
        else F[2] = .5 * (ring[kc][2][ty][tx] + ring[km][2][ty][tx]);
      }
      const real l0 = F[3] - F[0] * F[0], l1 = F[4] - F[1] * F[1], l2 = F[5] - F[2] * F[2], l3 = F[6] - F[0] * F[1],
                   l4 = F[7] - F[0] * F[2], l5 = F[8] - F[1] * F[2];
      // strain rate of the test-filtered velocity (sgs.f90:571-630). The eight differences of each off-diagonal component telescope
      // pairwise wherever the metric factor is the same (x and y: always; z: not, the grid is stretched), e.g.
      // (u_cpc-u_ccc)+(u_ccc-u_cmc)+(u_mpc-u_mcc)+(u_mcc-u_mmc) = u_cpc-u_cmc+u_mpc-u_mmc: fewer operations and fewer values in flight
      // (this pass is bound by its vector instructions and registers). Same sum in another association: round-off level differences.
#define RU(dk, dj, di) ring[dk][0][ty + (dj)][tx + (di)]
#define RV(dk, dj, di) ring[dk][1][ty + (dj)][tx + (di)]
#define RW(dk, dj, di) ring[dk][2][ty + (dj)][tx + (di)]
      const real dxi = A.dxi, dyi = A.dyi, zc = zcc, zm = zcm;
      real sij[6];
      { const real u_ccc = RU(kc, 0, 0), u_mcc = RU(kc, 0, -1), v_ccc = RV(kc, 0, 0), v_cmc = RV(kc, -1, 0), w_ccc = RW(kc, 0, 0), w_ccm = RW(km, 0, 0);
        sij[0] = (u_ccc - u_mcc) * dxi; sij[1] = (v_ccc - v_cmc) * dyi; sij[2] = (w_ccc - w_ccm) * zfc;
        // s12 = 1/8 [ dyi (u_cpc - u_cmc + u_mpc - u_mmc) + dxi (v_pcc + v_pmc - v_mcc - v_mmc) ]
        const real du = (RU(kc, 1, 0) - RU(kc, -1, 0)) + (RU(kc, 1, -1) - RU(kc, -1, -1));
        const real dv = (RV(kc, 0, 1) - RV(kc, 0, -1)) + (RV(kc, -1, 1) - RV(kc, -1, -1));
        sij[3] = .125 * (du * dyi + dv * dxi);
        // s13 = 1/8 [ zc (u_ccp - u_ccc + u_mcp - u_mcc) + zm (u_ccc - u_ccm + u_mcc - u_mcm) + dxi (w_pcc + w_pcm - w_mcc - w_mcm) ]
        const real up = (RU(kp, 0, 0) - u_ccc) + (RU(kp, 0, -1) - u_mcc), um = (u_ccc - RU(km, 0, 0)) + (u_mcc - RU(km, 0, -1));
        const real dw = (RW(kc, 0, 1) - RW(kc, 0, -1)) + (RW(km, 0, 1) - RW(km, 0, -1));
        sij[4] = .125 * (up * zc + um * zm + dw * dxi);
        // s23 = 1/8 [ zc (v_ccp - v_ccc + v_cmp - v_cmc) + zm (v_ccc - v_ccm + v_cmc - v_cmm) + dyi (w_cpc + w_cpm - w_cmc - w_cmm) ]
        const real vp = (RV(kp, 0, 0) - v_ccc) + (RV(kp, -1, 0) - v_cmc), vm = (v_ccc - RV(km, 0, 0)) + (v_cmc - RV(km, -1, 0));
        const real dw2 = (RW(kc, 1, 0) - RW(kc, -1, 0)) + (RW(km, 1, 0) - RW(km, -1, 0));
        sij[5] = .125 * (vp * zc + vm * zm + dw2 * dyi); }
#undef RU
#undef RV
#undef RW
      const real s0 = sqrt(2. * (sij[0] * sij[0] + sij[1] * sij[1] + sij[2] * sij[2] + 2. * (sij[3] * sij[3] + sij[4] * sij[4] + sij[5] * sij[5])));
      const real a2s0 = (LO || HI || ylo || yhi ? 2.52 : 4.00) * s0;      // alph2: cells next to any wall (sgs.f90:783-816)
      real m[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const real G = (LO || HI) ? 4. * xc[q] : xm[q] + 2. * xc[q] + xp[q];      // filter(|S|Sij) * 64
        m[q] = 2. * (G * (1. / 64.) - a2s0 * sij[q]);                             // Mij, sgs.f90:261-272
      }
      lm = m[0] * l0 + m[1] * l1 + m[2] * l2 + (m[3] * l3 + m[4] * l4 + m[5] * l5) * 2.;       // sgs.f90:344-349
      mm = m[0] * m[0] + m[1] * m[1] + m[2] * m[2] + (m[3] * m[3] + m[4] * m[4] + m[5] * m[5]) * 2.;       // sgs.f90:350-355
    }
    ssload(min(k + 2, g.n3 + 1), rw);
    lm += dpp_f64<0x111>(lm); lm += dpp_f64<0x112>(lm); mm += dpp_f64<0x111>(mm); mm += dpp_f64<0x112>(mm);      // row_shr 1, 2: lanes 3, 7, 11, ... hold four lanes' sum
    if ((tx & 3) == 3 && ty >= 1 && ty <= TL) { psum[buf][0][ty - 1][tx >> 2] = lm; psum[buf][1][ty - 1][tx >> 2] = mm; }
#pragma unroll
    for (int q = 0; q < 3; ++q) { sm[q] = sc[q]; sc[q] = sp[q]; sp[q] = sn[q]; }
#pragma unroll
    for (int q = 0; q < 6; ++q) { xm[q] = xc[q]; xc[q] = xp[q]; }
    zcm = zcc; zcc = zcn; zfc = zfn;
  };
  {
    const std::true_type T; const std::false_type F_;
    int k = kbeg;
    const int klast = (A.zhi && kend == g.n3) ? kend - 1 : kend;
    if (A.zlo && k == 1 && k <= kend) { plane(k, T, F_); ++k; }
    for (; k <= klast; ++k) plane(k, F_, F_);
    if (k <= kend) plane(k, F_, T);
  }
  __syncthreads();
  if (ty == 0 && kend >= kbeg) fold(kend, kend & 1);
  __syncthreads();
  // the chunk's block sums leave in one go
  for (int q = ty * 64 + tx; q < 2 * (kend - kbeg + 1); q += 64 * (TL + 2)) {
    const int w = q & 1, kk = q >> 1;
    A.part[(size_t)(w * g.n3 + kbeg + kk - 1) * A.nblk + blk] = bsum[w][kk];
  }
}

// K_A + K_C in one pass over u,v,w: strain rate (sgs.f90:571-630) stored as |S| and |S|Sij, cell-centred velocity (sgs.f90:860-869) and the
// test-filtered velocity (sgs.f90:632-679 with the wall rule), all from an LDS ring of three raw planes.
#ifndef TYS
#define TYS 14      // measured at 512^3 (A/B on one box): 14 beats 6 and 10 for the dynamic-model pass (-7 %) and the Smagorinsky pass (-26 %)
#endif
#ifndef TYC
#define TYC 10      // tile height of k_corr_strain_tile (K_AC with the projection folded in). Measured at 512^3 on one box: 10 rows (twelve waves, 144 VGPRs, 88 KB)
#endif              // 3.37 ms per call, 14 rows (sixteen waves at the 128-register cap, 117 KB) 3.49; a 14-row build that spilled fourteen registers 4.5
struct StrainTileArgs {
  const real *u[3];
  real *s0, *ssij[6], *uc[3], *uf[3];
  real2 *ss2[3];      // PAIR = 1: |S|Sij as three fields of pairs (S11,S22), (S33,S12), (S13,S23) instead of ssij
  const real *dzci, *dzfi;
  real dxi, dyi;
  int kchunk, zlo, zhi;
  int wmlo, wmhi; real flo, fhi;      // wall-model z faces: ghost planes of u,v by extrapolate(...,lwm), sgs.f90:683-748
  // SMAG = 1 (static Smagorinsky with van Driest damping for z walls, sgs.f90:98-152): the only output is visct
  real *visct; const real *zc, *del; real l3, visc;      // del(k) = (dx dy dzf(k))^(1/3)
  // SMAG = 1 with walls in y (ducts): wylo/wyhi = no-slip y walls owned by this rank (van Driest distance and shear, the latter from
  // twy(side, k, i) = sqrt(tau_w) of k_wall_shear_y); wmylo/wmyhi = wall-model y faces: the strain rate sees ghost rows of u and w
  // extrapolated from the interior (extrapolate(...,lwm) along y, sgs.f90:683-748)
  int wylo, wyhi, wmylo, wmyhi; const real *twy; real dl2;
  BandMap bm;      // block -> (x tile, y tile, k chunk) map of the 1-D launches (bm.gx = 0: plain 3-D grid)
  int perx;        // x periodic: halo columns beyond the ends of a row are the wrapped interior columns (the ghost columns may be stale inside cales_step)
  // CORR = 1 (cales_step, x and y periodic, one rank, z walls or z periodic, explicit diffusion, no wall model): u[] is the PREDICTION u*, v*, w* and the
  // projection (correc.f90:44-67 with the deferred bulk forcing) + pressure update (updatep.f90:30-47) happen here -- every velocity value the pass
  // reads is corrected while it is loaded, u = (u* + f) - dtrk grad(pp) at the periodically wrapped interior cell, the z ghost planes by their
  // boundary rule (bounduvw with is_correc, bound.f90:18-154), the corrected velocity of the tile's own cells goes to un[] (a second set of buffers:
  // neighbouring tiles still read u*) and p += pp in place. The pass k_correc_cell<1> (9 words per cell) disappears: 13 + 9 -> 19 words.
  const real *pp; real *p; real *un[3]; const real *force; int fmask; real cfi, cfj, cdt;
  real *vcg = nullptr;      // EXT = 1: the field whose ghost rows 0 and n2+1 receive v_c (LijMijArgs::vcg)
  const real *bcz[2][2];      // Dirichlet planes of u and v at the two z walls (bounduvw's 2 bc - u(1)), [component][side]
  int zper;                   // z periodic: ghost planes are the wrapped interior planes
  // several slabs (pery = 0): the ghost rows 0 and n2+1 of u*, v*, w* and pp hold the neighbours' rows (exchanged), and the one value further out that the
  // correction of v in row n2+1 needs -- pp of the upper neighbour's row 2 -- sits in ghost row n2+1 of the COMPANION field, allocated right behind pp
  // (ppd = its distance in bytes; cales_step copies row 2 there and exchanges both fields in one message). One slab: rows wrap around.
  int pery; size_t ppd;
};
// sqrt(tau_w) at the two y walls for every (i, k): the argument of the van Driest damping of the cells whose nearest wall is a y wall
// (sgs.f90:117-143, cases 3 and 4 of the select), from the fields themselves (their ghost cells, not the extrapolated ones)
__global__ __launch_bounds__(256) void k_wall_shear_y(Geom g, const real *__restrict__ u, const real *__restrict__ w, real visc, real dyi,
                                                      int lo, int hi, real *__restrict__ twy, int perx) {
  const int i = blockIdx.x * 64 + threadIdx.x + 1, k = blockIdx.y * 4 + threadIdx.y + 1;
  if (i > g.n1 || k > g.n3) return;
  const int n2 = g.n2;
  const int im1 = (perx && i == 1) ? g.n1 : i - 1;      // x periodic inside cales_step: the wrapped interior column, the ghost column 0 may be stale (step_xskip)
  if (lo) {
    const real t1 = u[g.ix(i, 1, k)] - u[g.ix(i, 0, k)] + u[g.ix(im1, 1, k)] - u[g.ix(im1, 0, k)];
    const real t2 = w[g.ix(i, 1, k)] - w[g.ix(i, 0, k)] + w[g.ix(i, 1, k - 1)] - w[g.ix(i, 0, k - 1)];
    twy[(size_t)k * g.s1 + i] = sqrt(0.5 * visc * (sqrt(t1 * t1 + t2 * t2) * dyi));
  }
  if (hi) {
    const real t1 = u[g.ix(i, n2, k)] - u[g.ix(i, n2 + 1, k)] + u[g.ix(im1, n2, k)] - u[g.ix(im1, n2 + 1, k)];
    const real t2 = w[g.ix(i, n2, k)] - w[g.ix(i, n2 + 1, k)] + w[g.ix(i, n2, k - 1)] - w[g.ix(i, n2 + 1, k - 1)];
    twy[(size_t)(g.n3 + 2 + k) * g.s1 + i] = sqrt(0.5 * visc * (sqrt(t1 * t1 + t2 * t2) * dyi));
  }
}
template <typename OFF, int TY, int YW, int PAIR = 0>      // YW = 1: walls or wall-model faces in y (ducts); the channel instantiations carry none of that logic
__global__ __launch_bounds__(64 * (TY + 2)) void k_strain_tile(Geom g, StrainTileArgs A) {
  // (one barrier per plane with four ring slots and double-buffered sums, as in k_lij_mij_tile, measured 13 % slower here)
  __shared__ real ring[3][3][TY + 2][66];      // rows: x-halo cell, 64 own cells, x-halo cell
  __shared__ real shs[3][TY + 2][64];
  const int tx = threadIdx.x, ty = threadIdx.y;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (A.bm.gx && !band_block(A.bm, bx, by, bz)) return;
  const int i = bx * 64 + tx + 1, j = by * TY + ty;        // whole 128-B lines in and out (see cales_create)
  const int kbeg = bz * A.kchunk + 1, kend = min(kbeg + A.kchunk - 1, g.n3);
  const bool edge = tx == 0 || tx == 63;
  const int ih0 = tx == 0 ? i - 1 : i + 1, hx = tx == 0 ? 0 : 65;
  const int ih = !A.perx ? ih0 : ih0 == 0 ? g.n1 : ih0 == g.n1 + 1 ? 1 : ih0;
  const bool ldok = i <= g.n1 + 1 && j <= g.n2 + 1, hok = edge && ih0 <= g.n1 + 1 && j <= g.n2 + 1;
  const bool outok = ty >= 1 && ty <= TY && i <= g.n1 && j <= g.n2;
  const int iq = (A.perx && i == g.n1 + 1) ? 1 : i;      // (the column right of the last cell of the row, loaded by the last tile's lane beside it)
  const OFF c0 = ldok ? (OFF)g.ix(iq, j, 0) * RSZ : 0, ch = hok ? (OFF)g.ix(ih, j, 0) * RSZ : 0, sk = (OFF)g.s12 * RSZ;   // byte offsets
  real fn[3], fh[3];
  // ghost rows at wall-model y faces: u and w are replaced by 2 Q(1) - Q(2) along y; v, normal to the face, is not
  const int yex = !YW ? 0 : (A.wmylo && j == 0) ? 1 : (A.wmyhi && j == g.n2 + 1) ? -1 : 0;
  const OFF sjb = (OFF)g.s1 * RSZ;
  auto ld = [&](int q, OFF o) -> real {
    if (YW && yex != 0 && q != 1) return yex > 0 ? 2. * ldb(A.u[q], o + sjb) - ldb(A.u[q], o + 2 * sjb) : 2. * ldb(A.u[q], o - sjb) - ldb(A.u[q], o - 2 * sjb);
    return ldb(A.u[q], o);
  };
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    ring[(kbeg - 1) % 3][q][ty][tx + 1] = ldok ? ld(q, c0 + (OFF)(kbeg - 1) * sk) : 0.;
    ring[kbeg % 3][q][ty][tx + 1] = ldok ? ld(q, c0 + (OFF)kbeg * sk) : 0.;
    if (edge) {
      ring[(kbeg - 1) % 3][q][ty][hx] = hok ? ld(q, ch + (OFF)(kbeg - 1) * sk) : 0.;
      ring[kbeg % 3][q][ty][hx] = hok ? ld(q, ch + (OFF)kbeg * sk) : 0.;
    }
    fn[q] = ldok ? ld(q, c0 + (OFF)(kbeg + 1) * sk) : 0.;
    fh[q] = hok ? ld(q, ch + (OFF)(kbeg + 1) * sk) : 0.;
    if (A.wmlo && kbeg == 1 && q < 2) {
      ring[0][q][ty][tx + 1] = (1. + A.flo) * ring[1][q][ty][tx + 1] - A.flo * fn[q];
      if (edge) ring[0][q][ty][hx] = (1. + A.flo) * ring[1][q][ty][hx] - A.flo * fh[q];
    }
  }
  int km = (kbeg - 1) % 3, kc = kbeg % 3, kp = (kbeg + 1) % 3;
  for (int k = kbeg; k <= kend; ++k) {
    const OFF idx = c0 + (OFF)k * sk;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const bool ex = A.wmhi && k == g.n3 && q < 2;
      ring[kp][q][ty][tx + 1] = ex ? (1. + A.fhi) * ring[kc][q][ty][tx + 1] - A.fhi * ring[km][q][ty][tx + 1] : fn[q];
      if (edge) ring[kp][q][ty][hx] = ex ? (1. + A.fhi) * ring[kc][q][ty][hx] - A.fhi * ring[km][q][ty][hx] : fh[q];
    }
    if (k + 2 <= g.n3 + 1) {
#pragma unroll
      for (int q = 0; q < 3; ++q) { fn[q] = ldok ? ld(q, idx + 2 * sk) : 0.; fh[q] = hok ? ld(q, ch + (OFF)(k + 2) * sk) : 0.; }
    }
    __syncthreads();
    const bool lo = A.zlo && k == 1, hi = A.zhi && k == g.n3;
    real r[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      auto zcomb = [&](int x) {
        const real qm = ring[km][q][ty][x], qc = ring[kc][q][ty][x], qp = ring[kp][q][ty][x];
        const real vm = (lo && q < 2) ? 2. * qc - qp : qm;      // u,v extrapolated through the walls, w (on the faces) not
        const real vp = (hi && q < 2) ? 2. * qc - qm : qp;
        return vm + 2. * qc + vp;
      };
      const real G = zcomb(tx + 1);
      real pv = lane_prev(G), nx = lane_next(G);
      if (edge) { const real Gh = zcomb(hx); if (tx == 0) pv = Gh; else nx = Gh; }
      r[q] = pv + 2. * G + nx;
      shs[q][ty][tx] = r[q];
    }
    if (outok) {
#define RU(dk, dj, di) ring[dk][0][ty + (dj)][tx + 1 + (di)]
#define RV(dk, dj, di) ring[dk][1][ty + (dj)][tx + 1 + (di)]
#define RW(dk, dj, di) ring[dk][2][ty + (dj)][tx + 1 + (di)]
      const real u_mcm = RU(km, 0, -1), u_ccm = RU(km, 0, 0), u_mmc = RU(kc, -1, -1), u_cmc = RU(kc, -1, 0), u_mcc = RU(kc, 0, -1),
                   u_ccc = RU(kc, 0, 0), u_mpc = RU(kc, 1, -1), u_cpc = RU(kc, 1, 0), u_mcp = RU(kp, 0, -1), u_ccp = RU(kp, 0, 0);
      const real v_cmm = RV(km, -1, 0), v_ccm = RV(km, 0, 0), v_mmc = RV(kc, -1, -1), v_cmc = RV(kc, -1, 0), v_pmc = RV(kc, -1, 1),
                   v_mcc = RV(kc, 0, -1), v_ccc = RV(kc, 0, 0), v_pcc = RV(kc, 0, 1), v_cmp = RV(kp, -1, 0), v_ccp = RV(kp, 0, 0);
      const real w_cmm = RW(km, -1, 0), w_mcm = RW(km, 0, -1), w_ccm = RW(km, 0, 0), w_pcm = RW(km, 0, 1), w_cpm = RW(km, 1, 0),
                   w_cmc = RW(kc, -1, 0), w_mcc = RW(kc, 0, -1), w_ccc = RW(kc, 0, 0), w_pcc = RW(kc, 0, 1), w_cpc = RW(kc, 1, 0);
#undef RU
#undef RV
#undef RW
      const real dxi = A.dxi, dyi = A.dyi, zc = ldc(A.dzci, k), zm = ldc(A.dzci, k - 1);
      const real s11 = (u_ccc - u_mcc) * dxi, s22 = (v_ccc - v_cmc) * dyi, s33 = (w_ccc - w_ccm) * ldc(A.dzfi, k);
      const real s12 = .125 * ((u_cpc - u_ccc) * dyi + (v_pcc - v_ccc) * dxi + (u_ccc - u_cmc) * dyi + (v_pmc - v_cmc) * dxi +
                                 (u_mpc - u_mcc) * dyi + (v_ccc - v_mcc) * dxi + (u_mcc - u_mmc) * dyi + (v_cmc - v_mmc) * dxi);
      const real s13 = .125 * ((u_ccp - u_ccc) * zc + (w_pcc - w_ccc) * dxi + (u_ccc - u_ccm) * zm + (w_pcm - w_ccm) * dxi +
                                 (u_mcp - u_mcc) * zc + (w_ccc - w_mcc) * dxi + (u_mcc - u_mcm) * zm + (w_ccm - w_mcm) * dxi);
      const real s23 = .125 * ((v_ccp - v_ccc) * zc + (w_cpc - w_ccc) * dyi + (v_ccc - v_ccm) * zm + (w_cpm - w_ccm) * dyi +
                                 (v_cmp - v_cmc) * zc + (w_ccc - w_cmc) * dyi + (v_cmc - v_cmm) * zm + (w_ccm - w_cmm) * dyi);
      const real s0v = sqrt(2. * (s11 * s11 + s22 * s22 + s33 * s33 + 2. * (s12 * s12 + s13 * s13 + s23 * s23)));
      {
        stb(A.s0, idx, s0v);                                              // stands for visct = s0 (sgs.f90:184) until the final kernel
        if (PAIR) {      // |S|Sij (sgs.f90:198-210), two components per 16-byte store
          const OFF i2 = 2 * idx;
          typedef real v2 __attribute__((ext_vector_type(2)));      // (streaming stores: -2 % on this pass, measured three times)
          __builtin_nontemporal_store(v2{s0v * s11, s0v * s22}, (v2 *)((char *)A.ss2[0] + i2));
          __builtin_nontemporal_store(v2{s0v * s33, s0v * s12}, (v2 *)((char *)A.ss2[1] + i2));
          __builtin_nontemporal_store(v2{s0v * s13, s0v * s23}, (v2 *)((char *)A.ss2[2] + i2));
        } else {
        stb(A.ssij[0], idx, s0v * s11); stb(A.ssij[1], idx, s0v * s22); stb(A.ssij[2], idx, s0v * s33);      // |S|Sij (sgs.f90:198-210)
        stb(A.ssij[3], idx, s0v * s12); stb(A.ssij[4], idx, s0v * s13); stb(A.ssij[5], idx, s0v * s23);
        }
        if (A.uc[0]) { stb(A.uc[0], idx, 0.5 * (u_ccc + u_mcc)); stb(A.uc[1], idx, 0.5 * (v_ccc + v_cmc)); stb(A.uc[2], idx, 0.5 * (w_ccc + w_ccm)); }      // (null: the last pass forms them itself)
      }
    }
    __syncthreads();
    if (outok) {
      // next to a no-slip y wall the ghost row of u and w is the extrapolation 2 Q(1) - Q(2) (extrapolate(...,cbc), sgs.f90:705-710): its
      // y combination is 4 Q(1); v, normal to the wall, keeps its ghost row
      const bool ylo = YW && A.wylo && j == 1, yhi = YW && A.wyhi && j == g.n2;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const real dn = shs[q][ty - 1][tx], up = shs[q][ty + 1][tx];
        const real a = (ylo && q != 1) ? 2. * r[q] - up : dn, b = (yhi && q != 1) ? 2. * r[q] - dn : up;
        stb(A.uf[q], idx, (a + 2. * r[q] + b) / 64.);
      }
    }
    const int t = km; km = kc; kc = kp; kp = t;
  }
}
// ---- K_AC with the projection folded in (StrainTileArgs, "CORR"; cales_step on one rank, x and y periodic, z walls or periodic): every velocity value
// the pass reads is corrected while it is loaded, u = (u* + f) - dtrk grad(pp) at the periodically wrapped interior cell (the operations of
// k_correc_cell in their order), the z ghost planes by their boundary rule; the corrected velocity of the tile's own cells goes to un[] (a second set of
// buffers: neighbouring tiles still read u*) and p += pp in place. Same tile, ring and strain-rate / filter arithmetic as k_strain_tile<.., PAIR = 1>.
// Memory accesses are laid out for the hardware's in-order counters (see k_momrk): every global access of the plane loop is unconditional straight-line
// code -- loads from clamped / wrapped cells, stores of lanes without an output to the x ghost cell of their row (dead inside cales_step: every reader
// wraps around, the step's last ghost-cell update rewrites it) -- so the waits are counted (the loads of plane k+2 stay in flight behind the eleven
// stores of plane k) instead of s_waitcnt vmcnt(0) at every use. A plane is completed ONE iteration after its loads were issued. The two y-halo
// waves have no outputs: they run a loop of their own and complete the tile's two x-halo COLUMNS, one tile row per lane (wave 0 the column left of the
// tile, wave TY+1 the one right of it), which as a branch of the edge lanes ran in every one of the sixteen waves.
// EXT = 1 (several slabs, StepPlan::fold_rows2): the tile rows start one row lower and the pass ALSO forms the two ghost rows j = 0 and n2+1 of everything
// it writes -- corrected u, v, w, p + pp, |S|, |S|Sij, the filtered velocity, and v_c of those two rows for the last pass (A.vcg) -- from a SECOND ghost
// row of the prediction (rows -1 / n2+2: the ghost rows 0 / n2+1 of the companion field that sits A.ppd bytes behind every velocity field) and a third of
// pp (row n2+3: its second companion): nothing this pass produces travels between the slabs afterwards.
template <typename OFF, int TY, int EXT = 0>
__global__ __launch_bounds__(64 * (TY + 2)) void k_corr_strain_tile(Geom g, StrainTileArgs A) {
  __shared__ real ring[3][3][TY + 2][66];      // rows: x-halo cell, 64 own cells, x-halo cell
  __shared__ real shs[3][TY + 2][64];
  // pp of the tile's cells, planes k+1 / k+2 (by parity), column 64 = the x-halo cell right of the row: the correction of a cell needs pp(i+1) (the lane
  // beside: DPP; last lane: column 64), pp(j+1) (the row above: here) and pp(k+1) (its own next value) -- two global loads per cell and plane less
  __shared__ real sP[2][TY + 2][65];
  const int tx = threadIdx.x, ty = threadIdx.y;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (A.bm.gx && !band_block(A.bm, bx, by, bz)) return;
  const int n1 = g.n1, n2 = g.n2, n3 = g.n3;      // (n1 a multiple of 64: every tile is full in x, dsmag_fast)
  const int i = bx * 64 + tx + 1, j = by * TY + ty - EXT;
  const int kbeg = bz * A.kchunk + 1, kend = min(kbeg + A.kchunk - 1, n3);
  const bool edge = tx == 0 || tx == 63;
  const int hx = tx == 0 ? 0 : 65;
  const bool outok = ty >= 1 && ty <= TY && j >= 1 - EXT && j <= n2 + EXT;
  // the rows this thread loads. One slab: wrapped in y (rows beyond n2+1 keep wrapping: row n2+2 holds pp of row 2, which the correction of row n2+1 --
  // the wrapped row 1, the halo of the last row -- reads from sP). Several slabs: the ghost rows as they are, pp of "row n2+2" from the companion field
  auto urow = [&](int r) { return A.pery ? (r == 0 ? n2 : r > n2 ? min(r - n2, n2) : r) : min(max(r, 0), n2 + 1); };
  // byte offset (from its field's base) of cell (col, row r, plane 0) of a field with `nc` companions behind it: several slabs -- the rows 0 .. n2+1 as they
  // are, row -1 in the first companion's ghost row 0, row n2+1+q (q = 1 .. nc) in companion q's ghost row n2+1 (rows further out: the last of them; the
  // values only reach tile rows that have no output). nc = 0: clamped to the ghost rows.
  auto foff = [&](int col, int r, int nc) -> OFF {
    if (A.pery) return (OFF)g.ix(col, urow(r), 0) * RSZ;
    if (EXT && r < 0) return (OFF)(A.ppd + g.ix(col, 0, 0) * RSZ);
    if (r <= n2 + 1 || nc == 0) return (OFF)g.ix(col, min(r, n2 + 1), 0) * RSZ;
    return (OFF)((size_t)min(r - (n2 + 1), nc) * A.ppd + g.ix(col, n2 + 1, 0) * RSZ);
  };
  constexpr int NCV = EXT, NCP = 1 + EXT;      // companions of the velocity fields / of pp
  const int jq = urow(j);
  const OFF sk = (OFF)g.s12 * RSZ;
  const OFF cl = foff(i, j, NCV), clp = foff(i, j, NCP), cly = foff(i, j + 1, NCP);
  const OFF cdump = (OFF)g.ix(0, min(max(j, 0), n2 + 1), 0) * RSZ;      // the x ghost cell of the row: where lanes / planes without an output of their own store
  const OFF cst = outok ? cl : cdump;
  const OFF cvc = (EXT && outok && (j == 0 || j == n2 + 1)) ? cl : cdump;
  // side job of the y-halo waves: lane tx < TY+2 completes the x-halo cell of tile row tx (other lanes repeat their own cell: no branches around loads)
  const bool hwave = ty == 0 || ty == TY + 1;
  const int sside = ty == 0 ? 0 : 1, hxs = sside ? 65 : 0, si0 = bx * 64 + (sside ? 65 : 0), sj0 = by * TY + tx - EXT;
  const bool sok = hwave && tx < TY + 2 && sj0 <= n2 + 1 + EXT;
  const int si = !sok ? i : si0 == 0 ? n1 : si0 == n1 + 1 ? 1 : si0, sjr = !sok ? jq : urow(sj0), sjx = !sok ? j : sj0;
  const int sxr = si >= n1 ? 1 : si + 1;
  const OFF so = foff(si, sjx, NCV), sop = foff(si, sjx, NCP), soy = foff(si, sjx + 1, NCP), sox = foff(sxr, sjx, NCP);
  const int srow = sok ? tx : 0;
  auto kz = [&](int kk) { return !A.zper ? kk : kk == 0 ? n3 : kk == n3 + 1 ? 1 : kk; };      // plane that holds the values of plane kk
  const real f0 = (A.fmask & 1) ? ldc(A.force, 0) : 0., f1 = (A.fmask & 2) ? ldc(A.force, 1) : 0., f2 = (A.fmask & 4) ? ldc(A.force, 2) : 0.;
  // what is in flight for one column between two iterations: the prediction of a plane and pp of the cell above it (k+1); the side job's column also
  // loads its pp(i+1) and pp(j+1), the own cells find theirs in sP (the top halo row, whose row above is another block's, loads its pp(j+1))
  struct Raw { real q[3], pz; };
  struct RawS { real q[3], pz, px, py; };
  auto rawload = [&](int kk, Raw &r) {      // kk: a plane index in 0..n3+1 (mapped by kz)
    const OFF a = cl + (OFF)kk * sk;
#pragma unroll
    for (int q = 0; q < 3; ++q) r.q[q] = ldb(A.u[q], a);
    r.pz = ldb(A.pp, clp + (OFF)min(kk + 1, n3 + 1) * sk);
  };
  auto rawloads = [&](int kk, RawS &r) {
    const OFF a = so + (OFF)kk * sk;
#pragma unroll
    for (int q = 0; q < 3; ++q) r.q[q] = ldb(A.u[q], a);
    r.pz = ldb(A.pp, sop + (OFF)min(kk + 1, n3 + 1) * sk); r.px = ldb(A.pp, sox + (OFF)kk * sk); r.py = ldb(A.pp, soy + (OFF)kk * sk);
  };
  // interior plane kq: (u* + f) - dtrk grad(pp), k_correc_cell's operations in their order
  auto fix = [&](const real *q, real P0, real px, real py, real pz, int kq, real *o) {
    o[0] = ((A.fmask & 1) ? q[0] + f0 : q[0]) - A.cfi * (px - P0);
    o[1] = ((A.fmask & 2) ? q[1] + f1 : q[1]) - A.cfj * (py - P0);
    o[2] = ((A.fmask & 4) ? q[2] + f2 : q[2]) - A.cdt * ldc(A.dzci, kq) * (pz - P0);
  };
  // ghost plane next to a wall (sd = 0: plane 0, 1: plane n3+1): u, v = 2 bc - (the plane beside it, bes[]) as bounduvw sets them, w(0) by the formula
  // without forcing (correc.f90 covers k = 0..n3 for w), w(n3+1) untouched (bounduvw with is_correc leaves the normal component's z faces alone)
  auto wallfix = [&](const real *q, real P0, real pz, int sd, size_t qbc, const real *bes, real *o) {
    o[0] = 2. * A.bcz[0][sd][qbc] - bes[0]; o[1] = 2. * A.bcz[1][sd][qbc] - bes[1];
    o[2] = sd == 0 ? q[2] - A.cdt * ldc(A.dzci, 0) * (pz - P0) : q[2];
  };
  const size_t q2 = (size_t)i + (size_t)(n1 + 2) * jq, q2s = (size_t)si + (size_t)(n1 + 2) * sjr;
  Raw rn; RawS rh = {}; real p0n, p0h = 0., pyt = 0.;      // in flight at the top of iteration k: plane k+1 of the own / the side job's column, pp of its cell, p of the own cell; pyt: pp(j+1) of the top halo row
  { // ---- planes kbeg-1 and kbeg complete, plane kbeg+1 in flight (pp's neighbours by direct loads here: sP serves the loop)
    real c1[3], c0v[3], h1[3] = {0., 0., 0.}, h0[3] = {0., 0., 0.};
    const bool low = !A.zper && kbeg == 1;      // plane kbeg-1 is the ghost plane below the lower wall
    const OFF clx = foff(i >= n1 ? 1 : i + 1, j, NCP);
    { const int kq = kz(kbeg); Raw r; rawload(kq, r); const real P0 = ldb(A.pp, clp + (OFF)kq * sk);
      fix(r.q, P0, ldb(A.pp, clx + (OFF)kq * sk), ldb(A.pp, cly + (OFF)kq * sk), r.pz, kq, c1); p0n = r.pz;
      const OFF a = cst + (OFF)kbeg * sk; stb(A.p, a, ldb(A.p, a) + P0);      // p += pp (updatep.f90:30-47, explicit diffusion); lanes without output: their ghost cell
      if (hwave) { RawS e; rawloads(kq, e); const real E0 = ldb(A.pp, sop + (OFF)kq * sk); fix(e.q, E0, e.px, e.py, e.pz, kq, h1); p0h = e.pz; } }
    { const int kq = kz(kbeg - 1); Raw r; rawload(kq, r); const real P0 = ldb(A.pp, clp + (OFF)kq * sk);
      if (low) wallfix(r.q, P0, r.pz, 0, q2, c1, c0v); else fix(r.q, P0, ldb(A.pp, clx + (OFF)kq * sk), ldb(A.pp, cly + (OFF)kq * sk), r.pz, kq, c0v);
      if (hwave) { RawS e; rawloads(kq, e); const real E0 = ldb(A.pp, sop + (OFF)kq * sk);
                   if (low) wallfix(e.q, E0, e.pz, 0, q2s, h1, h0); else fix(e.q, E0, e.px, e.py, e.pz, kq, h0); } }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      ring[(kbeg - 1) % 3][q][ty][tx + 1] = c0v[q]; ring[kbeg % 3][q][ty][tx + 1] = c1[q];
      if (sok) { ring[(kbeg - 1) % 3][q][srow][hxs] = h0[q]; ring[kbeg % 3][q][srow][hxs] = h1[q]; }
      stb(A.un[q], cst + (OFF)kbeg * sk, c1[q]);
    }
    sP[(kbeg + 1) & 1][ty][tx] = p0n;      // pp of plane kbeg+1
    if (sok && sside) sP[(kbeg + 1) & 1][srow][64] = p0h;
    rawload(kz(kbeg + 1), rn);
    if (hwave) { rawloads(kz(kbeg + 1), rh); pyt = ldb(A.pp, cly + (OFF)kz(kbeg + 1) * sk); }
    __syncthreads();
  }
  int km = (kbeg - 1) % 3, kc = kbeg % 3, kp = (kbeg + 1) % 3;
  real keep[3] = {0., 0., 0.};
  // one plane. HALO: a y-halo wave (side job, no outputs); TOP: plane k+1 is the ghost plane above the upper wall (k = n3, walls)
  auto plane = [&](const int k, auto halo_c, auto top_c) {
    constexpr bool HALO = decltype(halo_c)::value, TOP = decltype(top_c)::value;
    const OFF idx = cst + (OFF)k * sk;
    { // plane k+1, loaded during the last iteration, is completed, stored if it belongs to this chunk, and plane k+2 goes into flight
      const int par = (k + 1) & 1, k2 = kz(min(k + 2, n3 + 1));
      real cc[3];
      if (TOP) { const real bes[2] = {ring[kc][0][ty][tx + 1], ring[kc][1][ty][tx + 1]}; wallfix(rn.q, p0n, rn.pz, 1, q2, bes, cc); }
      else {
        real px = lane_next(p0n);
        if (tx == 63) px = sP[par][ty][64];
        const real py = (HALO && ty == TY + 1) ? pyt : sP[par][ty + (ty == TY + 1 ? 0 : 1)][tx];
        fix(rn.q, p0n, px, py, rn.pz, kz(k + 1), cc);
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) ring[kp][q][ty][tx + 1] = cc[q];
      sP[par ^ 1][ty][tx] = rn.pz;      // pp of plane k+2
      if (!HALO) {
        // (plane k+1 of the NEXT chunk, or the ghost plane n3+1, goes to the row's x ghost cell like the results of lanes without output)
        const OFF dst = (outok && k + 1 <= kend ? cst : cdump) + (OFF)(k + 1) * sk;
#pragma unroll
        for (int q = 0; q < 3; ++q) stb(A.un[q], dst, cc[q]);
      }
      p0n = rn.pz;
      rawload(k2, rn);
      // (the data registers of the last plane's three filtered-velocity stores stay allocated up to here: reused earlier, the compiler has to wait
      //  for those stores -- the youngest operations in flight, i.e. for everything -- before the first instruction that overwrites them)
      if (!HALO) asm volatile("" :: "v"(keep[0]), "v"(keep[1]), "v"(keep[2]));
      if (HALO) {
        real hh[3];
        if (TOP) { const real hbes[2] = {ring[kc][0][srow][hxs], ring[kc][1][srow][hxs]}; wallfix(rh.q, p0h, rh.pz, 1, q2s, hbes, hh); }
        else fix(rh.q, p0h, sside ? rh.px : sP[par][srow][0], rh.py, rh.pz, kz(k + 1), hh);
        if (sok) {
#pragma unroll
          for (int q = 0; q < 3; ++q) ring[kp][q][srow][hxs] = hh[q];
          if (sside) sP[par ^ 1][srow][64] = rh.pz;
        }
        p0h = rh.pz;
        rawloads(k2, rh);
        pyt = ldb(A.pp, cly + (OFF)k2 * sk);
      }
    }
    __syncthreads();
    // p += pp of plane k+1 (updatep.f90:30-47, explicit diffusion): loaded here, stored behind the second barrier -- load, wait and store inside one
    // iteration, nothing of it carried around the loop; pp(k+1) of the cell is still in sP
    const OFF dstp = (outok && k + 1 <= kend ? cst : cdump) + (OFF)(k + 1) * sk;
    real pl = 0.;
    if (!HALO) pl = ldb(A.p, dstp);
    const bool lo = A.zlo && k == 1, hi = A.zhi && k == n3;
    real r[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      auto zcomb = [&](int x) {
        const real qm = ring[km][q][ty][x], qc = ring[kc][q][ty][x], qp = ring[kp][q][ty][x];
        const real vm = (lo && q < 2) ? 2. * qc - qp : qm;      // u,v extrapolated through the walls, w (on the faces) not
        const real vp = (hi && q < 2) ? 2. * qc - qm : qp;
        return vm + 2. * qc + vp;
      };
      const real G = zcomb(tx + 1);
      real pv = lane_prev(G), nx = lane_next(G);
      if (edge) { const real Gh = zcomb(hx); if (tx == 0) pv = Gh; else nx = Gh; }
      r[q] = pv + 2. * G + nx;
      shs[q][ty][tx] = r[q];
    }
    if (!HALO) {
#define RU(dk, dj, di) ring[dk][0][ty + (dj)][tx + 1 + (di)]
#define RV(dk, dj, di) ring[dk][1][ty + (dj)][tx + 1 + (di)]
#define RW(dk, dj, di) ring[dk][2][ty + (dj)][tx + 1 + (di)]
      // the thirty ring values are read group by group (the scheduling barriers keep the compiler from hoisting all LDS reads to the top, which costs
      // the fourteen registers this pass does not have at sixteen waves per block); expressions and their order as in k_strain_tile
      const real dxi = A.dxi, dyi = A.dyi, zc = ldc(A.dzci, k), zm = ldc(A.dzci, k - 1);
      const real u_ccc = RU(kc, 0, 0), u_mcc = RU(kc, 0, -1), v_ccc = RV(kc, 0, 0), v_cmc = RV(kc, -1, 0), w_ccc = RW(kc, 0, 0), w_ccm = RW(km, 0, 0);
      const real s11 = (u_ccc - u_mcc) * dxi, s22 = (v_ccc - v_cmc) * dyi, s33 = (w_ccc - w_ccm) * ldc(A.dzfi, k);
      real s12, s13, s23;
      { const real u_mmc = RU(kc, -1, -1), u_cmc = RU(kc, -1, 0), u_mpc = RU(kc, 1, -1), u_cpc = RU(kc, 1, 0);
        const real v_mmc = RV(kc, -1, -1), v_pmc = RV(kc, -1, 1), v_mcc = RV(kc, 0, -1), v_pcc = RV(kc, 0, 1);
        s12 = .125 * ((u_cpc - u_ccc) * dyi + (v_pcc - v_ccc) * dxi + (u_ccc - u_cmc) * dyi + (v_pmc - v_cmc) * dxi +
                      (u_mpc - u_mcc) * dyi + (v_ccc - v_mcc) * dxi + (u_mcc - u_mmc) * dyi + (v_cmc - v_mmc) * dxi); }
      __builtin_amdgcn_sched_barrier(0);
      { const real u_mcm = RU(km, 0, -1), u_ccm = RU(km, 0, 0), u_mcp = RU(kp, 0, -1), u_ccp = RU(kp, 0, 0);
        const real w_mcm = RW(km, 0, -1), w_pcm = RW(km, 0, 1), w_mcc = RW(kc, 0, -1), w_pcc = RW(kc, 0, 1);
        s13 = .125 * ((u_ccp - u_ccc) * zc + (w_pcc - w_ccc) * dxi + (u_ccc - u_ccm) * zm + (w_pcm - w_ccm) * dxi +
                      (u_mcp - u_mcc) * zc + (w_ccc - w_mcc) * dxi + (u_mcc - u_mcm) * zm + (w_ccm - w_mcm) * dxi); }
      __builtin_amdgcn_sched_barrier(0);
      { const real v_cmm = RV(km, -1, 0), v_ccm = RV(km, 0, 0), v_cmp = RV(kp, -1, 0), v_ccp = RV(kp, 0, 0);
        const real w_cmm = RW(km, -1, 0), w_cpm = RW(km, 1, 0), w_cmc = RW(kc, -1, 0), w_cpc = RW(kc, 1, 0);
        s23 = .125 * ((v_ccp - v_ccc) * zc + (w_cpc - w_ccc) * dyi + (v_ccc - v_ccm) * zm + (w_cpm - w_ccm) * dyi +
                      (v_cmp - v_cmc) * zc + (w_ccc - w_cmc) * dyi + (v_cmc - v_cmm) * zm + (w_ccm - w_cmm) * dyi); }
#undef RU
#undef RV
#undef RW
      const real s0v = sqrt(2. * (s11 * s11 + s22 * s22 + s33 * s33 + 2. * (s12 * s12 + s13 * s13 + s23 * s23)));
      stb(A.s0, idx, s0v);                                              // stands for visct = s0 (sgs.f90:184) until the final kernel
      const OFF i2 = 2 * idx;      // |S|Sij (sgs.f90:198-210), two components per 16-byte store
      typedef real v2 __attribute__((ext_vector_type(2)));
      __builtin_nontemporal_store(v2{s0v * s11, s0v * s22}, (v2 *)((char *)A.ss2[0] + i2));
      __builtin_nontemporal_store(v2{s0v * s33, s0v * s12}, (v2 *)((char *)A.ss2[1] + i2));
      __builtin_nontemporal_store(v2{s0v * s13, s0v * s23}, (v2 *)((char *)A.ss2[2] + i2));
      // (the cell-centred velocity is not stored: the last pass forms it from u, v, w itself -- the only form this pass serves, dsmag_fast)
      // EXT: v_c of the two ghost rows, which the last pass reads from A.vcg (row 0: its lower neighbour v(-1) is in this tile's ring, nowhere else);
      // every other row stores to its dump cell -- no branch around the store
      if (EXT) stb(A.vcg, cvc + (OFF)k * sk, 0.5 * (v_ccc + v_cmc));
    }
    __syncthreads();
    if (!HALO) {
#pragma unroll
      for (int q = 0; q < 3; ++q) { keep[q] = (shs[q][ty - 1][tx] + 2. * r[q] + shs[q][ty + 1][tx]) / 64.; stb(A.uf[q], idx, keep[q]); }
      stb(A.p, dstp, pl + sP[(k + 1) & 1][ty][tx]);
    }
    const int t = km; km = kc; kc = kp; kp = t;
  };
  const std::true_type T_; const std::false_type F_;
  const bool topw = !A.zper && kend == n3;      // the chunk's last plane sits under the upper wall
  const int klast = topw ? kend - 1 : kend;
  // (a wave's row is uniform: scalar branches, each loop straight-line code)
  // The first plane is peeled off: the loop is then entered with the same operations in flight as its back edge carries (the loads of a plane with
  // that plane's stores behind them), and the compiler's counted waits hold; entered from the prologue, whose last operation is a load, the merge of
  // the two states at the loop header makes every wait a wait for everything.
  if (__builtin_amdgcn_readfirstlane((int)hwave)) {
    if (kbeg <= klast) plane(kbeg, T_, F_);
    for (int k = kbeg + 1; k <= klast; ++k) plane(k, T_, F_);
    if (topw) plane(kend, T_, T_);
  } else {
    if (kbeg <= klast) plane(kbeg, F_, F_);
    for (int k = kbeg + 1; k <= klast; ++k) plane(k, F_, F_);
    if (topw) plane(kend, F_, T_);
  }
}
// Static Smagorinsky, row-marching form (default): one wave per row of 62 cells marching in k with everything in registers -- x neighbours by
// DPP lane moves, the rows j-1 and j+1 loaded again by this wave (they are their own waves' rows: cache hits), z neighbours rolled from plane
// to plane. No LDS and no barriers: the tile form above is bound by its barrier pair per plane with two square roots and an exponential per
// cell in between (0.26 of the HBM peak); here waves never wait for each other and the VGPR count alone sets the occupancy.
// (sgs.f90:98-152, 598-680)
constexpr int SROWS = 4;      // rows (waves) per block
// EX = 1: the block holds a row next to a wall-model y face (block-uniform: the other blocks of a duct run the loads of the channel form).
// Every global access of the plane loop is straight-line code ("In-order counters"): the extrapolated ghost row is c1 Q(o1) - c2 Q(o2)
// with per-thread offsets and weights instead of a branch around two loads, the shear of the nearer y wall is loaded for every plane whether the
// nearest wall turns out to be that one or a z wall -- the branches of the first version made each of the plane's loads a round trip of its own
// (0.50 ms per call for the 512 x 256 x 256 duct against 0.37 ms now).
template <typename OFF, int YW, int EX>
__device__ __forceinline__ void smag_rows_body(const Geom &g, const StrainTileArgs &A, const int bx, const int by, const int bz) {
  const int tx = threadIdx.x;
  const int i = bx * 62 + tx, j = by * SROWS + threadIdx.y + 1;
  if (j > g.n2) return;
  const int kbeg = bz * A.kchunk + 1, kend = min(kbeg + A.kchunk - 1, g.n3);
  const bool ldok = i <= g.n1 + 1, outok = tx >= 1 && tx <= 62 && i <= g.n1;
  const OFF sj = (OFF)g.s1 * RSZ, sk = (OFF)g.s12 * RSZ;
  const int iw = !A.perx ? i : i == 0 ? g.n1 : i == g.n1 + 1 ? 1 : i;      // periodic x: wrapped columns instead of the ghost columns
  const OFF c0 = ldok ? (OFF)g.ix(iw, j, 0) * RSZ : (OFF)g.ix(1, j, 0) * RSZ;
  // ghost rows at wall-model y faces: u and w are replaced by 2 Q(1) - Q(2) along y; v, normal to the face, is not (extrapolate(...,lwm), sgs.f90:683-748)
  const bool exlo = EX && A.wmylo && j == 1, exhi = EX && A.wmyhi && j == g.n2;
  // (lanes beyond the row read cell 0 of the field and planes beyond n3+1 are clamped: every load is unconditional, nothing branches around it)
  const OFF a1 = exlo ? c0 : c0 - sj, a2 = exlo ? c0 + sj : c0 - sj, b1 = exhi ? c0 : c0 + sj, b2 = exhi ? c0 - sj : c0 + sj;      // rows j-1 (a) and j+1 (b)
  const real ac1 = exlo ? 2. : 1., ac2 = exlo ? 1. : 0., bc1 = exhi ? 2. : 1., bc2 = exhi ? 1. : 0.;
  auto ldu = [&](int q, int dj, int k) -> real {      // q = 0 (u) or 2 (w), dj = -1, 0, +1
    const OFF o = (OFF)k * sk;
    if (EX && dj < 0) return ac1 * ldb(A.u[q], a1 + o) - ac2 * ldb(A.u[q], a2 + o);
    if (EX && dj > 0) return bc1 * ldb(A.u[q], b1 + o) - bc2 * ldb(A.u[q], b2 + o);
    return ldb(A.u[q], dj < 0 ? c0 + o - sj : dj > 0 ? c0 + o + sj : c0 + o);
  };
  auto ldv = [&](int dj, int k) -> real { return ldb(A.u[1], c0 + (OFF)k * sk - (dj < 0 ? sj : 0)); };
  // planes k-1, k, k+1 of u(j), v(j-1), v(j); plane k of u(j-1), u(j+1); planes k-1, k of w(j-1), w(j), w(j+1)
  real u0m = ldu(0, 0, kbeg - 1), u0c = ldu(0, 0, kbeg), u0p = ldu(0, 0, kbeg + 1);
  real vAm = ldv(-1, kbeg - 1), vAc = ldv(-1, kbeg), vAp = ldv(-1, kbeg + 1);
  real vCm = ldv(0, kbeg - 1), vCc = ldv(0, kbeg), vCp = ldv(0, kbeg + 1);
  real uA = ldu(0, -1, kbeg), uB = ldu(0, 1, kbeg);
  real wAm = ldu(2, -1, kbeg - 1), wCm = ldu(2, 0, kbeg - 1), wBm = ldu(2, 1, kbeg - 1);
  real wAc = ldu(2, -1, kbeg), wCc = ldu(2, 0, kbeg), wBc = ldu(2, 1, kbeg);
  // van Driest: wall units from the shear at the nearer z wall of this column (sgs.f90:117-143), read from the fields themselves
  // (their ghost cells, not the extrapolated ones)
  real tw_lo = 0., tw_hi = 0.;
  if (outok) {
    const real *u = A.u[0], *v = A.u[1];
    const int im1 = (A.perx && i == 1) ? g.n1 : i - 1;
    if (A.zlo) {
      const real t1 = u[g.ix(i, j, 1)] - u[g.ix(i, j, 0)] + u[g.ix(im1, j, 1)] - u[g.ix(im1, j, 0)];
      const real t2 = v[g.ix(i, j, 1)] - v[g.ix(i, j, 0)] + v[g.ix(i, j - 1, 1)] - v[g.ix(i, j - 1, 0)];
      tw_lo = sqrt(0.5 * A.visc * (sqrt(t1 * t1 + t2 * t2) * A.dzci[0]));
    }
    if (A.zhi) {
      const int n3 = g.n3;
      const real t1 = u[g.ix(i, j, n3)] - u[g.ix(i, j, n3 + 1)] + u[g.ix(im1, j, n3)] - u[g.ix(im1, j, n3 + 1)];
      const real t2 = v[g.ix(i, j, n3)] - v[g.ix(i, j, n3 + 1)] + v[g.ix(i, j - 1, n3)] - v[g.ix(i, j - 1, n3 + 1)];
      tw_hi = sqrt(0.5 * A.visc * (sqrt(t1 * t1 + t2 * t2) * A.dzci[n3]));
    }
  }
  const real dxi = A.dxi, dyi = A.dyi;
  // van Driest with y walls: the nearer y wall of this row (the first one wins a tie) is known before the loop; its shear plane is read for every k
  const int jg = j + g.jlo;
  real dminy = YW && A.wylo ? A.dl2 * (jg - 0.5) : CALES_BIG; int locy = 2;
  { const real d = YW && A.wyhi ? A.dl2 * (g.ng2 - jg + 0.5) : CALES_BIG; if (d < dminy) { dminy = d; locy = 3; } }
  const real *twp = (YW && A.twy) ? A.twy + (size_t)(locy == 3 ? g.n3 + 2 : 0) * g.s1 + (ldok ? i : 1) : A.del;
  const int tws = (YW && A.twy) ? g.s1 : 0;
  real twyc = YW ? twp[(size_t)kbeg * tws] : 0.;
  for (int k = kbeg; k <= kend; ++k) {
    // next iteration's planes, in flight during this one's arithmetic
    const int k2 = min(k + 2, g.n3 + 1), k1 = k + 1;
    const real twyn = YW ? twp[(size_t)k1 * tws] : 0.;
    const real u0n = ldu(0, 0, k2), vAn = ldv(-1, k2), vCn = ldv(0, k2);
    const real uAn = ldu(0, -1, k1), uBn = ldu(0, 1, k1);
    const real wAn = ldu(2, -1, k1), wCn = ldu(2, 0, k1), wBn = ldu(2, 1, k1);
    // wall-model z faces: ghost planes of u, v by extrapolation with the grid factor
    real u_ccm = u0m, v_cmm = vAm, v_ccm = vCm, u_ccp = u0p, v_cmp = vAp, v_ccp = vCp;
    if (A.wmlo && k == 1) { u_ccm = (1. + A.flo) * u0c - A.flo * u0p; v_cmm = (1. + A.flo) * vAc - A.flo * vAp; v_ccm = (1. + A.flo) * vCc - A.flo * vCp; }
    if (A.wmhi && k == g.n3) { u_ccp = (1. + A.fhi) * u0c - A.fhi * u0m; v_cmp = (1. + A.fhi) * vAc - A.fhi * vAm; v_ccp = (1. + A.fhi) * vCc - A.fhi * vCm; }
    const real u_ccc = u0c, u_cmc = uA, u_cpc = uB, v_cmc = vAc, v_ccc = vCc;
    const real w_cmm = wAm, w_ccm = wCm, w_cpm = wBm, w_cmc = wAc, w_ccc = wCc, w_cpc = wBc;
    const real u_mcm = lane_prev(u_ccm), u_mcc = lane_prev(u_ccc), u_mcp = lane_prev(u_ccp), u_mmc = lane_prev(u_cmc), u_mpc = lane_prev(u_cpc);
    const real v_mmc = lane_prev(v_cmc), v_pmc = lane_next(v_cmc), v_mcc = lane_prev(v_ccc), v_pcc = lane_next(v_ccc);
    const real w_mcm = lane_prev(w_ccm), w_pcm = lane_next(w_ccm), w_mcc = lane_prev(w_ccc), w_pcc = lane_next(w_ccc);
    const real zc = ldc(A.dzci, k), zm = ldc(A.dzci, k - 1);
    const real s11 = (u_ccc - u_mcc) * dxi, s22 = (v_ccc - v_cmc) * dyi, s33 = (w_ccc - w_ccm) * ldc(A.dzfi, k);
    const real s12 = .125 * ((u_cpc - u_ccc) * dyi + (v_pcc - v_ccc) * dxi + (u_ccc - u_cmc) * dyi + (v_pmc - v_cmc) * dxi +
                               (u_mpc - u_mcc) * dyi + (v_ccc - v_mcc) * dxi + (u_mcc - u_mmc) * dyi + (v_cmc - v_mmc) * dxi);
    const real s13 = .125 * ((u_ccp - u_ccc) * zc + (w_pcc - w_ccc) * dxi + (u_ccc - u_ccm) * zm + (w_pcm - w_ccm) * dxi +
                               (u_mcp - u_mcc) * zc + (w_ccc - w_mcc) * dxi + (u_mcc - u_mcm) * zm + (w_ccm - w_mcm) * dxi);
    const real s23 = .125 * ((v_ccp - v_ccc) * zc + (w_cpc - w_ccc) * dyi + (v_ccc - v_ccm) * zm + (w_cpm - w_ccm) * dyi +
                               (v_cmp - v_cmc) * zc + (w_ccc - w_cmc) * dyi + (v_cmc - v_cmm) * zm + (w_ccm - w_cmm) * dyi);
    const real s0v = sqrt(2. * (s11 * s11 + s22 * s22 + s33 * s33 + 2. * (s12 * s12 + s13 * s13 + s23 * s23)));
    if (outok) {
      real fd = 1.;
      if (A.zlo || A.zhi || (YW && (A.wylo || A.wyhi))) {     // nearest wall in the order y-, y+, z-, z+: the first one wins a tie (minloc, sgs.f90:116)
        real dmin = dminy; int loc = locy;
        { const real d = A.zlo ? ldc(A.zc, k) : CALES_BIG; if (d < dmin) { dmin = d; loc = 4; } }
        { const real d = A.zhi ? A.l3 - ldc(A.zc, k) : CALES_BIG; if (d < dmin) { dmin = d; loc = 5; } }
        const real tw = loc < 4 ? twyc : loc == 4 ? tw_lo : tw_hi;
        const real dw_plus = dmin * tw * (1. / A.visc);
        fd = 1. - exp(-dw_plus / 25.);
      }
      const real t = 0.11 * ldc(A.del, k) * fd;      // c_smag, src/param.f90:33
      stb(A.visct, c0 + (OFF)k * sk, (t * t) * s0v);
    }
    u0m = u0c; u0c = u0p; u0p = u0n; vAm = vAc; vAc = vAp; vAp = vAn; vCm = vCc; vCc = vCp; vCp = vCn;
    uA = uAn; uB = uBn; wAm = wAc; wCm = wCc; wBm = wBc; wAc = wAn; wCc = wCn; wBc = wBn;
    twyc = twyn;
  }
}
template <typename OFF, int YW>
// (four waves per SIMD -- 128 VGPRs -- for channels; the duct logic needs a few registers more and would spill under that cap)
__global__ __launch_bounds__(64 * SROWS, (YW ? 3 : 4)) void k_smag_rows(Geom g, StrainTileArgs A) {
  // band map (common.hpp): the rows two waves both load (j-1, j+1, the shared columns of neighbouring x tiles) meet in one L2 instead of
  // being fetched from memory by two (measured 1.8 x the compulsory reads with the plain 3-D grid, 1.27 x with the bands)
  int bx, by, bz;
  if (!band_block(A.bm, bx, by, bz)) return;
  if (YW && ((A.wmylo && by == 0) || (A.wmyhi && (by + 1) * SROWS >= g.n2))) smag_rows_body<OFF, YW, 1>(g, A, bx, by, bz);
  else smag_rows_body<OFF, YW, 0>(g, A, bx, by, bz);
}
// p1d[which*n3 + k-1] = sum over the blocks' partials, fixed order (ave1d_channel, sgs.f90:462-472)
__global__ __launch_bounds__(256) void k_plane_fold(int n3, int nblk, const real *__restrict__ part, real *__restrict__ p1d) {
  __shared__ real sh[4];
  const real *p = part + (size_t)blockIdx.x * nblk;
  real acc = 0.;
  for (int q = threadIdx.x; q < nblk; q += 256) acc += p[q];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) p1d[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
  (void)n3;
}

int allreduce_res(cales_ctx *c, int slot, int count, int op);
int op_boundp(cales_ctx *c, real *p, int which);
int op_boundp_multi(cales_ctx *c, int nf, real **p, int which);
// v_c = (v(j) + v(j-1))/2 of the rows j = 1 and j = n2 (all i, k, ghost columns and planes included): the two rows whose copies the ghost-cell update /
// the slab exchange puts into the ghost rows 0 and n2+1 of `vc` -- k_lmf_tile<.., UCF = 1> reads row 0 from there
__global__ __launch_bounds__(256) void k_vc_edge_rows(Geom g, const real *__restrict__ v, real *__restrict__ vc) {
  const int i = blockIdx.x * 64 + threadIdx.x, k = blockIdx.y * 4 + threadIdx.y;
  if (i > g.n1 + 1 || k > g.n3 + 1) return;
  vc[g.ix(i, 1, k)] = 0.5 * (v[g.ix(i, 1, k)] + v[g.ix(i, 0, k)]);
  vc[g.ix(i, g.n2, k)] = 0.5 * (v[g.ix(i, g.n2, k)] + v[g.ix(i, g.n2 - 1, k)]);
}
// z faces of the normal velocity after a projection folded into k_corr_strain_tile (which writes the planes 1..n3): plane 0 is corrected like
// every other (correc.f90:60-66: k = 0..n3, ghost rows and columns included), plane n3+1 keeps the value the last bounduvw gave the prediction
__global__ __launch_bounds__(256) void k_wface_fold(Geom g, const real *__restrict__ ws, real *__restrict__ wd, const real *__restrict__ pp, real cz0) {
  const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
  if (i > g.n1 + 1 || j > g.n2 + 1) return;
  const size_t a = g.ix(i, j, 0), b = g.ix(i, j, g.n3 + 1);
  wd[a] = ws[a] - cz0 * (pp[a + g.s12] - pp[a]);
  wd[b] = ws[b];
}
// Several slabs, projection folded into the strain-rate pass: the corrected velocity and p + pp of the two GHOST rows (the neighbours' rows n2 and 1) are
// formed HERE, from the prediction's ghost rows and pp's -- the operands the neighbour's own pass has for those rows, the operations of `fix` in
// k_corr_strain_tile (correc.f90:44-67, updatep.f90:30-47) in their order -- instead of travelling: four field planes per substep less in the slab
// exchange (63 -> 51 planes per step). pp of "row n2+2" for v in the upper ghost row: the companion field (api.hip). Interior
// cells i = 1..n1, k = 1..n3 of the two rows; their x and z ghost cells follow from the ghost-cell kernel, whose loops cover the ghost rows, and
// plane 0 of w from k_wface_fold.
__global__ __launch_bounds__(256) void k_fold_ghost_rows(Geom g, const real *__restrict__ us, const real *__restrict__ vs, const real *__restrict__ ws,
                                                         real *__restrict__ ud, real *__restrict__ vd, real *__restrict__ wd, const real *__restrict__ pp,
                                                         const real *__restrict__ ppc, real *__restrict__ p, const real *__restrict__ force, int fmask,
                                                         real cfi, real cfj, real cdt, const real *__restrict__ dzci, int perx) {
  const int i = blockIdx.x * 64 + threadIdx.x + 1, k = blockIdx.y * 4 + threadIdx.y + 1, side = blockIdx.z;
  if (i > g.n1 || k > g.n3) return;
  const int j = side ? g.n2 + 1 : 0, ip = (perx && i == g.n1) ? 1 : i + 1;
  const size_t a = g.ix(i, j, k);
  const real P0 = pp[a], px = pp[g.ix(ip, j, k)], py = side ? ppc[a] : pp[a + g.s1], pz = pp[a + g.s12];
  const real f0 = (fmask & 1) ? force[0] : 0., f1 = (fmask & 2) ? force[1] : 0., f2 = (fmask & 4) ? force[2] : 0.;
  ud[a] = ((fmask & 1) ? us[a] + f0 : us[a]) - cfi * (px - P0);
  vd[a] = ((fmask & 2) ? vs[a] + f1 : vs[a]) - cfj * (py - P0);
  wd[a] = ((fmask & 4) ? ws[a] + f2 : ws[a]) - cdt * dzci[k] * (pz - P0);
  p[a] = p[a] + P0;
}
static bool dsmag_fast_ok(const cales_ctx *c) {
  for (int q = 0; q < 2; ++q) if (c->is_wall[q] != 0. || c->C.lwm[q] != 0) return false;      // walls or wall model in x: general path
  // walls / wall model in y (ducts): the fused last pass knows the wall rule along y, the two-pass form does not
  for (int q = 2; q < 4; ++q) if ((c->is_wall[q] != 0. || c->C.lwm[q] != 0) && c->n[1] < 3) return false;
  return c->n[2] >= 3 && !c->fl.dsmag_reference_sequence;
}
// |S|Sij as three fields of pairs between K_AC and the fused last pass: x and y periodic (the one-launch ghost-cell kernel takes a pair field as a
// field of twice the width), the cell-centred velocity formed by the last pass, 32-bit byte offsets still enough for a field twice as long
bool dsmag_pairs(const cales_ctx *c) {
  if (c->C.sgstype != 2 || !dsmag_fast_ok(c) || c->fl.dsmag_xghosts || c->fl.wide_offsets || c->fl.unmerged_bc) return false;
  for (int q = 0; q < 4; ++q) if (c->C.cbcpre[q] != 'P') return false;
  const bool perz = c->C.cbcpre[4] == 'P' && c->C.cbcpre[5] == 'P';
  if (!perz && !(c->is_wall[4] != 0. && c->is_wall[5] != 0.)) return false;      // z: periodic, or two walls (whose ghost planes the filters never read)
  return (2 * c->ntot + 64) * sizeof(real) < (1ull << 32);
}
static int dsmag_fast(cales_ctx *c) {
  const int *n = c->n; real **f = c->f; real *visct = f[CALES_VISCT];
  const bool pair = c->ss2[0] != nullptr;
  dim3 b(BX, BY, 1), gr = grid3(n[0], n[1], n[2], b);
  real **ssij = c->sij, **mij = c->mij;
  const int zlo = c->is_wall[4] != 0., zhi = c->is_wall[5] != 0.;
  // wall-model faces in z: the strain rates see ghost planes extrapolated from the interior (extrapolate(...,lwm) with the
  // grid factor, sgs.f90:683-748) instead of the stress-carrying ghost cells
  const int wmlo = ISB(c, 0, 3) && LWM(c, 0, 3) != 0, wmhi = ISB(c, 1, 3) && LWM(c, 1, 3) != 0;
  const real flo = (1. / c->dzci[0]) * c->dzci[1], fhi = (1. / c->dzci[n[2]]) * c->dzci[n[2] - 1];
  // y walls of a duct, on the rank that owns them: wall rule of the filters along y; wall-model y faces: extrapolated ghost rows for the strain rates
  const int wylo = ISB(c, 0, 2) && c->is_wall[2] != 0., wyhi = ISB(c, 1, 2) && c->is_wall[3] != 0.;
  const int wmylo = ISB(c, 0, 2) && LWM(c, 0, 2) != 0, wmyhi = ISB(c, 1, 2) && LWM(c, 1, 2) != 0;
  // tiles of 62 x TY columns marching in k; k is also split into chunks so that several rounds of blocks balance the chip
  auto tiles = [&](int ty, int wx, dim3 &mb, dim3 &mg, int &kchunk, int kmax = 1 << 30, int rows_more = 0) {
    mb = dim3(64, ty + 2, 1); mg = dim3((n[0] + wx - 1) / wx, (n[1] + rows_more + ty - 1) / ty, 1);
    kchunk = n[2];
    while ((long)mg.x * mg.y * ((n[2] + kchunk - 1) / kchunk) < tile_min_blocks(c) && kchunk > 32) kchunk = (kchunk + 1) / 2;
    // small grids: fewer blocks than one per CU leave most of the chip idle; shorter chunks (their three-plane prologue weighs more) beat that
    while ((long)mg.x * mg.y * ((n[2] + kchunk - 1) / kchunk) < 256 && kchunk > SMALL_KCH) kchunk = (kchunk + 1) / 2;
    kchunk = balanced_kchunk(c, (long)mg.x * mg.y, n[2], kchunk, kmax);
    if (int fk = tile_kchunk(c, (long)mg.x * mg.y, n[2])) kchunk = fk;
    mg.z = (n[2] + kchunk - 1) / kchunk;
  };
  dim3 mb, mg; int kch;
  const bool small = (c->ntot + 16) * sizeof(real) < (1ull << 32) && !c->fl.wide_offsets;      // 32-bit byte offsets (ldb/stb)
  // lazy form (homogeneous sgs BCs): |S| goes straight into the eddy-viscosity field and the last pass only makes the n3 plane coefficients
  bool lazy = true;
  for (int q = 0; q < 6; ++q) lazy = lazy && c->C.bcsgs[q] == 0.;
  // the cell-centred velocity is not stored where the last pass can form it itself (k_lmf_tile<.., UCF = 1>): x periodic, z walls or periodic
  const bool perz = CBP(c, 0, 3) == 'P' && CBP(c, 1, 3) == 'P';
  const bool ucf = CBP(c, 0, 1) == 'P' && CBP(c, 1, 1) == 'P' && !c->fl.dsmag_xghosts && ((zlo && zhi) || perz);
  // K_AC: |S|, |S|Sij, cell-centred and test-filtered velocity in one pass over u,v,w (no wall-model faces ->
  // extrapolate(...,lwm) is a no-op; u,v are extrapolated through the z walls, w on the faces is not, sgs.f90:705-710)
  const bool fold = c->fold_dtrk != 0. && pair && ucf;      // (cales_step decides; pair fields and the cell-centred velocity formed by the last pass: the instantiation that exists)
  const bool ext = fold && c->fold_rows2 && c->P > 1;      // ... which also forms the ghost rows of its outputs from two ghost rows of the prediction (StepPlan::fold_rows2)
  if (c->fold_dtrk != 0. && !fold) { c->err = "dsmag: projection folded into the strain-rate pass without pair fields"; return 1; }
  { ProfScope ps(c, fold ? "correc_strain_filter_uvw" : "strain_filter_uvw");
    tiles(fold ? TYC : TYS, 64, mb, mg, kch, 1 << 30, ext ? 2 : 0);
    StrainTileArgs S;
    S.u[0] = f[CALES_U]; S.u[1] = f[CALES_V]; S.u[2] = f[CALES_W]; S.s0 = lazy ? visct : c->s0;
    for (int m = 0; m < 6; ++m) S.ssij[m] = ssij[m];
    for (int m = 0; m < 3; ++m) S.ss2[m] = reinterpret_cast<real2 *>(c->ss2[m]);
    S.uc[0] = ucf ? nullptr : c->uc; S.uc[1] = ucf ? nullptr : c->vc; S.uc[2] = ucf ? nullptr : c->wc; S.uf[0] = c->uf; S.uf[1] = c->vf; S.uf[2] = c->wf;
    S.dzci = c->d_dzci; S.dzfi = c->d_dzfi; S.dxi = c->dli[0]; S.dyi = c->dli[1]; S.kchunk = kch; S.zlo = zlo; S.zhi = zhi; S.wmlo = wmlo; S.wmhi = wmhi; S.flo = flo; S.fhi = fhi;
    S.wylo = wylo; S.wyhi = wyhi; S.wmylo = wmylo; S.wmyhi = wmyhi; S.twy = nullptr; S.dl2 = c->dl[1];
    S.bm = BandMap{0, 0, 0, 0}; S.perx = c->step_xskip ? 1 : 0;
    if (band_wanted(mg.x)) { S.bm = band_map(mg.x, mg.y, mg.z); mg = dim3(band_blocks(S.bm), 1, 1); }
    if (fold) {      // the projection of this substep is pending (cales_step): corrected velocity on load, u, v, w to the second buffers, p += pp
      S.pp = f[CALES_PP]; S.p = f[CALES_P]; for (int q = 0; q < 3; ++q) S.un[q] = c->f2[q];
      S.force = c->d_force; S.fmask = c->defer_force ? (c->C.is_forced[0] ? 1 : 0) | (c->C.is_forced[1] ? 2 : 0) | (c->C.is_forced[2] ? 4 : 0) : 0;
      S.cdt = c->fold_dtrk; S.cfi = c->fold_dtrk * c->dli[0]; S.cfj = c->fold_dtrk * c->dli[1]; S.zper = perz ? 1 : 0;
      S.pery = c->P == 1 ? 1 : 0; S.ppd = c->pp_companion_bytes;
      const size_t pl = (size_t)(n[0] + 2) * (n[1] + 2);
      S.bcz[0][0] = c->bcu.z; S.bcz[0][1] = c->bcu.z + pl; S.bcz[1][0] = c->bcv.z; S.bcz[1][1] = c->bcv.z + pl;
      S.vcg = c->vc;
      if (ext) LAUNCH(c, (k_corr_strain_tile<unsigned, TYC, 1>), mg, mb, 0, c->stream, c->g, S);
      else LAUNCH(c, (k_corr_strain_tile<unsigned, TYC>), mg, mb, 0, c->stream, c->g, S);
    } else
    if (wylo || wyhi || wmylo || wmyhi) { if (small) LAUNCH(c, (k_strain_tile<unsigned, TYS, 1>), mg, mb, 0, c->stream, c->g, S); else LAUNCH(c, (k_strain_tile<size_t, TYS, 1>), mg, mb, 0, c->stream, c->g, S); }
    else if (pair) LAUNCH(c, (k_strain_tile<unsigned, TYS, 0, 1>), mg, mb, 0, c->stream, c->g, S);
    else if (small) LAUNCH(c, (k_strain_tile<unsigned, TYS, 0>), mg, mb, 0, c->stream, c->g, S); else LAUNCH(c, (k_strain_tile<size_t, TYS, 0>), mg, mb, 0, c->stream, c->g, S); }
  if (fold) {
    // the corrected velocity sits in the second buffers: swap (as the fused momentum pass does), give the normal component its two z faces -- the
    // reference's correc covers w(:,:,0) (correc.f90:60-66) and leaves w(:,:,n3+1) as it was, and bounduvw with is_correc touches neither -- and
    // fill the ghost cells of u, v, w and p (the pressure rides along) as main.f90:500-504 does after correc / updatep
    for (int q = 0; q < 3; ++q) std::swap(c->f[CALES_U + q], c->f2[q]);
    if (!perz)
      LAUNCH(c, k_wface_fold, dim3((n[0] + 2 + 63) / 64, (n[1] + 2 + 3) / 4), dim3(64, 4), 0, c->stream, c->g, c->f2[2], c->f[CALES_W], f[CALES_PP], c->fold_dtrk * c->dzci[0]);
    if (!c->fl.unmerged_bc) { c->bc_nride = 1; c->bc_ride[0] = f[CALES_P]; c->bc_ride_which[0] = 0; }
    // several slabs: the ghost rows of u, v, w and p do not travel at all -- k_fold_ghost_rows forms them from the prediction's ghost rows and pp's,
    // the ghost-cell kernel below gives them their x and z ghost cells (round 5: in the same exchange as the scratch fields', fifteen planes; now eleven)
    const bool local_rows = c->P > 1 && !c->fl.unmerged_bc;
    if (local_rows && !ext)      // (ext: the strain-rate pass has formed them itself)
      LAUNCH(c, k_fold_ghost_rows, dim3((n[0] + 63) / 64, (n[2] + 3) / 4, 2), dim3(64, 4), 0, c->stream, c->g, c->f2[0], c->f2[1], c->f2[2], c->f[CALES_U], c->f[CALES_V],
             c->f[CALES_W], f[CALES_PP], c->scr2, f[CALES_P], c->d_force, c->defer_force ? (c->C.is_forced[0] ? 1 : 0) | (c->C.is_forced[1] ? 2 : 0) | (c->C.is_forced[2] ? 4 : 0) : 0,
             c->fold_dtrk * c->dli[0], c->fold_dtrk * c->dli[1], c->fold_dtrk, c->d_dzci, c->step_xskip ? 1 : 0);
    const bool no_halo_before = c->bc_no_halo;
    c->bc_no_halo = no_halo_before || local_rows;
    const int e = op_bounduvw(c, c->bcu, c->bcv, c->bcw, 1, 1, c->f[CALES_U], c->f[CALES_V], c->f[CALES_W]);
    const bool rode = !c->fl.unmerged_bc && c->bc_nride == 0; c->bc_nride = 0;
    int e2 = 0;
    if (!e && !rode) e2 = op_boundp(c, f[CALES_P], 0);
    c->bc_no_halo = no_halo_before;
    if (e || e2) { c->deferred.clear(); c->deferred_wide.clear(); return e ? e : e2; }
  }
  // sgs-type ghost cells: only the periodic exchange matters (products of ghosts = ghosts of products; the wall ghosts are
  // replaced by the extrapolation rule inside the filters)
  // These twelve scratch fields are read by the tile kernels only: with periodic x their ghost columns are not filled (the
  // kernels wrap around instead; an x ghost update touches four cache lines per row for two values), and the z ghost planes
  // of the quantities the wall rule covers are never read.
  const int perx = (CBP(c, 0, 1) == 'P' && CBP(c, 1, 1) == 'P' && !c->fl.dsmag_xghosts) ? 1 : 0;
  const int skipz = (zlo && zhi) ? 4 : 0;
  // several ranks with the second stream: the y-halo rows of the twelve scratch fields travel while the interior tiles of the last pass run
  const bool overlap = c->P > 1 && c->comm.halo_s && c->comm_stream && !ext;      // (ext: no row of these fields travels)
  // (every check that can fail comes BEFORE the deferred exchange is queued: an error return behind halo_flush_deferred would leave the
  //  exchange in flight on the second stream with nobody joining it)
  const int lmf_ty = (wylo || wyhi || wmylo || wmyhi) ? TYL : TYLF;      // tile height of the fused last pass (k_lmf_tile<.., YW>)
  { dim3 tb, tg; int tk; tiles(lmf_ty, 62, tb, tg, tk);
    if ((size_t)2 * n[2] * tg.x * tg.y > c->ntot) { c->err = "dsmag: partial-sum scratch too small"; return 1; } }
  // several ranks: ONE exchange for the y-halo rows of all these fields (six |S|Sij, three filtered velocities, v_c, and |S| itself in the lazy form
  // inside cales_step) instead of one per ghost-cell call; their ghost-cell kernels run first, the rows that then arrive carry the neighbour's
  // x/z ghost cells
  c->defer_halo = c->P > 1 && !ext;
  const bool no_halo_before = c->bc_no_halo;
  if (ext) c->bc_no_halo = true;      // the ghost rows of every field below were formed by the strain-rate pass: their x and z ghost cells only
  c->bc_skip = perx | skipz;
  int e_ = pair ? op_boundp_wide(c, 3, c->ss2, 1) : op_boundp_multi(c, 6, ssij, 1);
  c->bc_skip = perx;
  if (!e_) e_ = op_bounduvw(c, c->bcuf, c->bcvf, c->bcwf, 0, 0, c->uf, c->vf, c->wf);
  c->bc_skip = perx | skipz;
  if (!e_ && !ucf) { real *cc[3] = {c->uc, c->vc, c->wc}; e_ = op_boundp_multi(c, 3, cc, 1); }
  if (!e_ && ucf && ext) { real *cc[1] = {c->vc}; e_ = op_boundp_multi(c, 1, cc, 1); }      // (v_c of the two ghost rows came from the strain-rate pass)
  else if (!e_ && ucf && !(wylo && wyhi)) {      // v_c of the rows 1 and n2 only: their copies in the ghost rows (periodic wrap or the slab neighbours') are what the last pass reads for row 0
    LAUNCH(c, k_vc_edge_rows, dim3((n[0] + 2 + 63) / 64, (n[2] + 2 + 3) / 4), dim3(64, 4), 0, c->stream, c->g, f[CALES_V], c->vc);
    real *cc[1] = {c->vc}; e_ = op_boundp_multi(c, 1, cc, 1); }
  c->bc_skip = 0;
  if (!e_ && lazy && c->in_step && c->P > 1) { e_ = op_boundp(c, visct, 1); c->visct_bc_done = !e_; }      // |S| is final (K_AC wrote it): its rows travel along
  c->defer_halo = false;
  c->bc_no_halo = no_halo_before;
  if (!e_ && c->P > 1) e_ = halo_flush_deferred(c, overlap);
  if (e_) { c->deferred.clear(); c->deferred_wide.clear(); return e_; }
  LijMijArgs L;
  L.uc[0] = ucf ? f[CALES_U] : c->uc; L.uc[1] = ucf ? f[CALES_V] : c->vc; L.uc[2] = ucf ? f[CALES_W] : c->wc; L.uf[0] = c->uf; L.uf[1] = c->vf; L.uf[2] = c->wf;
  L.vcg = c->vc; L.perz = perz ? 1 : 0; L.xwrap = c->step_xskip ? 1 : 0;
  for (int m = 0; m < 6; ++m) L.mf[m] = mij[m];
  L.part = c->wk[0]; L.dzci = c->d_dzci; L.dzfi = c->d_dzfi; L.dxi = c->dli[0]; L.dyi = c->dli[1];
  L.zlo = zlo; L.zhi = zhi; L.wmlo = wmlo; L.wmhi = wmhi; L.flo = flo; L.fhi = fhi; L.perx = perx;
  L.wylo = wylo; L.wyhi = wyhi; L.wmylo = wmylo; L.wmyhi = wmyhi;
  {
    // K_B + K_DF in one pass: filter(|S|Sij) on the fly, strain rate of the filtered velocity, Mij, Lij, contractions, plane partial sums
    ProfScope ps(c, "lij_mij_filter_contract");
    tiles(lmf_ty, 62, mb, mg, kch, LMF_KMAX);
    while (kch > LMF_KMAX) { kch = (kch + 1) / 2; mg.z = (n[2] + kch - 1) / kch; }      // the kernel keeps a chunk's block sums in LDS
    L.kchunk = kch; L.nblk = mg.x * mg.y;
    LmfArgs B; B.L = L; for (int m = 0; m < 6; ++m) B.ss[m] = ssij[m];
    for (int m = 0; m < 3; ++m) B.ss2[m] = reinterpret_cast<const real2 *>(c->ss2[m]);
    auto launch = [&](int by0, int nby) {
      if (nby <= 0) return;
      B.by0 = by0; B.gx = mg.x; dim3 gg(mg.x, nby, mg.z);
      B.bm = BandMap{0, 0, 0, 0};
      if (band_wanted(mg.x)) { B.bm = band_map(mg.x, nby, mg.z); gg = dim3(band_blocks(B.bm), 1, 1); }
      const bool yw = wylo || wyhi || wmylo || wmyhi;
#define LMF_LAUNCH(YWV, UCV) do { if (small) LAUNCH(c, (k_lmf_tile<unsigned, YWV, UCV>), gg, mb, 0, c->stream, c->g, B); else LAUNCH(c, (k_lmf_tile<size_t, YWV, UCV>), gg, mb, 0, c->stream, c->g, B); } while (0)
      if (pair) { if (ucf) LAUNCH(c, (k_lmf_tile<unsigned, 0, 1, 1>), gg, mb, 0, c->stream, c->g, B); else LAUNCH(c, (k_lmf_tile<unsigned, 0, 0, 1>), gg, mb, 0, c->stream, c->g, B); }
      else if (yw) { if (ucf) LMF_LAUNCH(1, 1); else LMF_LAUNCH(1, 0); } else { if (ucf) LMF_LAUNCH(0, 1); else LMF_LAUNCH(0, 0); }
#undef LMF_LAUNCH
    };
    if (overlap) {
      // tiles that read the ghost rows j = 0 or j = n2+1 wait for the rows in flight; the others run beside the exchange
      int hi0 = (int)mg.y; while (hi0 > 1 && (hi0 - 1) * lmf_ty + lmf_ty + 1 >= n[1] + 1) --hi0;      // first tile (> 0) that reaches row n2+1
      launch(1, hi0 - 1);
      if (int e = stream_after(c, c->stream, c->comm_stream)) return e;
      launch(0, 1); launch(hi0, (int)mg.y - hi0);
    } else launch(0, (int)mg.y);
    LAUNCH(c, k_plane_fold, dim3(2 * n[2]), dim3(256), 0, c->stream, n[2], L.nblk, c->wk[0], c->d_p1d);
  }
  if (c->P > 1) { if (int e = allreduce_res(c, (int)(c->d_p1d - c->res), 2 * n[2], 0)) return e; }   // sgs.f90:475
  const real gar = c->dl[0] * c->dl[1] / (c->C.l[0] * c->C.l[1]);
  if (lazy) {
    if (!c->d_cs) HIPCHK(c, hipMalloc(&c->d_cs, (n[2] + 2) * sizeof(real)));
    LAUNCH(c, k_dsmag_coef, dim3(1), dim3(256), 0, c->stream, n[2], gar, c->d_p1d, c->d_cs, CBP(c, 0, 3) == 'P' && CBP(c, 1, 3) == 'P' ? 1 : 0);
    c->visct_lazy = true;
  } else LAUNCH(c, k_dsmag_final, gr, b, 0, c->stream, c->g, gar, c->d_p1d, c->s0, visct);
  LAUNCHCHK(c);
  return 0;
}

static inline dim3 lin_grid(size_t n) { size_t b = (n + 255) / 256; if (b > 4096) b = 4096; return dim3((unsigned)b); }

// Static Smagorinsky for cases whose only walls are in z (channels, with or without wall model): strain rate and van Driest
// damping in one pass of the tile kernel, u,v,w -> visct (4 words/cell instead of the 20 of copy + extrapolate + strain + smag)
static bool smag_fast_ok(const cales_ctx *c) {
  for (int q = 0; q < 2; ++q) if (c->is_wall[q] != 0. || c->C.lwm[q] != 0) return false;      // walls or wall model in x: general path
  return c->n[2] >= 3 && c->n[1] >= 2 && !c->fl.smag_reference_sequence;
}
__global__ void k_smag_del(int n, real dl1, real dl2, const real *__restrict__ dzf, real *__restrict__ del) {
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k < n) del[k] = pow(dl1 * dl2 * dzf[k], 1. / 3.);       // the filter width depends on k only (sgs.f90:145)
}
// sqrt(tau_w) of the two y walls as planes twy(side, k, i) for the van Driest damping of cells whose nearest wall is a y wall (sgs.f90:117-143,
// cases 3 and 4). One slab: straight from the fields. Several slabs: the walls belong to the first and the last slab, every other rank needs
// their shear too -- the reference stays within "two subdomains between two opposite walls" (sanity.f90:98-111) because its 2-D pencil grid
// can (initmpi.f90:230-259); y slabs cannot, so the owners fill their plane, everybody else zeros, and one sum over the slabs (x + 0 + ... + 0:
// exact) hands both planes to every rank. The planes sit at the head of the staging buffer A, which is free between two exchanges.
static int wall_shear_y_planes(cales_ctx *c, int wylo, int wyhi, const real **out) {
  const int *n = c->n;
  const size_t cnt = (size_t)2 * (n[2] + 2) * c->g.s1;
  real *twy = c->wk[0];
  if (c->P > 1) {
    if (!c->comm.on) { c->err = "nranks > 1 but no communication hooks registered (cales_set_comm)"; return 1; }
    if ((int64_t)cnt > c->res - c->comm.A) { c->err = "smag: staging buffer too small for the wall-shear planes"; return 1; }
    twy = c->comm.A;
    if (c->comm_stream) if (int e = stream_after(c, c->stream, c->comm_stream)) return e;      // nothing of an overlapped exchange may still use A
    HIPCHK(c, hipMemsetAsync(twy, 0, cnt * sizeof(real), c->stream));
  } else if (cnt > c->ntot) { c->err = "smag: wall-shear scratch too small"; return 1; }
  const int lo = wylo && ISB(c, 0, 2) ? 1 : 0, hi = wyhi && ISB(c, 1, 2) ? 1 : 0;
  if (lo || hi)
    LAUNCH(c, k_wall_shear_y, dim3((n[0] + 63) / 64, (n[2] + 3) / 4), dim3(64, 4), 0, c->stream, c->g, c->f[CALES_U], c->f[CALES_W], c->visc, c->dli[1], lo, hi, twy, c->step_xskip ? 1 : 0);
  LAUNCHCHK(c);
  if (c->P > 1 && c->comm.allred(c->comm.user, 0, (int64_t)cnt, 0)) { c->err = "allreduce callback failed (wall-shear planes)"; return 1; }
  *out = twy;
  return 0;
}
static int smag_fast(cales_ctx *c) {
  const int *n = c->n; real **f = c->f;
  if (!c->d_del) {
    HIPCHK(c, hipMalloc(&c->d_del, (n[2] + 2) * sizeof(real)));
    LAUNCH(c, k_smag_del, dim3((n[2] + 2 + 63) / 64), dim3(64), 0, c->stream, n[2] + 2, c->dl[0], c->dl[1], c->d_dzf, c->d_del);
  }
  StrainTileArgs S = {};
  S.u[0] = f[CALES_U]; S.u[1] = f[CALES_V]; S.u[2] = f[CALES_W]; S.visct = f[CALES_VISCT];
  S.dzci = c->d_dzci; S.dzfi = c->d_dzfi; S.dxi = c->dli[0]; S.dyi = c->dli[1];
  S.zlo = c->is_wall[4] != 0.; S.zhi = c->is_wall[5] != 0.;
  S.wmlo = ISB(c, 0, 3) && LWM(c, 0, 3) != 0; S.wmhi = ISB(c, 1, 3) && LWM(c, 1, 3) != 0;
  S.flo = (1. / c->dzci[0]) * c->dzci[1]; S.fhi = (1. / c->dzci[n[2]]) * c->dzci[n[2] - 1];
  S.zc = c->d_zc; S.del = c->d_del; S.l3 = c->C.l[2]; S.visc = c->visc; S.perx = c->step_xskip ? 1 : 0;
  // walls in y (ducts): is_wall(2:3) is a property of the case, distances use global rows, and the shear of both y walls reaches every slab
  S.wylo = c->is_wall[2] != 0.; S.wyhi = c->is_wall[3] != 0.; S.dl2 = c->dl[1];
  S.wmylo = ISB(c, 0, 2) && LWM(c, 0, 2) != 0; S.wmyhi = ISB(c, 1, 2) && LWM(c, 1, 2) != 0;
  S.twy = nullptr;
  if (S.wylo || S.wyhi) { if (int e = wall_shear_y_planes(c, S.wylo, S.wyhi, &S.twy)) return e; }
  const bool small = (c->ntot + 16) * sizeof(real) < (1ull << 32) && !c->fl.wide_offsets;
  const bool yw = S.wylo || S.wyhi || S.wmylo || S.wmyhi;
  {
    // row-marching form: one wave per row of 62 cells; chunks of k so that every CU holds several blocks' worth of independent waves (the LDS-tile
    // form of this pass -- two barriers per plane, 2.4 against 2.9 TB/s -- lost its A/B in round 2 and went in round 4)
    const dim3 rb(64, SROWS, 1);
    const int rgx = (n[0] + 61) / 62, rgy = (n[1] + SROWS - 1) / SROWS;
    int kr = n[2];
    // sixteen waves per CU at a time: enough blocks for eight rounds or more, or the last round's idle CUs show (chunks pay a three-plane prologue)
    while ((long)rgx * rgy * ((n[2] + kr - 1) / kr) < 8192 && kr > 32) kr = (kr + 1) / 2;
    while ((long)rgx * rgy * ((n[2] + kr - 1) / kr) < 1024 && kr > 8) kr = (kr + 1) / 2;
    if (int fk = tile_kchunk(c, (long)rgx * rgy, n[2])) kr = fk;
    S.kchunk = kr; S.bm = band_map(rgx, rgy, (n[2] + kr - 1) / kr);
    const dim3 rg(band_blocks(S.bm), 1, 1);
    if (yw) { if (small) LAUNCH(c, (k_smag_rows<unsigned, 1>), rg, rb, 0, c->stream, c->g, S); else LAUNCH(c, (k_smag_rows<size_t, 1>), rg, rb, 0, c->stream, c->g, S); }
    else if (small) LAUNCH(c, (k_smag_rows<unsigned, 0>), rg, rb, 0, c->stream, c->g, S); else LAUNCH(c, (k_smag_rows<size_t, 0>), rg, rb, 0, c->stream, c->g, S);
    LAUNCHCHK(c);
    return 0;
  }
}

// every kernel the SGS pass of this case launches reads wrapped interior columns where x is periodic (cales_step may leave the x ghost columns stale)
bool sgs_wraps_x(const cales_ctx *c) {
  if (c->C.sgstype == 0) return true;
  if (c->C.sgstype == 1) return smag_fast_ok(c);
  return dsmag_fast_ok(c) && !c->fl.dsmag_xghosts;
}
// the form of cmpt_sgs this context takes (cales_describe_plan): the same predicates op_cmpt_sgs / dsmag_fast / smag_fast branch on
const char *sgs_path_name(const cales_ctx *c) {
  if (c->C.sgstype == 0) return "none";
  if (c->C.sgstype == 1) return smag_fast_ok(c) ? "smag_rows" : "smag_reference_sequence";
  if (!dsmag_fast_ok(c)) return "dsmag_reference_sequence";
  return dsmag_pairs(c) ? "dsmag_tiles(pair_fields)" : "dsmag_tiles";
}
int op_cmpt_sgs(cales_ctx *c) {
  const int *n = c->n; const size_t nt = c->ntot;
  real **f = c->f; real *visct = f[CALES_VISCT];
  if (c->C.sgstype == 0) {           // 'none': visct = 0 once (sgs.f90:62-68)
    if (c->sgs_first) { c->sgs_first = false; HIPCHK(c, hipMemsetAsync(visct, 0, nt * sizeof(real), c->stream)); c->visct_zero = true; }
    return 0;
  }
  c->visct_lazy = false;      // the field is rewritten from scratch
  ProfScope ps(c, c->C.sgstype == 1 ? "cmpt_sgs_smag" : "cmpt_sgs_dsmag");
  dim3 b(BX, BY, 1), gr = grid3(n[0], n[1], n[2], b);
  if (c->sgs_first) {
    c->sgs_first = false;
    if (c->C.sgstype == 2)
      LAUNCH(c, k_alph2, grid3(n[0] + 2, n[1] + 2, n[2] + 2, dim3(64, 4, 1)), dim3(64, 4, 1), 0, c->stream, c->g, c->is_wall[0], c->is_wall[1],
                         c->is_wall[2], c->is_wall[3], c->is_wall[4], c->is_wall[5], c->alph2);
  }
  if (c->C.sgstype == 2 && dsmag_fast_ok(c)) return dsmag_fast(c);
  if (c->C.sgstype == 1 && smag_fast_ok(c)) return smag_fast(c);
  real **wk = c->wk;
  const int if123[3] = {1, 2, 3};
  if (c->C.sgstype == 1) {
    // wk(1:3) = u,v,w ; extrapolate at wall-model faces ; strain rate (sgs.f90:84-92) -- without the copies: the ghost cells of
    // u,v,w at the wall-model faces are replaced in place, kept in wk(1), and given back before anything else reads them
    real *uvw[3] = {f[CALES_U], f[CALES_V], f[CALES_W]};
    const size_t need = 6 * ((size_t)(n[1] + 2) * (n[2] + 2) + (size_t)(n[0] + 2) * (n[2] + 2) + (size_t)(n[0] + 2) * (n[1] + 2));
    if (need <= c->ntot) {
      if (int e = extrapolate(c, 3, uvw, if123, 0, wk[0], 0)) return e;
      if (int e = strain_rate(c, uvw[0], uvw[1], uvw[2], c->s0, nullptr)) return e;
      if (int e = extrapolate(c, 3, uvw, if123, 0, wk[0], 1)) return e;
    } else {            // degenerate grids: the copies of the reference
      LAUNCH(c, k_copy3, lin_grid(nt), dim3(256), 0, c->stream, nt, f[CALES_U], f[CALES_V], f[CALES_W], wk[0], wk[1], wk[2]);
      if (int e = extrapolate(c, 3, wk, if123, 0)) return e;
      if (int e = strain_rate(c, wk[0], wk[1], wk[2], c->s0, nullptr)) return e;
    }
    SmagArgs A; A.w0 = c->is_wall[0]; A.w1 = c->is_wall[1]; A.w2 = c->is_wall[2]; A.w3 = c->is_wall[3]; A.w4 = c->is_wall[4]; A.w5 = c->is_wall[5];
    A.dl1 = c->dl[0]; A.dl2 = c->dl[1]; A.l3 = c->C.l[2]; A.dxi = c->dli[0]; A.dyi = c->dli[1]; A.visc = c->visc;
    A.sumw = 0.; for (int q = 0; q < 6; ++q) A.sumw += c->is_wall[q];
    if (!c->d_del) {
      HIPCHK(c, hipMalloc(&c->d_del, (n[2] + 2) * sizeof(real)));
      LAUNCH(c, k_smag_del, dim3((n[2] + 2 + 63) / 64), dim3(64), 0, c->stream, n[2] + 2, c->dl[0], c->dl[1], c->d_dzf, c->d_del);
    }
    const real *twy = nullptr;      // several slabs: the shear of the y walls comes from the slabs that own them
    if (c->P > 1 && (c->is_wall[2] != 0. || c->is_wall[3] != 0.)) { if (int e = wall_shear_y_planes(c, c->is_wall[2] != 0., c->is_wall[3] != 0., &twy)) return e; }
    LAUNCH(c, k_smag, gr, b, 0, c->stream, c->g, A, c->d_zc, c->d_dzci, c->d_del, f[CALES_U], f[CALES_V], f[CALES_W], c->s0, visct, twy);
    LAUNCHCHK(c);
    return 0;
  }
  // ---- dynamic model (sgs.f90:153-380): wk(1:3) = u,v,w ; extrapolate at wall-model faces ; strain rate (sgs.f90:173-181)
  LAUNCH(c, k_copy3, lin_grid(nt), dim3(256), 0, c->stream, nt, f[CALES_U], f[CALES_V], f[CALES_W], wk[0], wk[1], wk[2]);
  if (int e = extrapolate(c, 3, wk, if123, 0)) return e;
  real **sij = c->sij, **mij = c->mij, **lij = c->sij;
  if (int e = strain_rate(c, wk[0], wk[1], wk[2], c->s0, sij)) return e;
  LAUNCH(c, k_copy1, lin_grid(nt), dim3(256), 0, c->stream, nt, c->s0, visct);
  if (int e = op_boundp(c, c->s0, 1)) return e;
  for (int m = 0; m < 6; ++m) if (int e = op_boundp(c, sij[m], 1)) return e;
  CP6 csij; P6 pwk, pmij; CP6 cmij;
  for (int m = 0; m < 6; ++m) { csij.p[m] = sij[m]; pwk.p[m] = wk[m]; pmij.p[m] = mij[m]; cmij.p[m] = mij[m]; }
  LAUNCH(c, k_s0sij, lin_grid(nt), dim3(256), 0, c->stream, nt, c->s0, csij, pwk);
  const int if0[6] = {0, 0, 0, 0, 0, 0};
  if (int e = extrapolate(c, 6, wk, if0, 1)) return e;
  for (int m = 0; m < 6; ++m) LAUNCH(c, k_filter3d, gr, b, 0, c->stream, c->g, wk[m], mij[m]);
  LAUNCH(c, k_copy3, lin_grid(nt), dim3(256), 0, c->stream, nt, f[CALES_U], f[CALES_V], f[CALES_W], wk[0], wk[1], wk[2]);
  if (int e = extrapolate(c, 3, wk, if123, 1)) return e;
  LAUNCH(c, k_filter3d, gr, b, 0, c->stream, c->g, wk[0], c->uf);
  LAUNCH(c, k_filter3d, gr, b, 0, c->stream, c->g, wk[1], c->vf);
  LAUNCH(c, k_filter3d, gr, b, 0, c->stream, c->g, wk[2], c->wf);
  if (int e = op_bounduvw(c, c->bcuf, c->bcvf, c->bcwf, 0, 0, c->uf, c->vf, c->wf)) return e;
  real *ff[3] = {c->uf, c->vf, c->wf};
  if (int e = extrapolate(c, 3, ff, if123, 0)) return e;
  if (int e = strain_rate(c, c->uf, c->vf, c->wf, c->s0, sij)) return e;
  LAUNCH(c, k_mij, gr, b, 0, c->stream, c->g, pmij, c->alph2, c->s0, csij);
  LAUNCH(c, k_interp, gr, b, 0, c->stream, c->g, f[CALES_U], f[CALES_V], f[CALES_W], c->uc, c->vc, c->wc);
  if (int e = op_boundp(c, c->uc, 1)) return e;
  if (int e = op_boundp(c, c->vc, 1)) return e;
  if (int e = op_boundp(c, c->wc, 1)) return e;
  LAUNCH(c, k_uiuj, lin_grid(nt), dim3(256), 0, c->stream, nt, c->uc, c->vc, c->wc, pwk);
  if (int e = extrapolate(c, 6, wk, if0, 1)) return e;
  for (int m = 0; m < 6; ++m) LAUNCH(c, k_filter3d, gr, b, 0, c->stream, c->g, wk[m], lij[m]);
  real *cc[3] = {c->uc, c->vc, c->wc};
  if (int e = extrapolate(c, 3, cc, if0, 1)) return e;
  LAUNCH(c, k_filter3d, gr, b, 0, c->stream, c->g, c->uc, c->uf);
  LAUNCH(c, k_filter3d, gr, b, 0, c->stream, c->g, c->vc, c->vf);
  LAUNCH(c, k_filter3d, gr, b, 0, c->stream, c->g, c->wc, c->wf);
  CP6 clij; for (int m = 0; m < 6; ++m) clij.p[m] = lij[m];
  LAUNCH(c, k_contract, gr, b, 0, c->stream, c->g, cmij, clij, c->uf, c->vf, c->wf, wk[0], wk[1]);
  LAUNCH(c, k_plane_sum, dim3(n[2], 2), dim3(256), 0, c->stream, c->g, wk[0], wk[1], c->d_p1d);
  if (c->P > 1) { if (int e = allreduce_res(c, (int)(c->d_p1d - c->res), 2 * n[2], 0)) return e; }   // sgs.f90:475
  const real gar = c->dl[0] * c->dl[1] / (c->C.l[0] * c->C.l[1]);
  LAUNCH(c, k_dsmag_final, gr, b, 0, c->stream, c->g, gar, c->d_p1d, visct, visct);
  LAUNCHCHK(c);
  return 0;
}
