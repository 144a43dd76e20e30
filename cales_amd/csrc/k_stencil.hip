// Stencil / streaming kernels of the RK substep and of the projection:
//   mom_xyz_ad (src/mom.f90:17-309), RK update (src/rk.f90:77-119), bulk forcing (src/mom.f90:311-335,
//   src/utils.f90:16-47), fillps (src/fillps.f90), correc (src/correc.f90), updatep (src/updatep.f90),
//   chkdt (src/chkdt.f90), chkdiv (src/chkdiv.f90).
// Layout: x contiguous, one thread per cell, 64 lanes along x (coalesced 512-B wavefront rows).
// All are HBM-bound; no MFMA on this path.
#include "common.hpp"

#define BX 64
#define BY 4

__constant__ real c_rk[3][2] = {{32. / 60., 0.}, {25. / 60., -17. / 60.}, {45. / 60., -25. / 60.}};

// ------------------------------------------------------------------------------------------ mom_xyz_ad
template <int IMP>
__global__ __launch_bounds__(BX *BY) void k_mom(Geom g, const real *__restrict__ u, const real *__restrict__ v,
                                                 const real *__restrict__ w, const real *__restrict__ s,
                                                 const real *__restrict__ dzci, const real *__restrict__ dzfi, real dxi,
                                                 real dyi, real visc, real *__restrict__ dudt, real *__restrict__ dvdt,
                                                 real *__restrict__ dwdt, real *__restrict__ dudtd,
                                                 real *__restrict__ dvdtd, real *__restrict__ dwdtd) {
  int bx_, by_, bz_; stencil_block(bx_, by_, bz_);
  const int i = bx_ * BX + threadIdx.x + 1, j = by_ * BY + threadIdx.y + 1, k = bz_ + 1;
  if (i > g.n1 || j > g.n2) return;
  const size_t c = g.ix(i, j, k);
  const long sj = g.s1, sk = g.s12;
#define LD(a, di, dj, dk) a[c + (di) + (dj)*sj + (dk)*sk]
  const real u_ccm = LD(u, 0, 0, -1), u_cmc = LD(u, 0, -1, 0),
               u_mcc = LD(u, -1, 0, 0), u_ccc = LD(u, 0, 0, 0), u_pcc = LD(u, 1, 0, 0), u_mpc = LD(u, -1, 1, 0),
               u_cpc = LD(u, 0, 1, 0), u_mcp = LD(u, -1, 0, 1), u_ccp = LD(u, 0, 0, 1);
  const real v_ccm = LD(v, 0, 0, -1), v_cmc = LD(v, 0, -1, 0), v_pmc = LD(v, 1, -1, 0),
               v_mcc = LD(v, -1, 0, 0), v_ccc = LD(v, 0, 0, 0), v_pcc = LD(v, 1, 0, 0), v_cpc = LD(v, 0, 1, 0),
               v_cmp = LD(v, 0, -1, 1), v_ccp = LD(v, 0, 0, 1);
  const real w_ccm = LD(w, 0, 0, -1), w_pcm = LD(w, 1, 0, -1), w_cpm = LD(w, 0, 1, -1), w_cmc = LD(w, 0, -1, 0),
               w_mcc = LD(w, -1, 0, 0), w_ccc = LD(w, 0, 0, 0), w_pcc = LD(w, 1, 0, 0), w_cpc = LD(w, 0, 1, 0),
               w_ccp = LD(w, 0, 0, 1);
  const real s_ccm = LD(s, 0, 0, -1), s_pcm = LD(s, 1, 0, -1), s_cpm = LD(s, 0, 1, -1), s_cmc = LD(s, 0, -1, 0),
               s_pmc = LD(s, 1, -1, 0), s_mcc = LD(s, -1, 0, 0), s_ccc = LD(s, 0, 0, 0), s_pcc = LD(s, 1, 0, 0),
               s_mpc = LD(s, -1, 1, 0), s_cpc = LD(s, 0, 1, 0), s_cmp = LD(s, 0, -1, 1), s_mcp = LD(s, -1, 0, 1),
               s_ccp = LD(s, 0, 0, 1), s_ppc = LD(s, 1, 1, 0), s_pcp = LD(s, 1, 0, 1), s_cpp = LD(s, 0, 1, 1);
#undef LD
  const real dzci_k = dzci[k], dzci_m = dzci[k - 1], dzfi_k = dzfi[k], dzfi_p = dzfi[k + 1];
  real visc_ip, visc_im, visc_jp, visc_jm, visc_kp, visc_km;
  // ---- x momentum (mom.f90:143-186)
  visc_ip = s_pcc; visc_im = s_ccc;
  visc_jp = 0.25 * (s_ccc + s_pcc + s_cpc + s_ppc); visc_jm = 0.25 * (s_ccc + s_pcc + s_cmc + s_pmc);
  visc_kp = 0.25 * (s_ccc + s_pcc + s_ccp + s_pcp); visc_km = 0.25 * (s_ccc + s_pcc + s_ccm + s_pcm);
  const real dudx_ip = (u_pcc - u_ccc) * dxi, dudx_im = (u_ccc - u_mcc) * dxi, dudy_jp = (u_cpc - u_ccc) * dyi,
               dudy_jm = (u_ccc - u_cmc) * dyi, dudz_kp = (u_ccp - u_ccc) * dzci_k, dudz_km = (u_ccc - u_ccm) * dzci_m;
  const real dvdx_jp = (v_pcc - v_ccc) * dxi, dvdx_jm = (v_pmc - v_cmc) * dxi, dwdx_kp = (w_pcc - w_ccc) * dxi,
               dwdx_km = (w_pcm - w_ccm) * dxi;
  const real uu_ip = 0.25 * (u_pcc + u_ccc) * (u_ccc + u_pcc), uu_im = 0.25 * (u_mcc + u_ccc) * (u_ccc + u_mcc),
               vu_jp = 0.25 * (v_pcc + v_ccc) * (u_ccc + u_cpc), vu_jm = 0.25 * (v_pmc + v_cmc) * (u_ccc + u_cmc),
               wu_kp = 0.25 * (w_pcc + w_ccc) * (u_ccc + u_ccp), wu_km = 0.25 * (w_pcm + w_ccm) * (u_ccc + u_ccm);
  const real dudtd_xy = visc * (dudx_ip - dudx_im) * dxi + visc * (dudy_jp - dudy_jm) * dyi;
  const real dudtd_z = visc * (dudz_kp - dudz_km) * dzfi_k;
  real dudt_s = -(uu_ip - uu_im) * dxi - (vu_jp - vu_jm) * dyi - (wu_kp - wu_km) * dzfi_k +
                  (visc_ip * (dudx_ip + dudx_ip) - visc_im * (dudx_im + dudx_im)) * dxi +
                  (visc_jp * (dudy_jp + dvdx_jp) - visc_jm * (dudy_jm + dvdx_jm)) * dyi +
                  (visc_kp * (dudz_kp + dwdx_kp) - visc_km * (dudz_km + dwdx_km)) * dzfi_k;
  // ---- y momentum (mom.f90:188-231)
  visc_ip = 0.25 * (s_ccc + s_cpc + s_pcc + s_ppc); visc_im = 0.25 * (s_ccc + s_cpc + s_mcc + s_mpc);
  visc_jp = s_cpc; visc_jm = s_ccc;
  visc_kp = 0.25 * (s_ccc + s_cpc + s_ccp + s_cpp); visc_km = 0.25 * (s_ccc + s_cpc + s_ccm + s_cpm);
  const real dvdx_ip = (v_pcc - v_ccc) * dxi, dvdx_im = (v_ccc - v_mcc) * dxi, dvdy_jp = (v_cpc - v_ccc) * dyi,
               dvdy_jm = (v_ccc - v_cmc) * dyi, dvdz_kp = (v_ccp - v_ccc) * dzci_k, dvdz_km = (v_ccc - v_ccm) * dzci_m;
  const real dudy_ip = (u_cpc - u_ccc) * dyi, dudy_im = (u_mpc - u_mcc) * dyi, dwdy_kp = (w_cpc - w_ccc) * dyi,
               dwdy_km = (w_cpm - w_ccm) * dyi;
  const real uv_ip = 0.25 * (u_ccc + u_cpc) * (v_ccc + v_pcc), uv_im = 0.25 * (u_mcc + u_mpc) * (v_ccc + v_mcc),
               vv_jp = 0.25 * (v_ccc + v_cpc) * (v_ccc + v_cpc), vv_jm = 0.25 * (v_ccc + v_cmc) * (v_ccc + v_cmc),
               wv_kp = 0.25 * (w_ccc + w_cpc) * (v_ccc + v_ccp), wv_km = 0.25 * (w_ccm + w_cpm) * (v_ccc + v_ccm);
  const real dvdtd_xy = visc * (dvdx_ip - dvdx_im) * dxi + visc * (dvdy_jp - dvdy_jm) * dyi;
  const real dvdtd_z = visc * (dvdz_kp - dvdz_km) * dzfi_k;
  real dvdt_s = -(uv_ip - uv_im) * dxi - (vv_jp - vv_jm) * dyi - (wv_kp - wv_km) * dzfi_k +
                  (visc_ip * (dvdx_ip + dudy_ip) - visc_im * (dvdx_im + dudy_im)) * dxi +
                  (visc_jp * (dvdy_jp + dvdy_jp) - visc_jm * (dvdy_jm + dvdy_jm)) * dyi +
                  (visc_kp * (dvdz_kp + dwdy_kp) - visc_km * (dvdz_km + dwdy_km)) * dzfi_k;
  // ---- z momentum (mom.f90:233-276)
  visc_ip = 0.25 * (s_ccc + s_ccp + s_pcc + s_pcp); visc_im = 0.25 * (s_ccc + s_ccp + s_mcc + s_mcp);
  visc_jp = 0.25 * (s_ccc + s_ccp + s_cpc + s_cpp); visc_jm = 0.25 * (s_ccc + s_ccp + s_cmc + s_cmp);
  visc_kp = s_ccp; visc_km = s_ccc;
  const real dwdx_ip = (w_pcc - w_ccc) * dxi, dwdx_im = (w_ccc - w_mcc) * dxi, dwdy_jp = (w_cpc - w_ccc) * dyi,
               dwdy_jm = (w_ccc - w_cmc) * dyi, dwdz_kp = (w_ccp - w_ccc) * dzfi_p, dwdz_km = (w_ccc - w_ccm) * dzfi_k;
  const real dudz_ip = (u_ccp - u_ccc) * dzci_k, dudz_im = (u_mcp - u_mcc) * dzci_k, dvdz_jp = (v_ccp - v_ccc) * dzci_k,
               dvdz_jm = (v_cmp - v_cmc) * dzci_k;
  const real uw_ip = 0.25 * (u_ccc + u_ccp) * (w_ccc + w_pcc), uw_im = 0.25 * (u_mcc + u_mcp) * (w_ccc + w_mcc),
               vw_jp = 0.25 * (v_ccc + v_ccp) * (w_ccc + w_cpc), vw_jm = 0.25 * (v_cmc + v_cmp) * (w_ccc + w_cmc),
               ww_kp = 0.25 * (w_ccc + w_ccp) * (w_ccc + w_ccp), ww_km = 0.25 * (w_ccc + w_ccm) * (w_ccc + w_ccm);
  const real dwdtd_xy = visc * (dwdx_ip - dwdx_im) * dxi + visc * (dwdy_jp - dwdy_jm) * dyi;
  const real dwdtd_z = visc * (dwdz_kp - dwdz_km) * dzci_k;
  real dwdt_s = -(uw_ip - uw_im) * dxi - (vw_jp - vw_jm) * dyi - (ww_kp - ww_km) * dzci_k +
                  (visc_ip * (dwdx_ip + dudz_ip) - visc_im * (dwdx_im + dudz_im)) * dxi +
                  (visc_jp * (dwdy_jp + dvdz_jp) - visc_jm * (dwdy_jm + dvdz_jm)) * dyi +
                  (visc_kp * (dwdz_kp + dwdz_kp) - visc_km * (dwdz_km + dwdz_km)) * dzci_k;
  if (IMP == 2) {          // _IMPDIFF_1D: mom.f90:278-284
    dudt[c] = dudt_s + dudtd_xy; dvdt[c] = dvdt_s + dvdtd_xy; dwdt[c] = dwdt_s + dwdtd_xy;
    dudtd[c] = dudtd_z; dvdtd[c] = dvdtd_z; dwdtd[c] = dwdtd_z;
  } else if (IMP == 1) {   // _IMPDIFF, all directions implicit: mom.f90:285-288
    dudt[c] = dudt_s; dvdt[c] = dvdt_s; dwdt[c] = dwdt_s;
    dudtd[c] = dudtd_xy + dudtd_z; dvdtd[c] = dvdtd_xy + dvdtd_z; dwdtd[c] = dwdtd_xy + dwdtd_z;
  } else {                 // explicit: mom.f90:297-302
    dudt[c] = dudt_s + dudtd_xy + dudtd_z; dvdt[c] = dvdt_s + dvdtd_xy + dvdtd_z; dwdt[c] = dwdt_s + dwdtd_xy + dwdtd_z;
  }
}

int op_mom(cales_ctx *c) {
  if (int e = materialize_visct(c)) return e;
  ProfScope ps(c, "mom_xyz_ad");
  dim3 b(BX, BY, 1), gr = grid3(c->n[0], c->n[1], c->n[2], b);
  real **f = c->f;
  if (c->C.impdiff == 2)
    LAUNCH(c, k_mom<2>, gr, b, 0, c->stream, c->g, f[CALES_U], f[CALES_V], f[CALES_W], f[CALES_VISCT], c->d_dzci, c->d_dzfi,
                       c->dli[0], c->dli[1], c->visc, f[CALES_DUDT], f[CALES_DVDT], f[CALES_DWDT], f[CALES_DUDTD], f[CALES_DVDTD], f[CALES_DWDTD]);
  else if (c->C.impdiff == 1)
    LAUNCH(c, k_mom<1>, gr, b, 0, c->stream, c->g, f[CALES_U], f[CALES_V], f[CALES_W], f[CALES_VISCT], c->d_dzci, c->d_dzfi,
                       c->dli[0], c->dli[1], c->visc, f[CALES_DUDT], f[CALES_DVDT], f[CALES_DWDT], f[CALES_DUDTD], f[CALES_DVDTD], f[CALES_DWDTD]);
  else
    LAUNCH(c, k_mom<0>, gr, b, 0, c->stream, c->g, f[CALES_U], f[CALES_V], f[CALES_W], f[CALES_VISCT], c->d_dzci, c->d_dzfi,
                       c->dli[0], c->dli[1], c->visc, f[CALES_DUDT], f[CALES_DVDT], f[CALES_DWDT], (real *)nullptr, (real *)nullptr, (real *)nullptr);
  LAUNCHCHK(c);
  return 0;
}

// ------------------------------------------------------------------------------------------ RK update (rk.f90:77-94)
template <int IMP>
__global__ __launch_bounds__(BX *BY) void k_rk_update(Geom g, real f1, real f2, real f12, real dxi, real dyi, real bfx,
                                                       real bfy, real bfz, const real *__restrict__ dzci,
                                                       const real *__restrict__ p, real *__restrict__ u, real *__restrict__ v,
                                                       real *__restrict__ w, const real *__restrict__ du,
                                                       const real *__restrict__ dv, const real *__restrict__ dw,
                                                       const real *__restrict__ duo, const real *__restrict__ dvo,
                                                       const real *__restrict__ dwo, const real *__restrict__ dud,
                                                       const real *__restrict__ dvd, const real *__restrict__ dwd) {
  const int i = blockIdx.x * BX + threadIdx.x + 1, j = blockIdx.y * BY + threadIdx.y + 1, k = blockIdx.z + 1;
  if (i > g.n1 || j > g.n2) return;
  const size_t c = g.ix(i, j, k);
  const real pc = p[c];
  real un = u[c] + f1 * du[c] + f2 * duo[c] + f12 * (bfx - dxi * (p[c + 1] - pc));
  real vn = v[c] + f1 * dv[c] + f2 * dvo[c] + f12 * (bfy - dyi * (p[c + g.s1] - pc));
  real wn = w[c] + f1 * dw[c] + f2 * dwo[c] + f12 * (bfz - dzci[k] * (p[c + g.s12] - pc));
  if (IMP) { un = un + f12 * dud[c]; vn = vn + f12 * dvd[c]; wn = wn + f12 * dwd[c]; }
  u[c] = un; v[c] = vn; w[c] = wn;
}

// Helmholtz r.h.s. (rk.f90:110-119)
__global__ __launch_bounds__(BX *BY) void k_rk_imp_rhs(Geom g, real hf12, real *__restrict__ u, real *__restrict__ v,
                                                        real *__restrict__ w, const real *__restrict__ dud,
                                                        const real *__restrict__ dvd, const real *__restrict__ dwd) {
  const int i = blockIdx.x * BX + threadIdx.x + 1, j = blockIdx.y * BY + threadIdx.y + 1, k = blockIdx.z + 1;
  if (i > g.n1 || j > g.n2) return;
  const size_t c = g.ix(i, j, k);
  u[c] = u[c] - hf12 * dud[c]; v[c] = v[c] - hf12 * dvd[c]; w[c] = w[c] - hf12 * dwd[c];
}

// ------------------------------------------------------------------------------------------ deterministic reductions
// stage 1: one partial per block over the interior; stage 2: one block folds the partials in index order.
__device__ inline real wave_sum(real v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ inline real wave_max(real v) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  return v;
}
template <int OP>   // 0 sum, 1 max
__device__ inline real block_reduce(real v, real *sh) {
  const int tid = threadIdx.y * blockDim.x + threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = (blockDim.x * blockDim.y + 63) >> 6;
  v = OP ? wave_max(v) : wave_sum(v);
  if (lane == 0) sh[wv] = v;
  __syncthreads();
  real r = 0.;
  if (tid == 0) { r = sh[0]; for (int q = 1; q < nw; ++q) r = OP ? fmax(r, sh[q]) : r + sh[q]; }
  __syncthreads();
  return r;
}

// bulk_mean (utils.f90:35-44): sum p*grid_vol_ratio(k) over the interior
// (block = 64 x 4: four rows at a time, lanes along x -- whole lines, no division per cell; 16 x n3 blocks: 15 -> 9 us at 256 x 128 x 128)
__global__ __launch_bounds__(256) void k_bulk_mean_partial(Geom g, const real *__restrict__ p, const real *__restrict__ gvr,
                                                           real *__restrict__ part) {
  __shared__ real sh[4];
  const int k = blockIdx.y + 1;
  real acc = 0.;
  for (int j = blockIdx.x * 4 + threadIdx.y + 1; j <= g.n2; j += gridDim.x * 4) {
    const real *row = p + g.ix(0, j, k);
    for (int i = threadIdx.x + 1; i <= g.n1; i += 64) acc += row[i];
  }
  acc *= gvr[k];
  const real r = block_reduce<0>(acc, sh);
  if (threadIdx.x == 0 && threadIdx.y == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = r;
}
// out[slot] = op(partials)
__global__ __launch_bounds__(256) void k_fold(const real *__restrict__ part, int np, int op, real *__restrict__ out, int slot) {
  __shared__ real sh[4];
  real acc = 0.;
  for (int q = threadIdx.x; q < np; q += 256) acc = op ? fmax(acc, part[q]) : acc + part[q];
  const real r = op ? block_reduce<1>(acc, sh) : block_reduce<0>(acc, sh);
  if (threadIdx.x == 0) out[slot] = r;
}
// fold of the partial sums + f = velf - mean in one launch (one rank: no all-reduce between the two)
__global__ __launch_bounds__(256) void k_fold_force(const real *__restrict__ part, int np, real velf, real *__restrict__ res, int slot, real *__restrict__ force, int comp) {
  __shared__ real sh[4];
  real acc = 0.;
  for (int q = threadIdx.x; q < np; q += 256) acc += part[q];
  const real r = block_reduce<0>(acc, sh);
  if (threadIdx.x == 0) { res[slot] = r; const real f = velf - r; force[comp] = f; force[3 + comp] += f; }
}
// f = velf - mean (rk.f90:209-221), dpdl += f (main.f90:492)
__global__ void k_force_finish(const real *__restrict__ res, int slot, real velf, real *__restrict__ force, int comp) {
  if (threadIdx.x == 0) { const real f = velf - res[slot]; force[comp] = f; force[3 + comp] += f; }
}
// all-reduce of res[slot..slot+count) across the slabs (utils.f90:46, chkdiv.f90:50-51, chkdt.f90:98, sgs.f90:475)
int allreduce_res(cales_ctx *c, int slot, int count, int op) {
  if (c->P == 1) return 0;
  if (!c->comm.on) { c->err = "nranks > 1 but no communication hooks registered (cales_set_comm)"; return 1; }
  const int64_t off = (c->res - c->comm.A) + slot;
  if (c->comm.allred(c->comm.user, off, count, op)) { c->err = "allreduce callback failed"; return 1; }
  return 0;
}

int op_bulk_mean_dev(cales_ctx *c, const real *p, int c_or_f, real *d_out) {
  (void)d_out;
  ProfScope ps(c, "bulk_mean");
  const int nbx = 16;      // (partials: 16 (n3 + 2) of d_red, cales_create)
  dim3 gr(nbx, c->n[2]);
  LAUNCH(c, k_bulk_mean_partial, gr, dim3(64, 4), 0, c->stream, c->g, p, c_or_f ? c->d_gvr_f : c->d_gvr_c, c->d_red + 64);
  LAUNCH(c, k_fold, dim3(1), dim3(256), 0, c->stream, c->d_red + 64, nbx * c->n[2], 0, c->res, 16);
  LAUNCHCHK(c);
  return allreduce_res(c, 16, 1, 0);
}

static int forcing_component(cales_ctx *c, int comp) {   // cmpt_bulk_forcing, rk.f90:197-222
  const int nbx = 16;
  dim3 gr(nbx, c->n[2]);
  const real *p = c->f[CALES_U + comp];
  LAUNCH(c, k_bulk_mean_partial, gr, dim3(64, 4), 0, c->stream, c->g, p, comp == 2 ? c->d_gvr_c : c->d_gvr_f, c->d_red + 64);
  if (c->P == 1) LAUNCH(c, k_fold_force, dim3(1), dim3(256), 0, c->stream, c->d_red + 64, nbx * c->n[2], c->C.velf[comp], c->res, 8 + comp, c->d_force, comp);
  else {
    LAUNCH(c, k_fold, dim3(1), dim3(256), 0, c->stream, c->d_red + 64, nbx * c->n[2], 0, c->res, 8 + comp);
    if (int e = allreduce_res(c, 8 + comp, 1, 0)) return e;
    LAUNCH(c, k_force_finish, dim3(1), dim3(64), 0, c->stream, c->res, 8 + comp, c->C.velf[comp], c->d_force, comp);
  }
  LAUNCHCHK(c);
  return 0;
}

// the same from per-block partial sums made by another pass (k_fft_x8<0,KIND,1>): part[comp*nblk + b]
int op_force_from_partials(cales_ctx *c, int mask, const real *part, int nblk) {
  for (int comp = 0; comp < 3; ++comp) {
    if (!(mask >> comp & 1)) continue;
    if (c->P == 1) { LAUNCH(c, k_fold_force, dim3(1), dim3(256), 0, c->stream, part + (size_t)comp * nblk, nblk, c->C.velf[comp], c->res, 8 + comp, c->d_force, comp); continue; }
    LAUNCH(c, k_fold, dim3(1), dim3(256), 0, c->stream, part + (size_t)comp * nblk, nblk, 0, c->res, 8 + comp);
    if (int e = allreduce_res(c, 8 + comp, 1, 0)) return e;
    LAUNCH(c, k_force_finish, dim3(1), dim3(64), 0, c->stream, c->res, 8 + comp, c->C.velf[comp], c->d_force, comp);
  }
  LAUNCHCHK(c);
  return 0;
}

__global__ void k_zero_force(real *force) { if (threadIdx.x < 3) force[threadIdx.x] = 0.; }

int op_rk(cales_ctx *c, int irk, real dt) {
  static const real rk[3][2] = {{32. / 60., 0.}, {25. / 60., -17. / 60.}, {45. / 60., -25. / 60.}};   // param.f90:27-29
  return op_rk_par(c, rk[irk - 1][0], rk[irk - 1][1], dt);
}
// rk(rkpar, ..., dt, ...) of rk.f90:17 with the caller's coefficients
int op_rk_par(cales_ctx *c, real rkpar1, real rkpar2, real dt) {
  const real f1 = rkpar1 * dt, f2 = rkpar2 * dt, f12 = f1 + f2;
  real **f = c->f;
  dim3 b(BX, BY, 1), gr = grid3(c->n[0], c->n[1], c->n[2], b);
  const bool unfused = c->fl.unfused_rk;
  if (!unfused && c->n[2] >= 2) {
    if (int e = op_momrk(c, f1, f2, f12)) return e;
  } else {
    if (int e = op_mom(c)) return e;
    ProfScope ps(c, "rk_update");
    if (c->C.impdiff)
      LAUNCH(c, k_rk_update<1>, gr, b, 0, c->stream, c->g, f1, f2, f12, c->dli[0], c->dli[1], c->C.bforce[0], c->C.bforce[1], c->C.bforce[2],
                         c->d_dzci, f[CALES_P], f[CALES_U], f[CALES_V], f[CALES_W], f[CALES_DUDT], f[CALES_DVDT], f[CALES_DWDT],
                         f[CALES_DUDTO], f[CALES_DVDTO], f[CALES_DWDTO], f[CALES_DUDTD], f[CALES_DVDTD], f[CALES_DWDTD]);
    else
      LAUNCH(c, k_rk_update<0>, gr, b, 0, c->stream, c->g, f1, f2, f12, c->dli[0], c->dli[1], c->C.bforce[0], c->C.bforce[1], c->C.bforce[2],
                         c->d_dzci, f[CALES_P], f[CALES_U], f[CALES_V], f[CALES_W], f[CALES_DUDT], f[CALES_DVDT], f[CALES_DWDT],
                         f[CALES_DUDTO], f[CALES_DVDTO], f[CALES_DWDTO], (real *)nullptr, (real *)nullptr, (real *)nullptr);
  }
  for (int q = 0; q < 3; ++q) std::swap(f[CALES_DUDT + q], f[CALES_DUDTO + q]);     // swap, rk.f90:98-100
  if (!(c->C.is_forced[0] && c->C.is_forced[1] && c->C.is_forced[2]) && !c->force_zeroed) {      // unforced components stay zero for good
    LAUNCH(c, k_zero_force, dim3(1), dim3(64), 0, c->stream, c->d_force); c->force_zeroed = true; }
  for (int q = 0; q < 3; ++q) if (c->C.is_forced[q] && !(c->fuse_mean_mask >> q & 1)) if (int e = forcing_component(c, q)) return e;
  c->hf12 = .5 * f12;
  if (c->C.impdiff && !c->defer_imp_rhs) {
    ProfScope ps(c, "rk_imp_rhs");
    LAUNCH(c, k_rk_imp_rhs, gr, b, 0, c->stream, c->g, .5 * f12, f[CALES_U], f[CALES_V], f[CALES_W], f[CALES_DUDTD], f[CALES_DVDTD], f[CALES_DWDTD]);
  }
  LAUNCHCHK(c);
  return 0;
}

// bulk_forcing (mom.f90:311-335): u += f, f read from device memory (no host round trip)
__global__ __launch_bounds__(BX *BY) void k_bulk_forcing(Geom g, real *__restrict__ u, real *__restrict__ v, real *__restrict__ w,
                                                          const real *__restrict__ force, int fx, int fy, int fz) {
  const int i = blockIdx.x * BX + threadIdx.x + 1, j = blockIdx.y * BY + threadIdx.y + 1, k = blockIdx.z + 1;
  if (i > g.n1 || j > g.n2) return;
  const size_t c = g.ix(i, j, k);
  if (fx) u[c] += force[0];
  if (fy) v[c] += force[1];
  if (fz) w[c] += force[2];
}
int op_bulk_forcing(cales_ctx *c) {
  if (!(c->C.is_forced[0] || c->C.is_forced[1] || c->C.is_forced[2]) || c->defer_imp_rhs || c->defer_force) return 0;
  ProfScope ps(c, "bulk_forcing");
  dim3 b(BX, BY, 1), gr = grid3(c->n[0], c->n[1], c->n[2], b);
  LAUNCH(c, k_bulk_forcing, gr, b, 0, c->stream, c->g, c->f[CALES_U], c->f[CALES_V], c->f[CALES_W], c->d_force,
                     c->C.is_forced[0], c->C.is_forced[1], c->C.is_forced[2]);
  LAUNCHCHK(c);
  return 0;
}

// ------------------------------------------------------------------------------------------ fillps (fillps.f90:36-47)
__global__ __launch_bounds__(BX *BY) void k_fillps(Geom g, real dti, real dtidxi, real dtidyi, const real *__restrict__ dzfi,
                                                    const real *__restrict__ u, const real *__restrict__ v,
                                                    const real *__restrict__ w, real *__restrict__ p) {
  const int i = blockIdx.x * BX + threadIdx.x + 1, j = blockIdx.y * BY + threadIdx.y + 1, k = blockIdx.z + 1;
  if (i > g.n1 || j > g.n2) return;
  const size_t c = g.ix(i, j, k);
  p[c] = ((w[c] - w[c - g.s12]) * dti * dzfi[k] + (v[c] - v[c - g.s1]) * dtidyi + (u[c] - u[c - 1]) * dtidxi);
}
int op_fillps(cales_ctx *c, real dti) {
  ProfScope ps(c, "fillps");
  dim3 b(BX, BY, 1), gr = grid3(c->n[0], c->n[1], c->n[2], b);
  LAUNCH(c, k_fillps, gr, b, 0, c->stream, c->g, dti, dti * c->dli[0], dti * c->dli[1], c->d_dzfi, c->f[CALES_U], c->f[CALES_V],
                     c->f[CALES_W], c->f[CALES_PP]);
  LAUNCHCHK(c);
  return 0;
}

// ------------------------------------------------------------------------------------------ correc (correc.f90:44-67)
// ranges include ghost planes: u: i=0..n1, j,k=0..n+1; v: j=0..n2; w: k=0..n3
__global__ __launch_bounds__(BX *BY) void k_correc(Geom g, real fi, real fj, real dt, const real *__restrict__ dzci,
                                                    const real *__restrict__ p, real *__restrict__ u, real *__restrict__ v,
                                                    real *__restrict__ w) {
  const int i = blockIdx.x * BX + threadIdx.x, j = blockIdx.y * BY + threadIdx.y, k = blockIdx.z;
  if (i > g.n1 + 1 || j > g.n2 + 1) return;
  const size_t c = g.ix(i, j, k);
  const real pc = p[c];
  if (i <= g.n1) u[c] = u[c] - fi * (p[c + 1] - pc);
  if (j <= g.n2) v[c] = v[c] - fj * (p[c + g.s1] - pc);
  if (k <= g.n3) w[c] = w[c] - dt * dzci[k] * (p[c + g.s12] - pc);
}
// Line-aligned form of the same loops, optionally fused with updatep (updatep.f90:30-47; UPD = 1 explicit, 2 z-implicit):
// a wave covers the 64 cells i = 1 + 64 bx + lane of one row (whole 128-B lines), p(i+1) comes from the next lane, pp is
// read once for both operators. The ghost columns i = 0 and i = n1+1 of the reference's ranges are left to k_correc_edge.
// fmask != 0 (cales_step, see defer_force): the bulk-forcing increment of this substep (mom.f90:311-335, interior cells) is
// added here instead of in a pass of its own -- (u + f) - dt dp/dx, the same two roundings in the same order.
// (A k-marching variant with the pressure planes in registers measured slower: the kernel is a pure stream.)
template <int UPD>
__global__ __launch_bounds__(BX *BY) void k_correc_cell(Geom g, real fi, real fj, real dt, real alpha, const real *__restrict__ dzci,
                                                         const real *__restrict__ dzfi, const real *__restrict__ pp, real *__restrict__ u,
                                                         real *__restrict__ v, real *__restrict__ w, real *__restrict__ p,
                                                         const real *__restrict__ force, int fmask, int perx) {      // perx: pp(n1+1) is read as pp(1)
  const int tx = threadIdx.x, i = blockIdx.x * BX + tx + 1, j = blockIdx.y * BY + threadIdx.y, k = blockIdx.z;
  if (j > g.n2 + 1) return;
  const bool on = i <= g.n1, lastlane = tx == BX - 1 || i == g.n1;
  const size_t c = g.ix(on ? i : g.n1, j, k);
  const real pc = on ? pp[c] : 0.;
  real px = lane_next(pc);
  if (lastlane) px = (perx && i == g.n1) ? pp[g.ix(1, j, k)] : pp[c + 1];
  if (!on) return;
  const bool inner = j >= 1 && j <= g.n2 && k >= 1 && k <= g.n3;
  const real f0 = (fmask & 1) && inner ? force[0] : 0., f1 = (fmask & 2) && inner ? force[1] : 0., f2 = (fmask & 4) && inner ? force[2] : 0.;
  u[c] = (fmask & 1 ? u[c] + f0 : u[c]) - fi * (px - pc);
  if (j <= g.n2) v[c] = (fmask & 2 ? v[c] + f1 : v[c]) - fj * (pp[c + g.s1] - pc);
  const real pn = k <= g.n3 ? pp[c + g.s12] : 0.;
  if (k <= g.n3) w[c] = (fmask & 4 ? w[c] + f2 : w[c]) - dt * dzci[k] * (pn - pc);
  if (UPD && j >= 1 && j <= g.n2 && k >= 1 && k <= g.n3) {
    if (UPD == 1) p[c] = p[c] + pc;
    else p[c] = p[c] + pc + alpha * (((pn - pc) * dzci[k] - (pc - pp[c - g.s12]) * dzci[k - 1]) * dzfi[k]);
  }
}
// ghost columns i = 0 (u,v,w) and i = n1+1 (v,w) of correc.f90:44-67
__global__ __launch_bounds__(256) void k_correc_edge(Geom g, real fi, real fj, real dt, const real *__restrict__ dzci,
                                                      const real *__restrict__ pp, real *__restrict__ u, real *__restrict__ v,
                                                      real *__restrict__ w) {
  const int j = blockIdx.x * 64 + threadIdx.x, k = blockIdx.y * 4 + threadIdx.y, side = blockIdx.z;
  if (j > g.n2 + 1 || k > g.n3 + 1) return;
  const size_t c = g.ix(side ? g.n1 + 1 : 0, j, k);
  const real pc = pp[c];
  if (!side) u[c] = u[c] - fi * (pp[c + 1] - pc);
  if (j <= g.n2) v[c] = v[c] - fj * (pp[c + g.s1] - pc);
  if (k <= g.n3) w[c] = w[c] - dt * dzci[k] * (pp[c + g.s12] - pc);
}
// upd = 0: correc only; 1: correc + updatep in one pass (cales_step)
int op_correc_updatep(cales_ctx *c, real dt, real alpha, int upd) {
  ProfScope ps(c, upd ? "correc_updatep" : "correc");
  const int *n = c->n;
  dim3 b(BX, BY, 1), gr((n[0] + BX - 1) / BX, (n[1] + 2 + BY - 1) / BY, n[2] + 2);
  real *f_[4] = {c->f[CALES_U], c->f[CALES_V], c->f[CALES_W], c->f[CALES_P]};
  const real fi = dt * c->dli[0], fj = dt * c->dli[1];
  const int mode = !upd ? 0 : (c->C.impdiff == 2 ? 2 : 1);
  const int fmask = c->defer_force ? (c->C.is_forced[0] ? 1 : 0) | (c->C.is_forced[1] ? 2 : 0) | (c->C.is_forced[2] ? 4 : 0) : 0;
  const int perx = c->step_xskip ? 1 : 0;      // (operator-level calls read the ghost column of pp the caller provided)
  if (mode == 0) LAUNCH(c, k_correc_cell<0>, gr, b, 0, c->stream, c->g, fi, fj, dt, alpha, c->d_dzci, c->d_dzfi, c->f[CALES_PP], f_[0], f_[1], f_[2], f_[3], c->d_force, fmask, perx);
  else if (mode == 1) LAUNCH(c, k_correc_cell<1>, gr, b, 0, c->stream, c->g, fi, fj, dt, alpha, c->d_dzci, c->d_dzfi, c->f[CALES_PP], f_[0], f_[1], f_[2], f_[3], c->d_force, fmask, perx);
  else LAUNCH(c, k_correc_cell<2>, gr, b, 0, c->stream, c->g, fi, fj, dt, alpha, c->d_dzci, c->d_dzfi, c->f[CALES_PP], f_[0], f_[1], f_[2], f_[3], c->d_force, fmask, perx);
  // periodic x: the ghost columns are overwritten by the periodic copy of the bounduvw that always follows (main.f90:500) -- inside
  // cales_step their correction is dead work
  if (!(c->in_step && c->cbcvel[0] == 'P' && c->cbcvel[1] == 'P'))
    LAUNCH(c, k_correc_edge, dim3((n[1] + 2 + 63) / 64, (n[2] + 2 + 3) / 4, 2), dim3(64, 4, 1), 0, c->stream, c->g, fi, fj, dt, c->d_dzci, c->f[CALES_PP],
                       f_[0], f_[1], f_[2]);
  LAUNCHCHK(c);
  return 0;
}
int op_correc(cales_ctx *c, real dt) {
  if (!c->fl.unfused_correc) return op_correc_updatep(c, dt, 0., 0);
  ProfScope ps(c, "correc");
  dim3 b(BX, BY, 1), gr = grid3(c->n[0] + 2, c->n[1] + 2, c->n[2] + 2, b);
  LAUNCH(c, k_correc, gr, b, 0, c->stream, c->g, dt * c->dli[0], dt * c->dli[1], dt, c->d_dzci, c->f[CALES_PP], c->f[CALES_U],
                     c->f[CALES_V], c->f[CALES_W]);
  LAUNCHCHK(c);
  return 0;
}

// ------------------------------------------------------------------------------------------ updatep (updatep.f90:30-47)
template <int IMP>
__global__ __launch_bounds__(BX *BY) void k_updatep(Geom g, real alpha, real dxi, real dyi, const real *__restrict__ dzci,
                                                     const real *__restrict__ dzfi, const real *__restrict__ pp, real *__restrict__ p) {
  const int i = blockIdx.x * BX + threadIdx.x + 1, j = blockIdx.y * BY + threadIdx.y + 1, k = blockIdx.z + 1;
  if (i > g.n1 || j > g.n2) return;
  const size_t c = g.ix(i, j, k);
  if (IMP == 0) p[c] = p[c] + pp[c];
  else if (IMP == 1)       // updatep.f90:35-42 with the x and y terms
    p[c] = p[c] + pp[c] + alpha * ((pp[c + 1] - 2. * pp[c] + pp[c - 1]) * (dxi * dxi) + (pp[c + g.s1] - 2. * pp[c] + pp[c - g.s1]) * (dyi * dyi) +
                                   ((pp[c + g.s12] - pp[c]) * dzci[k] - (pp[c] - pp[c - g.s12]) * dzci[k - 1]) * dzfi[k]);
  else p[c] = p[c] + pp[c] + alpha * (((pp[c + g.s12] - pp[c]) * dzci[k] - (pp[c] - pp[c - g.s12]) * dzci[k - 1]) * dzfi[k]);
}
int op_updatep(cales_ctx *c, real alpha) {
  ProfScope ps(c, "updatep");
  dim3 b(BX, BY, 1), gr = grid3(c->n[0], c->n[1], c->n[2], b);
  if (c->C.impdiff == 2) LAUNCH(c, k_updatep<2>, gr, b, 0, c->stream, c->g, alpha, c->dli[0], c->dli[1], c->d_dzci, c->d_dzfi, c->f[CALES_PP], c->f[CALES_P]);
  else if (c->C.impdiff == 1) LAUNCH(c, k_updatep<1>, gr, b, 0, c->stream, c->g, alpha, c->dli[0], c->dli[1], c->d_dzci, c->d_dzfi, c->f[CALES_PP], c->f[CALES_P]);
  else LAUNCH(c, k_updatep<0>, gr, b, 0, c->stream, c->g, alpha, c->dli[0], c->dli[1], c->d_dzci, c->d_dzfi, c->f[CALES_PP], c->f[CALES_P]);
  LAUNCHCHK(c);
  return 0;
}

// ------------------------------------------------------------------------------------------ chkdiv (chkdiv.f90:35-47)
__global__ __launch_bounds__(256) void k_chkdiv_partial(Geom g, real dxi, real dyi, const real *__restrict__ dzfi,
                                                        const real *__restrict__ u, const real *__restrict__ v,
                                                        const real *__restrict__ w, real *__restrict__ psum, real *__restrict__ pmax) {
  __shared__ real sh[4];
  const int k = blockIdx.y + 1;
  real acc = 0., mx = 0.;
  for (int j = blockIdx.x * 4 + threadIdx.y + 1; j <= g.n2; j += gridDim.x * 4)
  for (int i = threadIdx.x + 1; i <= g.n1; i += 64) {
    const size_t c = g.ix(i, j, k);
    const real div = (w[c] - w[c - g.s12]) * dzfi[k] + (v[c] - v[c - g.s1]) * dyi + (u[c] - u[c - 1]) * dxi;
    mx = fmax(mx, fabs(div)); acc += div;
  }
  const real rs = block_reduce<0>(acc, sh), rm = block_reduce<1>(mx, sh);
  if (threadIdx.x == 0 && threadIdx.y == 0) { const size_t o = (size_t)blockIdx.y * gridDim.x + blockIdx.x; psum[o] = rs; pmax[o] = rm; }
}
int op_chkdiv(cales_ctx *c, real *divtot, real *divmax) {
  const int nbx = 8, np = nbx * c->n[2];
  LAUNCH(c, k_chkdiv_partial, dim3(nbx, c->n[2]), dim3(64, 4), 0, c->stream, c->g, c->dli[0], c->dli[1], c->d_dzfi, c->f[CALES_U],
                     c->f[CALES_V], c->f[CALES_W], c->d_red + 64, c->d_red + 64 + np);
  LAUNCH(c, k_fold, dim3(1), dim3(256), 0, c->stream, c->d_red + 64, np, 0, c->res, 0);
  LAUNCH(c, k_fold, dim3(1), dim3(256), 0, c->stream, c->d_red + 64 + np, np, 1, c->res, 1);
  if (int e = allreduce_res(c, 0, 1, 0)) return e;
  if (int e = allreduce_res(c, 1, 1, 1)) return e;
  HIPCHK(c, hipMemcpyAsync(c->h_red, c->res, 2 * sizeof(real), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *divtot = c->h_red[0]; *divmax = c->h_red[1];
  return 0;
}

// ------------------------------------------------------------------------------------------ plane statistics (output.f90:509-700)
// First block of out1d_single_point_chan (idir = 3): 27 sums per z plane -- velocity moments up to the fourth, <uw> at the cell
// edge, pressure, vorticity and its squares, the modelled stresses, <visct>, <du/dz> -- times dx dy/(lx ly). One block row per
// plane, partial sums in a fixed order (deterministic); the divisions by the spacings are kept as the reference writes them.
#define NSTAT 27
__global__ __launch_bounds__(256) void k_stats_chan_partial(Geom g, real dx, real dy, const real *__restrict__ dzc, const real *__restrict__ dzf,
                                                            const real *__restrict__ u, const real *__restrict__ v, const real *__restrict__ w,
                                                            const real *__restrict__ p, const real *__restrict__ s, real *__restrict__ part) {
  __shared__ real sh[4];
  const int k = blockIdx.y + 1;
  real b[NSTAT];
#pragma unroll
  for (int q = 0; q < NSTAT; ++q) b[q] = 0.;
  const long nplane = (long)g.n1 * g.n2, sj = g.s1, sk = g.s12;
  const real zc = dzc[k], zfp = dzf[k + 1], zf = dzf[k];
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nplane; q += (long)gridDim.x * 256) {
    const int i = (int)(q % g.n1) + 1, j = (int)(q / g.n1) + 1;
    const size_t c = g.ix(i, j, k);
    const real uc = u[c], vc = v[c], wc = w[c], pc = p[c];
    const real u_kp = u[c + sk], u_ip = u[c + 1], u_im = u[c - 1], u_jp = u[c + sj];
    const real v_kp = v[c + sk], v_ip = v[c + 1], v_jp = v[c + sj], v_jm = v[c - sj];
    const real w_ip = w[c + 1], w_jp = w[c + sj], w_kp = w[c + sk], w_km = w[c - sk];
    b[0] += uc; b[1] += vc; b[2] += wc;
    b[3] += uc * uc; b[4] += vc * vc; b[5] += wc * wc;
    b[6] += 0.25 * (u_kp + uc) * (wc + w_ip);
    b[7] += uc * uc * uc; b[8] += vc * vc * vc; b[9] += wc * wc * wc;
    b[10] += (uc * uc) * (uc * uc); b[11] += (vc * vc) * (vc * vc); b[12] += (wc * wc) * (wc * wc);
    b[13] += pc; b[14] += pc * pc;
    const real ox = (w_jp - wc) / dy - (v_kp - vc) / zc, oy = (u_kp - uc) / zc - (w_ip - wc) / dx, oz = (v_ip - vc) / dx - (u_jp - uc) / dy;
    b[15] += ox; b[16] += oy; b[17] += oz; b[18] += ox * ox; b[19] += oy * oy; b[20] += oz * oz;
    const real s_ccc = s[c], s_pcc = s[c + 1], s_cpc = s[c + sj], s_ccp = s[c + sk], s_pcp = s[c + 1 + sk];
    const real dudx_ip = (u_ip - uc) / dx, dudx_im = (uc - u_im) / dx, dvdy_jp = (v_jp - vc) / dy, dvdy_jm = (vc - v_jm) / dy;
    const real dwdz_kp = (w_kp - wc) / zfp, dwdz_km = (wc - w_km) / zf, dudz = (u_kp - uc) / zc, dwdx = (w_ip - wc) / dx;
    b[21] -= 0.5 * (s_pcc * (dudx_ip + dudx_ip) + s_ccc * (dudx_im + dudx_im));
    b[22] -= 0.5 * (s_cpc * (dvdy_jp + dvdy_jp) + s_ccc * (dvdy_jm + dvdy_jm));
    b[23] -= 0.5 * (s_ccp * (dwdz_kp + dwdz_kp) + s_ccc * (dwdz_km + dwdz_km));
    b[24] -= 0.25 * (s_ccc + s_pcc + s_ccp + s_pcp) * (dudz + dwdx);
    b[25] += s_ccc;
    b[26] += dudz;
  }
#pragma unroll
  for (int q = 0; q < NSTAT; ++q) {
    const real r = block_reduce<0>(b[q], sh);
    if (threadIdx.x == 0) part[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * NSTAT + q] = r;
  }
}
__global__ void k_stats_fold(int n3, int nbx, real ratio, const real *__restrict__ part, real *__restrict__ out) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= NSTAT * n3) return;
  const int q = t % NSTAT, k = t / NSTAT;
  real a = 0.;
  for (int bx = 0; bx < nbx; ++bx) a += part[((size_t)k * nbx + bx) * NSTAT + q];
  out[t] = a * ratio;
}
// buf: (27, n3) column-major on the host; with several ranks the sums of THIS rank's rows (the caller adds the ranks, output.f90:691)
int op_stats_chan(cales_ctx *c, real *buf) {
  if (int e = materialize_visct(c)) return e;
  const int nbx = 8, n3 = c->n[2];
  const size_t need = (size_t)NSTAT * n3 * (nbx + 1);
  if (!c->d_stat) HIPCHK(c, hipMalloc(&c->d_stat, need * sizeof(real)));
  real *part = c->d_stat, *out = c->d_stat + (size_t)NSTAT * n3 * nbx;
  LAUNCH(c, k_stats_chan_partial, dim3(nbx, n3), dim3(256), 0, c->stream, c->g, c->dl[0], c->dl[1], c->d_dzc, c->d_dzf, c->f[CALES_U], c->f[CALES_V],
                     c->f[CALES_W], c->f[CALES_P], c->f[CALES_VISCT], part);
  const real ratio = c->dl[0] * c->dl[1] / (c->C.l[0] * c->C.l[1]);
  LAUNCH(c, k_stats_fold, dim3((NSTAT * n3 + 255) / 256), dim3(256), 0, c->stream, n3, nbx, ratio, part, out);
  LAUNCHCHK(c);
  HIPCHK(c, hipMemcpyAsync(buf, out, (size_t)NSTAT * n3 * sizeof(real), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

// Second block of out1d_single_point_chan (output.f90:700-1001): the 38 plane sums of the mean-kinetic-energy and Reynolds-stress
// budgets (transport, pressure-strain, dissipation pieces at cell centres and cell edges), same launch shape as the first block.
#define NBUDGET 38
__global__ __launch_bounds__(256) void k_stats_budget_partial(Geom g, real dx, real dy, const real *__restrict__ dzc, const real *__restrict__ dzf,
                                                              const real *__restrict__ u, const real *__restrict__ v, const real *__restrict__ w,
                                                              const real *__restrict__ p, real *__restrict__ part) {
  __shared__ real sh[4];
  const int k = blockIdx.y + 1;
  real b[NBUDGET];
#pragma unroll
  for (int q = 0; q < NBUDGET; ++q) b[q] = 0.;
  const long nplane = (long)g.n1 * g.n2, sj = g.s1, sk = g.s12;
  const real zc = dzc[k], zcm = dzc[k - 1], zf = dzf[k], zfp = dzf[k + 1];
  auto sq = [](real x) { return x * x; };
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nplane; q += (long)gridDim.x * 256) {
    const int i = (int)(q % g.n1) + 1, j = (int)(q / g.n1) + 1;
    const size_t c = g.ix(i, j, k);
    const real uc = u[c], u_kp = u[c + sk], u_km = u[c - sk], u_im = u[c - 1], u_jp = u[c + sj], u_jm = u[c - sj];
    const real u_im_kp = u[c - 1 + sk], u_im_km = u[c - 1 - sk], u_im_jp = u[c - 1 + sj], u_im_jm = u[c - 1 - sj];
    const real vc = v[c], v_kp = v[c + sk], v_km = v[c - sk], v_ip = v[c + 1], v_im = v[c - 1], v_jm = v[c - sj];
    const real v_ip_jm = v[c + 1 - sj], v_im_jm = v[c - 1 - sj], v_jm_kp = v[c - sj + sk], v_jm_km = v[c - sj - sk];
    const real wc = w[c], w_kp = w[c + sk], w_km = w[c - sk], w_ip = w[c + 1], w_im = w[c - 1], w_jp = w[c + sj], w_jm = w[c - sj];
    const real w_ip_km = w[c + 1 - sk], w_im_km = w[c - 1 - sk], w_jp_km = w[c + sj - sk], w_jm_km = w[c - sj - sk], w_ip_kp = w[c + 1 + sk];
    const real pc = p[c], p_kp = p[c + sk];
    const real dudz4 = 0.25 * ((u_kp - uc) / zc + (uc - u_km) / zcm + (u_im_kp - u_im) / zc + (u_im - u_im_km) / zcm);
    const real dwdx4 = 0.25 * ((w_ip - wc) / dx + (wc - w_im) / dx + (w_ip_km - w_km) / dx + (w_km - w_im_km) / dx);
    const real dudy4 = 0.25 * ((u_jp - uc) / dy + (uc - u_jm) / dy + (u_im_jp - u_im) / dy + (u_im - u_im_jm) / dy);
    const real dwdy4 = 0.25 * ((w_jp - wc) / dy + (wc - w_jm) / dy + (w_jp_km - w_km) / dy + (w_km - w_jm_km) / dy);
    b[0] += uc;
    b[1] += 0.5 * (uc + u_kp);
    b[2] += (u_kp - uc) / zc;
    b[3] += (u_kp * u_kp - uc * uc) / zc;
    b[4] += 0.25 * (u_kp + uc) * (wc + w_ip);
    b[5] += 0.25 * (u_im + uc) * (wc + w_km);
    b[6] += dudz4;
    b[7] += 0.125 * sq(u_kp + uc) * (wc + w_ip);
    b[8] += pc;
    b[9] += (uc - u_im) / dx * pc;
    b[10] += sq((uc - u_im) / dx) + 0.25 * (sq((u_jp - uc) / dy) + sq((uc - u_jm) / dy) + sq((u_im_jp - u_im) / dy) + sq((u_im - u_im_jm) / dy)) +
             0.25 * (sq((u_kp - uc) / zc) + sq((uc - u_km) / zcm) + sq((u_im_kp - u_im) / zc) + sq((u_im - u_im_km) / zcm));
    b[11] += (v_kp * v_kp - vc * vc) / zc;
    b[12] += 0.125 * sq(v_kp + vc) * (wc + w_jp);
    b[13] += (vc - v_jm) / dy * pc;
    b[14] += 0.25 * (sq((v_ip - vc) / dx) + sq((vc - v_im) / dx) + sq((v_ip_jm - v_jm) / dx) + sq((v_jm - v_im_jm) / dx)) + sq((vc - v_jm) / dy) +
             0.25 * (sq((v_kp - vc) / zc) + sq((vc - v_km) / zcm) + sq((v_jm_kp - v_jm) / zc) + sq((v_jm - v_jm_km) / zcm));
    b[15] += 0.5 * ((w_kp * w_kp - wc * wc) / zfp + (wc * wc - w_km * w_km) / zf);
    b[16] += wc * wc * wc;
    b[17] += wc * 0.5 * (p_kp + pc);
    b[18] += (wc - w_km) / zf * pc;
    b[19] += 0.25 * (sq((w_ip - wc) / dx) + sq((wc - w_im) / dx) + sq((w_ip_km - w_km) / dx) + sq((w_km - w_im_km) / dx)) +
             0.25 * (sq((w_jp - wc) / dy) + sq((wc - w_jm) / dy) + sq((w_jp_km - w_km) / dy) + sq((w_km - w_jm_km) / dy)) + sq((wc - w_km) / zf);
    b[20] += 0.5 * (wc * wc + w_km * w_km);
    b[21] += (0.25 * (wc + w_kp + w_ip_kp + w_ip) * u_kp - 0.25 * (wc + w_km + w_ip_km + w_ip) * uc) / zc;
    b[22] += wc * wc;
    b[23] += 0.125 * (u_kp + uc) * sq(wc + w_ip);
    b[24] += 0.5 * (p_kp + pc);
    b[25] += 0.25 * (uc + u_kp + u_im_kp + u_im) * 0.5 * (p_kp + pc);
    b[26] += dudz4 * pc + dwdx4 * pc;
    b[27] += (uc - u_im) / dx * dwdx4 + dudy4 * dwdy4 + dudz4 * ((wc - w_km) / zf);
    b[28] += (u_kp - uc) / zc;
    b[29] += sq((uc - u_im) / dx); b[30] += sq((u_jp - uc) / dy); b[31] += sq((u_kp - uc) / zc);
    b[32] += sq((v_ip - vc) / dx); b[33] += sq((vc - v_jm) / dy); b[34] += sq((v_kp - vc) / zc);
    b[35] += sq((w_ip - wc) / dx); b[36] += sq((w_jp - wc) / dy); b[37] += sq((wc - w_km) / zf);
  }
#pragma unroll
  for (int q = 0; q < NBUDGET; ++q) {
    const real r = block_reduce<0>(b[q], sh);
    if (threadIdx.x == 0) part[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * NBUDGET + q] = r;
  }
}
// Third block (output.f90:1005-1041): six divergence measures per plane -- max |div|, sum |div|, sum div, and the same weighted by dzf(k)
__global__ __launch_bounds__(256) void k_stats_leak_partial(Geom g, real dx, real dy, const real *__restrict__ dzf, const real *__restrict__ u,
                                                            const real *__restrict__ v, const real *__restrict__ w, real *__restrict__ part) {
  __shared__ real sh[4];
  const int k = blockIdx.y + 1;
  real mx = 0., sa = 0., sd = 0.;
  const long nplane = (long)g.n1 * g.n2;
  const real zf = dzf[k];
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nplane; q += (long)gridDim.x * 256) {
    const int i = (int)(q % g.n1) + 1, j = (int)(q / g.n1) + 1;
    const size_t c = g.ix(i, j, k);
    const real div = (w[c] - w[c - g.s12]) / zf + (v[c] - v[c - g.s1]) / dy + (u[c] - u[c - 1]) / dx;
    mx = fmax(mx, fabs(div)); sa += fabs(div); sd += div;
  }
  const real rm = block_reduce<1>(mx, sh), ra = block_reduce<0>(sa, sh), rd = block_reduce<0>(sd, sh);
  if (threadIdx.x == 0) { real *o = part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3; o[0] = rm; o[1] = ra; o[2] = rd; }
}
__global__ void k_stats_fold_n(int nstat, int n3, int nbx, real ratio, const real *__restrict__ part, real *__restrict__ out) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nstat * n3) return;
  const int q = t % nstat, k = t / nstat;
  real a = 0.;
  for (int bx = 0; bx < nbx; ++bx) a += part[((size_t)k * nbx + bx) * nstat + q];
  out[t] = a * ratio;
}
__global__ void k_stats_leak_fold(int n3, int nbx, real ratio, const real *__restrict__ dzf, const real *__restrict__ part, real *__restrict__ out) {
  const int k = blockIdx.x * 64 + threadIdx.x;      // plane k+1
  if (k >= n3) return;
  real mx = 0., sa = 0., sd = 0.;
  for (int bx = 0; bx < nbx; ++bx) { const real *o = part + ((size_t)k * nbx + bx) * 3; mx = fmax(mx, o[0]); sa += o[1]; sd += o[2]; }
  const real zf = dzf[k + 1];
  real *r = out + 6 * (size_t)k;
  r[0] = mx; r[1] = sa * ratio; r[2] = sd * ratio; r[3] = mx * zf; r[4] = sa * zf * ratio; r[5] = sd * zf * ratio;
}
// budget: (38, n3), leak: (6, n3), column-major on the host (either may be NULL); this rank's rows when there are several ranks
int op_stats_chan_budget(cales_ctx *c, real *budget, real *leak) {
  const int nbx = 8, n3 = c->n[2];
  const size_t need = (size_t)NBUDGET * n3 * (nbx + 1);
  if (!c->d_stat2) HIPCHK(c, hipMalloc(&c->d_stat2, need * sizeof(real)));
  real *part = c->d_stat2, *out = c->d_stat2 + (size_t)NBUDGET * n3 * nbx;
  const real ratio = c->dl[0] * c->dl[1] / (c->C.l[0] * c->C.l[1]);
  if (budget) {
    LAUNCH(c, k_stats_budget_partial, dim3(nbx, n3), dim3(256), 0, c->stream, c->g, c->dl[0], c->dl[1], c->d_dzc, c->d_dzf, c->f[CALES_U], c->f[CALES_V],
                       c->f[CALES_W], c->f[CALES_P], part);
    LAUNCH(c, k_stats_fold_n, dim3((NBUDGET * n3 + 255) / 256), dim3(256), 0, c->stream, NBUDGET, n3, nbx, ratio, part, out);
    HIPCHK(c, hipMemcpyAsync(budget, out, (size_t)NBUDGET * n3 * sizeof(real), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  if (leak) {
    LAUNCH(c, k_stats_leak_partial, dim3(nbx, n3), dim3(256), 0, c->stream, c->g, c->dl[0], c->dl[1], c->d_dzf, c->f[CALES_U], c->f[CALES_V], c->f[CALES_W], part);
    LAUNCH(c, k_stats_leak_fold, dim3((n3 + 63) / 64), dim3(64), 0, c->stream, n3, nbx, ratio, c->d_dzf, part, out);
    HIPCHK(c, hipMemcpyAsync(leak, out, (size_t)6 * n3 * sizeof(real), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  LAUNCHCHK(c);
  return 0;
}

// ------------------------------------------------------------------------------------------ profiles and duct statistics (output.f90:50-163, 317-507)
// out1d (output.f90:50-163): the profile of a scalar along direction idir, averaged over the other two (weighted with dz(k) when z is one of them);
// one block per entry of the profile, fixed summation order (threads stride the plane, wave and block sums in a fixed tree)
__global__ __launch_bounds__(256) void k_out1d(Geom g, int idir, real ratio, const real *__restrict__ dz, const real *__restrict__ p, real *__restrict__ out) {
  __shared__ real sh[4];
  const int e = blockIdx.x + 1;
  real acc = 0.;
  if (idir == 3) {
    const long np = (long)g.n1 * g.n2;
    for (long q = threadIdx.x; q < np; q += 256) acc += p[g.ix((int)(q % g.n1) + 1, (int)(q / g.n1) + 1, e)];
  } else if (idir == 2) {
    const long np = (long)g.n1 * g.n3;
    for (long q = threadIdx.x; q < np; q += 256) { const int k = (int)(q / g.n1) + 1; acc += p[g.ix((int)(q % g.n1) + 1, e, k)] * dz[k]; }
  } else {
    const long np = (long)g.n2 * g.n3;
    for (long q = threadIdx.x; q < np; q += 256) { const int k = (int)(q / g.n2) + 1; acc += p[g.ix(e, (int)(q % g.n2) + 1, k)] * dz[k]; }
  }
  const real r = block_reduce<0>(acc, sh);
  if (threadIdx.x == 0) out[e - 1] = r * ratio;
}
int op_out1d(cales_ctx *c, int field, int idir, int use_dzc, real *buf) {
  if (field < 0 || field >= CALES_NFIELDS || idir < 1 || idir > 3) { c->err = "cales_out1d: bad field or direction"; return 1; }
  if (field == CALES_VISCT) if (int e = materialize_visct(c)) return e;
  const int ne = c->n[idir - 1];
  real *out = nullptr; HIPCHK(c, hipMalloc(&out, (size_t)ne * sizeof(real)));
  // grid_area_ratio of the reference: dl(1) dl(2) / (l(1) l(2)) along z, dl(1) / (l(1) l(3)) along y, dl(2) / (l(2) l(3)) along x
  const real ratio = idir == 3 ? c->dl[0] * c->dl[1] / (c->C.l[0] * c->C.l[1]) : idir == 2 ? c->dl[0] / (c->C.l[0] * c->C.l[2]) : c->dl[1] / (c->C.l[1] * c->C.l[2]);
  LAUNCH(c, k_out1d, dim3(ne), dim3(256), 0, c->stream, c->g, idir, ratio, use_dzc ? c->d_dzc : c->d_dzf, c->f[field], out);
  LAUNCHCHK(c);
  HIPCHK(c, hipMemcpyAsync(buf, out, (size_t)ne * sizeof(real), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  hipFree(out);
  return 0;
}
// out1d_chan (output.f90:317-405, idir = 3): um, vm, wm, u2, v2, w2, uw per plane
__global__ __launch_bounds__(256) void k_out1d_chan(Geom g, real ratio, const real *__restrict__ u, const real *__restrict__ v, const real *__restrict__ w, real *__restrict__ out) {
  __shared__ real sh[4];
  const int k = blockIdx.x + 1;
  real b[7] = {0., 0., 0., 0., 0., 0., 0.};
  const long np = (long)g.n1 * g.n2, sk = g.s12;
  for (long q = threadIdx.x; q < np; q += 256) {
    const size_t c = g.ix((int)(q % g.n1) + 1, (int)(q / g.n1) + 1, k);
    const real uc = u[c], vc = v[c], wc = w[c], wm = w[c - sk];
    b[0] += uc; b[1] += vc; b[2] += 0.50 * (wm + wc);
    b[3] += uc * uc; b[4] += vc * vc; b[5] += 0.50 * (wc * wc + wm * wm);
    b[6] += 0.25 * (u[c - 1] + uc) * (wm + wc);
  }
#pragma unroll
  for (int q = 0; q < 7; ++q) { const real r = block_reduce<0>(b[q], sh); if (threadIdx.x == 0) out[q + 7 * (size_t)(k - 1)] = r * ratio; }
}
int op_out1d_chan(cales_ctx *c, real *buf) {
  const int n3 = c->n[2];
  real *out = nullptr; HIPCHK(c, hipMalloc(&out, (size_t)7 * n3 * sizeof(real)));
  LAUNCH(c, k_out1d_chan, dim3(n3), dim3(256), 0, c->stream, c->g, c->dl[0] * c->dl[1] / (c->C.l[0] * c->C.l[1]), c->f[CALES_U], c->f[CALES_V], c->f[CALES_W], out);
  LAUNCHCHK(c);
  HIPCHK(c, hipMemcpyAsync(buf, out, (size_t)7 * n3 * sizeof(real), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  hipFree(out);
  return 0;
}
// out2d_duct (output.f90:406-507, streamwise direction x): nine cell-centred averages along x for every (j, k); one wave per (j, k)
__global__ __launch_bounds__(256) void k_out2d_duct(Geom g, real ratio, const real *__restrict__ u, const real *__restrict__ v, const real *__restrict__ w, real *__restrict__ out) {
  const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6) + 1, k = blockIdx.y + 1;
  if (j > g.n2) return;
  real b[9] = {0., 0., 0., 0., 0., 0., 0., 0., 0.};
  const long sj = g.s1, sk = g.s12;
  for (int i = lane + 1; i <= g.n1; i += 64) {
    const size_t c = g.ix(i, j, k);
    const real uc = u[c], um = u[c - 1], vc = v[c], vm = v[c - sj], wc = w[c], wm = w[c - sk];
    b[0] += uc; b[1] += 0.5 * (vm + vc); b[2] += 0.5 * (wm + wc);
    b[3] += uc * uc; b[4] += 0.5 * (vm * vm + vc * vc); b[5] += 0.5 * (wm * wm + wc * wc);
    b[6] += 0.25 * (um + uc) * (vm + vc); b[7] += 0.25 * (um + uc) * (wm + wc); b[8] += 0.25 * (vm + vc) * (wm + wc);
  }
#pragma unroll
  for (int q = 0; q < 9; ++q) { const real r = wave_sum(b[q]); if (lane == 0) out[q + 9 * ((size_t)(j - 1) + (size_t)g.n2 * (k - 1))] = r * ratio; }
}
int op_out2d_duct(cales_ctx *c, real *buf) {
  const int n2 = c->n[1], n3 = c->n[2];
  real *out = nullptr; HIPCHK(c, hipMalloc(&out, (size_t)9 * n2 * n3 * sizeof(real)));
  LAUNCH(c, k_out2d_duct, dim3((n2 + 3) / 4, n3), dim3(256), 0, c->stream, c->g, c->dl[0] / c->C.l[0], c->f[CALES_U], c->f[CALES_V], c->f[CALES_W], out);
  LAUNCHCHK(c);
  HIPCHK(c, hipMemcpyAsync(buf, out, (size_t)9 * n2 * n3 * sizeof(real), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  hipFree(out);
  return 0;
}

// ------------------------------------------------------------------------------------------ chkdt (chkdt.f90:50-98)
template <int IMP>
__global__ __launch_bounds__(256) void k_chkdt_partial(Geom g, real dxi, real dyi, real visc, const real *__restrict__ dzci,
                                                       const real *__restrict__ dzfi, const real *__restrict__ s,
                                                       const real *__restrict__ u, const real *__restrict__ v,
                                                       const real *__restrict__ w, real *__restrict__ pa, real *__restrict__ pd) {
  __shared__ real sh[4];
  const int k = blockIdx.y + 1;
  const real dl2i = dxi * dxi + dyi * dyi, zf2 = dzfi[k] * dzfi[k], zc2 = dzci[k] * dzci[k];
  real dti = 0., dtid = 0.;
  const long sj = g.s1, sk = g.s12;
  // (block = 64 x 4: lanes along x, four rows at a time -- no division per cell)
  for (int j = blockIdx.x * 4 + threadIdx.y + 1; j <= g.n2; j += gridDim.x * 4)
  for (int i = threadIdx.x + 1; i <= g.n1; i += 64) {
    const size_t c = g.ix(i, j, k);
    const real ux = fabs(u[c]), vx = 0.25 * fabs(v[c] + v[c - sj] + v[c + 1] + v[c + 1 - sj]),
                 wx = 0.25 * fabs(w[c] + w[c - sk] + w[c + 1] + w[c + 1 - sk]);
    const real uy = 0.25 * fabs(u[c] + u[c + sj] + u[c - 1 + sj] + u[c - 1]), vy = fabs(v[c]),
                 wy = 0.25 * fabs(w[c] + w[c + sj] + w[c + sj - sk] + w[c - sk]);
    const real uz = 0.25 * fabs(u[c] + u[c - 1] + u[c - 1 + sk] + u[c + sk]), vz = 0.25 * fabs(v[c] + v[c - sj] + v[c - sj + sk] + v[c + sk]),
                 wz = fabs(w[c]);
    const real dtix = ux * dxi + vx * dyi + wx * dzfi[k], dtiy = uy * dxi + vy * dyi + wy * dzfi[k], dtiz = uz * dxi + vz * dyi + wz * dzci[k];
    dti = fmax(fmax(fmax(dti, dtix), dtiy), dtiz);
    const real viscx = 0.5 * (s[c] + s[c + 1]), viscy = 0.5 * (s[c] + s[c + sj]), viscz = 0.5 * (s[c] + s[c + sk]);
    real dtidx = viscx * (dl2i + zf2), dtidy = viscy * (dl2i + zf2), dtidz = viscz * (dl2i + zc2);
    if (IMP != 1) { dtidx += visc * dl2i; dtidy += visc * dl2i; dtidz += visc * dl2i; }
    if (IMP == 0) { dtidx += visc * zf2; dtidy += visc * zf2; dtidz += visc * zc2; }
    dtid = fmax(fmax(fmax(dtid, dtidx), dtidy), dtidz);
  }
  const real ra = block_reduce<1>(dti, sh), rd = block_reduce<1>(dtid, sh);
  if (threadIdx.x == 0 && threadIdx.y == 0) { const size_t o = (size_t)blockIdx.y * gridDim.x + blockIdx.x; pa[o] = ra; pd[o] = rd; }
}
int op_chkdt(cales_ctx *c, real *dtmax) {
  if (int e = materialize_visct(c)) return e;
  const int nbx = 8, np = nbx * c->n[2];
  real **f = c->f;
  if (c->C.impdiff == 2)
    LAUNCH(c, k_chkdt_partial<2>, dim3(nbx, c->n[2]), dim3(64, 4), 0, c->stream, c->g, 1. / c->dl[0], 1. / c->dl[1], c->visc, c->d_dzci,
                       c->d_dzfi, f[CALES_VISCT], f[CALES_U], f[CALES_V], f[CALES_W], c->d_red + 64, c->d_red + 64 + np);
  else if (c->C.impdiff == 1)
    LAUNCH(c, k_chkdt_partial<1>, dim3(nbx, c->n[2]), dim3(64, 4), 0, c->stream, c->g, 1. / c->dl[0], 1. / c->dl[1], c->visc, c->d_dzci,
                       c->d_dzfi, f[CALES_VISCT], f[CALES_U], f[CALES_V], f[CALES_W], c->d_red + 64, c->d_red + 64 + np);
  else
    LAUNCH(c, k_chkdt_partial<0>, dim3(nbx, c->n[2]), dim3(64, 4), 0, c->stream, c->g, 1. / c->dl[0], 1. / c->dl[1], c->visc, c->d_dzci,
                       c->d_dzfi, f[CALES_VISCT], f[CALES_U], f[CALES_V], f[CALES_W], c->d_red + 64, c->d_red + 64 + np);
  LAUNCH(c, k_fold, dim3(1), dim3(256), 0, c->stream, c->d_red + 64, np, 1, c->res, 0);
  LAUNCH(c, k_fold, dim3(1), dim3(256), 0, c->stream, c->d_red + 64 + np, np, 1, c->res, 1);
  if (int e = allreduce_res(c, 0, 2, 1)) return e;
  HIPCHK(c, hipMemcpyAsync(c->h_red, c->res, 2 * sizeof(real), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  real dti = c->h_red[0], dtid = c->h_red[1];
  if (dti == 0.) dti = 1.;
  if (dtid == 0.) dtid = CALES_EPS;
  *dtmax = std::fmin(0.4125 / dtid, 1.732 / dti);
  return 0;
}
