// Fused momentum r.h.s. + RK update (reference src/mom.f90:17-309 + src/rk.f90:77-94) as a plane-marching tile kernel.
//
// The kernel-per-loop version (k_mom + k_rk_update in k_stencil.hip) issues 55 + 13 vector loads per cell and is bound by the
// L1/TA rate, and it writes the r.h.s. only to read it back. Here a block of 64 x (TYM+2) threads owns a tile of 64 x TYM
// columns starting at i = 1 + 64 bx (whole 128-B lines in and out, see cales_create) and marches in k: every thread loads its
// own cell of u,v,w,visct,p per plane (next plane prefetched), lanes 0 and 63 also the x-halo cell beside it; the planes
// k-1..k+1 live in a 4-slot LDS ring (rows of 66) from which the 13/16-point stencils are read, and the RK update
// is applied in the same pass. Velocities are written to a second set of buffers (the stencil still needs the old values of
// the neighbours); the host swaps the pointers. The arithmetic block restates src/mom.f90:83-276 term by term in the reference's expression order (and keeps its
// names for the stencil values): that is what the 1e-13 parity with the reference's own r.h.s. rests on; everything around it is this design's.
// Algorithmic traffic: 14 words/cell (5 in + 3 old r.h.s. in + 3 velocities + 3 r.h.s. out) instead of 7 + 13.
#include "common.hpp"

#ifndef TYM
#define TYM 10      // measured at 512^3: 10 (12 waves, 120 KB of LDS, one block per CU) 2.89 ms, 6 (two blocks per CU) 2.92-3.00, 8: 3.17
#endif


struct MomRkArgs {
  const real *u, *v, *w, *s, *p, *duo, *dvo, *dwo;
  real *un, *vn, *wn, *du, *dv, *dw, *dud, *dvd, *dwd;
  const real *dzci, *dzfi;
  const real *cs;      // != nullptr: s holds |S| of the dynamic model and visct = s * cs(k) (see visct_lazy in common.hpp)
  real dxi, dyi, visc, f1, f2, f12, bfx, bfy, bfz;
  int kchunk;
  BandMap bm;      // block -> tile map of the 1-D launch (bm.gx = 0: plain 3-D grid)
  // low-storage RK3 (param.f90:27-29): the first substep has f2 = 0 -> the old r.h.s. is not read; inside cales_step the r.h.s. of the
  // third substep is never used (the next step starts with f2 = 0) -> not written
  int rd_old, wr_new;
  int perx;      // x periodic: the x-halo columns of the tiles at the ends of a row are the wrapped interior columns (valid whether or not the ghost columns are up to date)
  // CORR = 1 (cales_step without subgrid model, substeps 2 and 3): u, v, w are the PREDICTION u*, v*, w* of the substep before and the projection
  // (correc.f90:44-67 with the deferred bulk forcing) + pressure update (updatep.f90:30-47) happen while the planes are loaded: every INTERIOR cell
  // (1..n in all three directions) is corrected on its way into the LDS ring, u = (u* + f) - cfi (pp(i+1) - pp(i)) etc., the ghost cells hold the
  // final values already (op_bounduvw through the corrected view, k_bound.hip), and p + pp -- valid in the ghost cells too, both fields' conditions
  // being linear and homogeneous -- goes to the ring and, for the tile's own cells, to pn. The pass k_correc_cell<1> (9 words per cell) disappears.
  const real *pp; real *pn; real cfi, cfj, cdt; const real *force; int fmask;
};

// NOS = 1: no subgrid model (visct is identically zero, sgs.f90:62-68): its loads, LDS traffic and terms are compiled out.
// RD / WR: the old r.h.s. is read (weight f2 != 0) / the new one is stored -- compile-time, because EVERY global access of the plane loop is
// unconditional straight-line code: lanes, rows and planes outside the field are clamped onto valid cells on the way in and sent to the row's x ghost
// cell (which the ghost-cell update that always follows rewrites) on the way out. With loads and stores inside divergent branches the compiler
// cannot count the operations in flight and waits with s_waitcnt vmcnt(0): the wave then stalls at the top of every plane until the six to nine
// STORES of the plane before are acknowledged (on gfx9 stores count in vmcnt too), and the eddy viscosity scaled while loading (visct = |S| cs(k),
// a multiplication right behind the load) made the prefetch of plane k+2 a blocking one -- tools/memseq.py shows the order of accesses and waits.
// The two y-halo waves of a block (no outputs) run a loop of their own without stencil and stores; every wave fetches the two x-halo cells of ITS row
// for the five fields with its first ten lanes.
template <int IMP, typename OFF, int NOS, int RD, int WR, int CORR = 0>
__global__ __launch_bounds__(64 * (TYM + 2), (TYM <= 6 ? 4 : 3)) void k_momrk(Geom g, MomRkArgs A) {
  __shared__ real sh[4][4][TYM + 2][66];
  __shared__ real shp[3][TYM + 2][66];
  const int tx = threadIdx.x, ty = threadIdx.y;
  int bx_ = blockIdx.x, by_ = blockIdx.y, bz_ = blockIdx.z;
  if (A.bm.gx && !band_block(A.bm, bx_, by_, bz_)) return;
  const int i = bx_ * 64 + tx + 1, j = by_ * TYM + ty;
  const int kbeg = bz_ * A.kchunk + 1, kend = min(kbeg + A.kchunk - 1, g.n3);
  const bool outok = ty >= 1 && ty <= TYM && i <= g.n1 && j <= g.n2;
  // own cell, clamped into the field (a row that does not fill its last tile: the lane beside the last cell loads the column right of it -- the
  // wrapped first column when the ghost columns are not maintained; lanes further right load that column again, rows beyond n2+1 the last ghost row)
  const int ic = A.perx ? (i > g.n1 ? (i - 1) % g.n1 + 1 : i) : min(i, g.n1 + 1), jc = min(j, g.n2 + 1);      // (wrapped columns one after the other: the lane right of the last cell finds ITS right neighbour in the next lane, which the folded projection needs)
  const OFF c0 = (OFF)g.ix(ic, jc, 0) * RSZ;        // byte offsets (see ldb in common.hpp)
  const OFF cst = outok ? c0 : (OFF)g.ix(0, jc, 0) * RSZ;      // where this thread's results go: its cell, or the x ghost cell of its row (dead until the next ghost-cell update)
  const OFF sk = (OFF)g.s12 * RSZ;
  // x-halo cells (i = 64 bx and 64 bx + 65) of this wave's row: lanes 0..4 the five fields of the left one, lanes 5..9 of the right one
  const int hf_ = tx % 5, hside = (tx / 5) & 1, hxs = hside ? 65 : 0, hi0 = bx_ * 64 + (hside ? 65 : 0);
  const int hi_ = !A.perx ? hi0 : hi0 == 0 ? g.n1 : hi0 == g.n1 + 1 ? 1 : hi0;
  const bool hok = tx < 10 && hi0 <= g.n1 + 1 && j <= g.n2 + 1 && !(NOS && hf_ == 3);
  const real *hp = A.u + g.ix(ic, jc, 0);      // (lanes without a halo cell load their own cell of u again: every lane loads, nothing branches)
  if (hok) { const real *fp = hf_ == 0 ? A.u : hf_ == 1 ? A.v : hf_ == 2 ? A.w : hf_ == 3 ? A.s : A.p; hp = fp + g.ix(hi_, jc, 0); }
  const size_t sk64 = (size_t)g.s12;
  // CORR: pp of the own cell one plane ahead (q[5]; the plane's own value is rolled in ppc), pp of the cell above in y (q[6]); the halo lanes the two
  // pp values their field's correction needs (q[7], q[8]): the halo cell's own and the one beside it in the field's direction
  const bool rowin = j >= 1 && j <= g.n2;
  const OFF cup = CORR ? (OFF)g.ix(ic, min(jc + 1, g.n2 + 1), 0) * RSZ : 0;
  const real *hpa = A.pp, *hpb = A.pp; size_t hbk = 0; bool hin = false, hfm = false; real hcf = 0., hfo = 0.;
  real fo[3] = {0., 0., 0.};
  if (CORR) {
#pragma unroll
    for (int q = 0; q < 3; ++q) fo[q] = (A.fmask >> q & 1) ? ldc(A.force, q) : 0.;
    hpa = hpb = A.pp + g.ix(ic, jc, 0);
    if (hok) {
      const int hb_ = (A.perx && hi_ == g.n1) ? 1 : min(hi_ + 1, g.n1 + 1);
      hpa = A.pp + g.ix(hi_, jc, 0);
      hpb = hf_ == 0 ? A.pp + g.ix(hb_, jc, 0) : hf_ == 1 ? A.pp + g.ix(hi_, min(jc + 1, g.n2 + 1), 0) : hpa;
      hbk = hf_ == 2 ? 1 : 0;      // w: the plane above
      hin = rowin && (A.perx || (hi0 >= 1 && hi0 <= g.n1));
      hcf = hf_ == 0 ? A.cfi : hf_ == 1 ? A.cfj : 0.; hfo = hf_ < 3 ? fo[hf_] : 0.; hfm = hf_ < 3 && (A.fmask >> hf_ & 1);
    }
  }
  real ppc = 0.;      // CORR: pp of the own cell in the plane that put() takes next
  auto ld5 = [&](int k, real *q, real &h) {      // raw values: nothing is computed from them until put()
    const int kk = min(k, g.n3 + 1); const OFF c = c0 + (OFF)kk * sk;
    q[0] = ldb(A.u, c); q[1] = ldb(A.v, c); q[2] = ldb(A.w, c); q[3] = NOS ? 0. : ldb(A.s, c); q[4] = ldb(A.p, c);
    h = hp[(size_t)kk * sk64];
    if (CORR) {
      const int kn = min(k + 1, g.n3 + 1);
      q[5] = ldb(A.pp, c0 + (OFF)kn * sk); q[6] = ldb(A.pp, cup + (OFF)kk * sk);
      q[7] = hpa[(size_t)kk * sk64]; q[8] = hpb[(size_t)min(kk + (int)hbk, g.n3 + 1) * sk64];
    }
  };
  // plane kk of the five fields -> ring slot kk&3 (u,v,w,visct) and kk%3 (p); the eddy viscosity of the dynamic model's lazy form is |S| cs(k)
  auto put = [&](int kk, const real *q, real h) {
    const real csk = (!NOS && A.cs) ? ldc(A.cs, min(kk, g.n3 + 1)) : 1.;
    if (CORR) {
      const bool pin = kk >= 1 && kk <= g.n3, m = pin && rowin && (i <= g.n1 || (A.perx && i == g.n1 + 1));      // (the wrapped first column right of a short row's last cell is an interior cell)
      const real czk = A.cdt * ldc(A.dzci, min(kk, g.n3));
      real ppx = lane_next(ppc);
      const real pph = lane_bcast<5>(q[7]);      // pp at the tile's right halo column (lane 5: field u, right side)
      if (tx == 63) ppx = pph;
      // the expressions of k_correc_cell: (u + f) - fi (pp(i+1) - pp(i)); forcing in the inner cells only (all of these are)
      const real uc_ = ((A.fmask & 1) ? q[0] + fo[0] : q[0]) - A.cfi * (ppx - ppc);
      const real vc_ = ((A.fmask & 2) ? q[1] + fo[1] : q[1]) - A.cfj * (q[6] - ppc);
      const real wc_ = ((A.fmask & 4) ? q[2] + fo[2] : q[2]) - czk * (q[5] - ppc);
      sh[0][kk & 3][ty][tx + 1] = m ? uc_ : q[0]; sh[1][kk & 3][ty][tx + 1] = m ? vc_ : q[1]; sh[2][kk & 3][ty][tx + 1] = m ? wc_ : q[2];
      shp[kk % 3][ty][tx + 1] = CORR == 1 ? q[4] + ppc : q[4];      // (CORR = 2: the pressure was updated by a pass of its own, z-implicit diffusion)
      // halo lanes: everything but the one LDS write is computed by all lanes (a branch around arithmetic on loaded values would hold the waits for them)
      const real hcfk = hf_ == 2 ? czk : hcf;
      const real hc_ = (hfm ? h + hfo : h) - hcfk * (q[8] - q[7]);
      const real hv = hf_ == 4 ? (CORR == 1 ? h + q[7] : h) : ((hin && pin) ? hc_ : h);
      // (lanes without a halo cell write to their own place in the ring of the eddy viscosity, which this instantiation -- no subgrid model -- never reads)
      real *hd = !hok ? &sh[3][kk & 3][ty][tx + 1] : hf_ == 4 ? &shp[kk % 3][ty][hxs] : &sh[hf_ < 3 ? hf_ : 0][kk & 3][ty][hxs];
      *hd = hv;
      ppc = q[5];
      return;
    }
#pragma unroll
    for (int f = 0; f < 3; ++f) sh[f][kk & 3][ty][tx + 1] = q[f];
    if (!NOS) sh[3][kk & 3][ty][tx + 1] = q[3] * csk;
    shp[kk % 3][ty][tx + 1] = q[4];
    if (hok) { if (hf_ < 3) sh[hf_][kk & 3][ty][hxs] = h; else if (hf_ == 3) sh[3][kk & 3][ty][hxs] = h * csk; else shp[kk % 3][ty][hxs] = h; }
  };
  { real q[CORR ? 9 : 5], h;
    if (CORR) ppc = ldb(A.pp, c0 + (OFF)(kbeg - 1) * sk);
    ld5(kbeg - 1, q, h); put(kbeg - 1, q, h); ld5(kbeg, q, h); put(kbeg, q, h); ld5(kbeg + 1, q, h); put(kbeg + 1, q, h); }
  auto march = [&](auto halo_c) {
    constexpr bool HALO = decltype(halo_c)::value;      // a y-halo wave: loads and LDS only
  // CORR: the values stored by the plane before stay in their registers until this plane's loads are issued -- a register that is the data of a store in
  // flight may only be overwritten after a wait for that store (s_waitcnt vmcnt(0) right behind the barrier otherwise)
  real keep[7] = {0., 0., 0., 0., 0., 0., 0.};
  for (int k = kbeg; k <= kend; ++k) {
    __syncthreads();
    // the old r.h.s. of this plane first, then the prefetch of plane k+2: loads return in order, so waiting for the r.h.s. (needed at the end of this
    // iteration) leaves the prefetch in flight -- the other way round the wait for the r.h.s. would also be a wait for the prefetch
    real duo = 0., dvo = 0., dwo = 0.;
    if (!HALO && RD) { const OFF c = c0 + (OFF)k * sk; duo = ldb(A.duo, c); dvo = ldb(A.dvo, c); dwo = ldb(A.dwo, c); }
    real pf[CORR ? 9 : 5], hf;
    ld5(k + 2, pf, hf);                                     // prefetch, in flight during the stencil
    if (CORR && !HALO) asm volatile("" :: "v"(keep[0]), "v"(keep[1]), "v"(keep[2]), "v"(keep[3]), "v"(keep[4]), "v"(keep[5]), "v"(keep[6]));
    if (!HALO) {
      const OFF cs_ = cst + (OFF)k * sk;
      const int km = (k - 1) & 3, kc = k & 3, kp = (k + 1) & 3;
#define LS(f, sl, di, dj) sh[f][sl][ty + (dj)][tx + 1 + (di)]
      const real u_ccm = LS(0, km, 0, 0), u_cmc = LS(0, kc, 0, -1), u_mcc = LS(0, kc, -1, 0),
                   u_ccc = LS(0, kc, 0, 0), u_pcc = LS(0, kc, 1, 0), u_mpc = LS(0, kc, -1, 1), u_cpc = LS(0, kc, 0, 1), u_mcp = LS(0, kp, -1, 0),
                   u_ccp = LS(0, kp, 0, 0);
      const real v_ccm = LS(1, km, 0, 0), v_cmc = LS(1, kc, 0, -1), v_pmc = LS(1, kc, 1, -1), v_mcc = LS(1, kc, -1, 0),
                   v_ccc = LS(1, kc, 0, 0), v_pcc = LS(1, kc, 1, 0), v_cpc = LS(1, kc, 0, 1), v_cmp = LS(1, kp, 0, -1), v_ccp = LS(1, kp, 0, 0);
      const real w_ccm = LS(2, km, 0, 0), w_pcm = LS(2, km, 1, 0), w_cpm = LS(2, km, 0, 1), w_cmc = LS(2, kc, 0, -1), w_mcc = LS(2, kc, -1, 0),
                   w_ccc = LS(2, kc, 0, 0), w_pcc = LS(2, kc, 1, 0), w_cpc = LS(2, kc, 0, 1), w_ccp = LS(2, kp, 0, 0);
#define LSV(sl, di, dj) (NOS ? 0. : LS(3, sl, di, dj))
      const real s_ccm = LSV(km, 0, 0), s_pcm = LSV(km, 1, 0), s_cpm = LSV(km, 0, 1), s_cmc = LSV(kc, 0, -1), s_pmc = LSV(kc, 1, -1),
                   s_mcc = LSV(kc, -1, 0), s_ccc = LSV(kc, 0, 0), s_pcc = LSV(kc, 1, 0), s_mpc = LSV(kc, -1, 1), s_cpc = LSV(kc, 0, 1),
                   s_cmp = LSV(kp, 0, -1), s_mcp = LSV(kp, -1, 0), s_ccp = LSV(kp, 0, 0), s_ppc = LSV(kc, 1, 1), s_pcp = LSV(kp, 1, 0),
                   s_cpp = LSV(kp, 0, 1);
#undef LSV
#undef LS
      const int pc = k % 3, pn = (k + 1) % 3;
      const real p_ccc = shp[pc][ty][tx + 1], p_pcc = shp[pc][ty][tx + 2], p_cpc = shp[pc][ty + 1][tx + 1], p_ccp = shp[pn][ty][tx + 1];
      const real dxi = A.dxi, dyi = A.dyi, visc = A.visc;
      const real dzci_k = ldc(A.dzci, k), dzci_m = ldc(A.dzci, k - 1), dzfi_k = ldc(A.dzfi, k), dzfi_p = ldc(A.dzfi, k + 1);
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
      const real dudt_s = -(uu_ip - uu_im) * dxi - (vu_jp - vu_jm) * dyi - (wu_kp - wu_km) * dzfi_k +
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
      const real dvdt_s = -(uv_ip - uv_im) * dxi - (vv_jp - vv_jm) * dyi - (wv_kp - wv_km) * dzfi_k +
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
      const real dwdt_s = -(uw_ip - uw_im) * dxi - (vw_jp - vw_jm) * dyi - (ww_kp - ww_km) * dzci_k +
                            (visc_ip * (dwdx_ip + dudz_ip) - visc_im * (dwdx_im + dudz_im)) * dxi +
                            (visc_jp * (dwdy_jp + dvdz_jp) - visc_jm * (dwdy_jm + dvdz_jm)) * dyi +
                            (visc_kp * (dwdz_kp + dwdz_kp) - visc_km * (dwdz_km + dwdz_km)) * dzci_k;
      real du, dv, dw, dud = 0., dvd = 0., dwd = 0.;
      if (IMP == 2) { du = dudt_s + dudtd_xy; dv = dvdt_s + dvdtd_xy; dw = dwdt_s + dwdtd_xy; dud = dudtd_z; dvd = dvdtd_z; dwd = dwdtd_z; }   // mom.f90:278-284
      else if (IMP == 1) { du = dudt_s; dv = dvdt_s; dw = dwdt_s; dud = dudtd_xy + dudtd_z; dvd = dvdtd_xy + dvdtd_z; dwd = dwdtd_xy + dwdtd_z; }       // mom.f90:285-288
      else { du = dudt_s + dudtd_xy + dudtd_z; dv = dvdt_s + dvdtd_xy + dvdtd_z; dw = dwdt_s + dwdtd_xy + dwdtd_z; }                        // mom.f90:297-302
      // ---- RK update (rk.f90:81-91)
      real un = u_ccc + A.f1 * du + A.f2 * duo + A.f12 * (A.bfx - dxi * (p_pcc - p_ccc));
      real vn = v_ccc + A.f1 * dv + A.f2 * dvo + A.f12 * (A.bfy - dyi * (p_cpc - p_ccc));
      real wn = w_ccc + A.f1 * dw + A.f2 * dwo + A.f12 * (A.bfz - dzci_k * (p_ccp - p_ccc));
      if (IMP) { un = un + A.f12 * dud; vn = vn + A.f12 * dvd; wn = wn + A.f12 * dwd; stb(A.dud, cs_, dud); stb(A.dvd, cs_, dvd); stb(A.dwd, cs_, dwd); }
      stb(A.un, cs_, un); stb(A.vn, cs_, vn); stb(A.wn, cs_, wn);
      if (CORR) { if (CORR == 1) stb(A.pn, cs_, p_ccc);      // p + pp of the own cell (updatep.f90:30-47, explicit form)
                  keep[0] = un; keep[1] = vn; keep[2] = wn; keep[3] = CORR == 1 ? p_ccc : 0.; if (WR) { keep[4] = du; keep[5] = dv; keep[6] = dw; } }
      if (WR) { stb(A.du, cs_, du); stb(A.dv, cs_, dv); stb(A.dw, cs_, dw); }
    }
    put(k + 2, pf, hf);        // slots (k+2)&3 and (k+2)%3 were last read in iteration k-1, i.e. before this iteration's barrier
  }
  };
  // (the row of a wave is uniform: a scalar branch, each loop straight-line code)
  if (__builtin_amdgcn_readfirstlane((int)(ty == 0 || ty == TYM + 1))) march(std::true_type{}); else march(std::false_type{});
}

// The reference updates u,v,w in place, so their ghost cells keep the values of the last bounduvw until the next one -- and the wall model
// reads them when its sampling height lies between the wall and the first cell centre (index_wm = 1 or n: wmodel.f90:120-131 with i1 = 0 /
// n+1). The fused kernel writes the new velocities to a second set of buffers: with a wall model the ghost layers travel along.
struct GhostCopy { const real *src[3]; real *dst[3]; };
__global__ __launch_bounds__(256) void k_copy_ghosts(Geom g, GhostCopy G) {      // blockIdx.z = 6 (idir - 1) + 2 field + side; the grid spans the largest face
  const int idir = blockIdx.z / 6 + 1, fs = blockIdx.z % 6;
  const int na = idir == 1 ? g.n2 : g.n1, nb = idir == 3 ? g.n2 : g.n3, n = idir == 1 ? g.n1 : idir == 2 ? g.n2 : g.n3;
  const int a = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y * 4 + threadIdx.y, f = fs >> 1, side = fs & 1;
  if (a > na + 1 || b > nb + 1) return;
  const int m = side ? n + 1 : 0;
  const size_t c = idir == 1 ? g.ix(m, a, b) : idir == 2 ? g.ix(a, m, b) : g.ix(a, b, m);
  G.dst[f][c] = G.src[f][c];
}
// mom_xyz_ad + update of rk (rk.f90:74-94); leaves the new velocities in c->f[CALES_U..W] (pointers swapped with c->f2)
int op_momrk(cales_ctx *c, real f1, real f2, real f12) {
  ProfScope ps(c, "mom_rk_fused");
  const int *n = c->n; real **f = c->f;
  MomRkArgs A;
  A.u = f[CALES_U]; A.v = f[CALES_V]; A.w = f[CALES_W]; A.s = f[CALES_VISCT]; A.p = f[CALES_P];
  A.duo = f[CALES_DUDTO]; A.dvo = f[CALES_DVDTO]; A.dwo = f[CALES_DWDTO];
  A.un = c->f2[0]; A.vn = c->f2[1]; A.wn = c->f2[2];
  A.du = f[CALES_DUDT]; A.dv = f[CALES_DVDT]; A.dw = f[CALES_DWDT]; A.dud = f[CALES_DUDTD]; A.dvd = f[CALES_DVDTD]; A.dwd = f[CALES_DWDTD];
  A.cs = c->visct_lazy ? c->d_cs : nullptr;
  A.dzci = c->d_dzci; A.dzfi = c->d_dzfi; A.dxi = c->dli[0]; A.dyi = c->dli[1]; A.visc = c->visc;
  A.rd_old = f2 != 0.; A.wr_new = !c->skip_rhs_store;
  A.perx = c->step_xskip ? 1 : 0;      // (operator-level calls read the ghost columns the caller provided, as the reference does)
  A.f1 = f1; A.f2 = f2; A.f12 = f12; A.bfx = c->C.bforce[0]; A.bfy = c->C.bforce[1]; A.bfz = c->C.bforce[2];
  // the projection of the substep before is still pending (cales_step, fold_mom): this pass applies it while loading
  const bool corr = c->fold_mom_dtrk != 0.;
  A.pp = f[CALES_PP]; A.pn = c->scr1; A.cfi = c->fold_mom_dtrk * c->dli[0]; A.cfj = c->fold_mom_dtrk * c->dli[1]; A.cdt = c->fold_mom_dtrk;
  A.force = c->d_force; A.fmask = c->fold_mom_fmask;
  dim3 b(64, TYM + 2, 1), gr((n[0] + 63) / 64, (n[1] + TYM - 1) / TYM, 1);
  int kchunk = n[2];
  while ((long)gr.x * gr.y * ((n[2] + kchunk - 1) / kchunk) < tile_min_blocks(c) && kchunk > 32) kchunk = (kchunk + 1) / 2;
  // small grids: fewer blocks than one per CU leave most of the chip idle; shorter chunks (their three-plane prologue weighs more) beat that
  while ((long)gr.x * gr.y * ((n[2] + kchunk - 1) / kchunk) < 256 && kchunk > SMALL_KCH) kchunk = (kchunk + 1) / 2;
  kchunk = balanced_kchunk(c, (long)gr.x * gr.y, n[2], kchunk);
  if (int fk = tile_kchunk(c, (long)gr.x * gr.y, n[2])) kchunk = fk;
  gr.z = (n[2] + kchunk - 1) / kchunk; A.kchunk = kchunk;
  A.bm = BandMap{0, 0, 0, 0};
  if (band_wanted(gr.x)) { A.bm = band_map(gr.x, gr.y, gr.z); gr = dim3(band_blocks(A.bm), 1, 1); }
  const bool small = (c->ntot + 16) * sizeof(real) < (1ull << 32) && !c->fl.wide_offsets;      // 32-bit byte offsets
  const int nos = c->C.sgstype == 0 && c->visct_zero;     // visct known to be identically zero (never set by the host since the last zeroing)
#define MOMRK_L2(IMP_, RD_, WR_)                                                                                      \
  do {                                                                                                                 \
    if (small) { if (nos) LAUNCH(c, (k_momrk<IMP_, unsigned, 1, RD_, WR_>), gr, b, 0, c->stream, c->g, A);              \
                 else LAUNCH(c, (k_momrk<IMP_, unsigned, 0, RD_, WR_>), gr, b, 0, c->stream, c->g, A); }                \
    else { if (nos) LAUNCH(c, (k_momrk<IMP_, size_t, 1, RD_, WR_>), gr, b, 0, c->stream, c->g, A);                      \
           else LAUNCH(c, (k_momrk<IMP_, size_t, 0, RD_, WR_>), gr, b, 0, c->stream, c->g, A); }                        \
  } while (0)
  // (read the old r.h.s., store the new one): (0,1) first substep, (1,1) second, (1,0) third inside cales_step; (0,0) a first substep whose r.h.s. nobody keeps
#define MOMRK_LAUNCH(IMP_)                                                                                             \
  do {                                                                                                                 \
    if (A.rd_old && A.wr_new) MOMRK_L2(IMP_, 1, 1); else if (A.rd_old) MOMRK_L2(IMP_, 1, 0);                            \
    else if (A.wr_new) MOMRK_L2(IMP_, 0, 1); else MOMRK_L2(IMP_, 0, 0);                                                 \
  } while (0)
  const bool pdone = corr && c->fold_mom_pdone;      // the pressure was updated by a pass of its own (z-implicit diffusion): velocity only (CORR = 2)
  if (corr) {
    if (!(nos && ((c->C.impdiff == 0 && !pdone) || (c->C.impdiff == 2 && pdone)))) { c->err = "momrk: a pending projection needs the no-subgrid-model form, explicit or z-implicit"; return 1; }
    // periodic rows read without ghost columns: lane 63 of the last tile takes pp(i+1) from its right-halo lane, which has a cell of its own only while the
    // wrapped column n1+1 is not lane 63 itself -- rows shorter than a tile or whole tiles. step_xskip implies the radix-8 x plan (a power of two,
    // solver_can_fuse_fillps), so nothing reaches this today; the kernel's assumption is enforced here rather than left to that coincidence
    if (A.perx && n[0] % 64 == 63) { c->err = "momrk: the folded projection with wrapped x columns needs n1 % 64 != 63"; return 1; }
#define MOMRK_CORR(IMP_, CORR_, RD_, WR_) do { if (small) LAUNCH(c, (k_momrk<IMP_, unsigned, 1, RD_, WR_, CORR_>), gr, b, 0, c->stream, c->g, A); else LAUNCH(c, (k_momrk<IMP_, size_t, 1, RD_, WR_, CORR_>), gr, b, 0, c->stream, c->g, A); } while (0)
#define MOMRK_CORR4(IMP_, CORR_) do { if (A.rd_old && A.wr_new) MOMRK_CORR(IMP_, CORR_, 1, 1); else if (A.rd_old) MOMRK_CORR(IMP_, CORR_, 1, 0); else if (A.wr_new) MOMRK_CORR(IMP_, CORR_, 0, 1); else MOMRK_CORR(IMP_, CORR_, 0, 0); } while (0)
    if (pdone) MOMRK_CORR4(2, 2); else MOMRK_CORR4(0, 1);
#undef MOMRK_CORR4
#undef MOMRK_CORR
  } else
  if (c->C.impdiff == 2) MOMRK_LAUNCH(2); else if (c->C.impdiff == 1) MOMRK_LAUNCH(1); else MOMRK_LAUNCH(0);
#undef MOMRK_LAUNCH
#undef MOMRK_L2
  LAUNCHCHK(c);
  bool wm = false; for (int q = 0; q < 6; ++q) wm = wm || c->C.lwm[q] != 0;
  // (inside cales_step the only reader of those ghost layers before the next bounduvw rewrites them is a wall model that samples the ghost cell itself)
  if (wm && (!c->in_step || wm_samples_ghost(c) || c->fl.unmerged_bc)) {
    GhostCopy G; for (int q = 0; q < 3; ++q) { G.src[q] = c->f[CALES_U + q]; G.dst[q] = c->f2[q]; }
    const int na = std::max(n[0], n[1]), nb = std::max(n[1], n[2]);
    LAUNCH(c, k_copy_ghosts, dim3((na + 2 + 63) / 64, (nb + 2 + 3) / 4, 18), dim3(64, 4, 1), 0, c->stream, c->g, G);
    LAUNCHCHK(c);
  }
  for (int q = 0; q < 3; ++q) std::swap(c->f[CALES_U + q], c->f2[q]);
  if (corr) { if (!pdone) std::swap(c->f[CALES_P], c->scr1); c->fold_mom_dtrk = 0.; c->fold_mom_pdone = false; }      // the updated pressure (interior cells; the caller renews its ghost cells)
  return 0;
}
