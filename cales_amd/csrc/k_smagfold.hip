// Static Smagorinsky with the projection of the substep folded in (cales_step; reference src/correc.f90:44-67 + src/updatep.f90:30-47 +
// src/sgs.f90:84-152): ONE pass reads the prediction u*, v*, w*, the correction pressure pp and p, and writes the projected velocity, p + pp (with the
// z Laplacian of pp for z-implicit diffusion) and the eddy viscosity -- 10 words per cell instead of the 9 of the correction pass (k_correc_cell)
// plus the 4 of the Smagorinsky pass (k_smag_rows), and the strain rate is formed from planes that are on the chip anyway.
//
// Who is corrected where. The ghost cells of the prediction are FINAL before this pass runs: cales_step calls bounduvw(is_correc) right after the
// solve through the corrected view (common.hpp view_rd: sources that are interior cells are read as projected), wall model included, so ghost rows,
// ghost planes and -- where they are maintained -- ghost columns hold what the reference's correc + bounduvw leave there. This pass corrects every
// INTERIOR cell (1..n in the three directions) while it loads it, (u* + f) - dtrk grad(pp) with k_correc_cell's operations in their order, takes
// ghost rows and ghost planes as stored and wraps around in x (periodic x, whole 64-cell tiles). The corrected velocity of a tile's own cells goes to
// the second velocity buffers (neighbouring tiles still read u*); the host swaps the buffers and copies the ghost layers over.
// Tile, LDS ring, side job of the y-halo waves and the layout of the global accesses (every access of the plane loop unconditional straight-line code,
// a plane completed one iteration after its loads were issued) are those of k_corr_strain_tile (k_sgs.hip), whose strain-rate block this pass shares.
#include "common.hpp"
#include <type_traits>

#ifndef TYSF
#define TYSF 10
#endif

struct SmagFoldArgs {
  const real *u[3]; real *un[3];
  const real *pp; real *p; real *visct;
  const real *dzci, *dzfi, *zc, *del, *force;
  const real *twz, *twy;      // sqrt(tau_w) of the two z walls, twz[side][j][i] (pitch s1), and of the two y walls, twy[side][k][i] (k_wall_shear_view)
  real cfi, cfj, cdt, alpha, dxi, dyi, l3, visc, dl2, flo, fhi;
  int fmask, imp2, kchunk;
  int zlo, zhi, wmlo, wmhi;            // z walls (van Driest distance / shear); wall-model z faces: the strain rate sees ghost planes of u, v extrapolated from the interior
  int wylo, wyhi, wmylo, wmyhi;        // the same for y (ducts): ghost rows of u, w extrapolated (extrapolate(...,lwm), sgs.f90:683-748)
  BandMap bm;
};

// sqrt(tau_w) planes for the van Driest damping (sgs.f90:117-143) of the PROJECTED velocity, through the corrected view: blockIdx.z = 0 the two z walls
// (a = i, b = j), 1 the two y walls (a = i, b = k). Same arithmetic as smag_rows_body / k_wall_shear_y (k_sgs.hip).
__global__ __launch_bounds__(256) void k_wall_shear_view(Geom g, CorrView V, const real *__restrict__ u, const real *__restrict__ v, const real *__restrict__ w,
                                                         real visc, real dyi, real dzci0, real dzcin, int zlo, int zhi, int ylo, int yhi,
                                                         real *__restrict__ twz, real *__restrict__ twy) {
  const int i = blockIdx.x * 64 + threadIdx.x + 1, b = blockIdx.y * 4 + threadIdx.y + 1;
  if (i > g.n1) return;
  const int im1 = (V.perx && i == 1) ? g.n1 : i - 1;
  if (blockIdx.z == 0) {
    const int j = b; if (j > g.n2) return;
    if (zlo) {
      const real t1 = view_rd(g, V, 1, u, i, j, 1) - view_rd(g, V, 1, u, i, j, 0) + view_rd(g, V, 1, u, im1, j, 1) - view_rd(g, V, 1, u, im1, j, 0);
      const real t2 = view_rd(g, V, 2, v, i, j, 1) - view_rd(g, V, 2, v, i, j, 0) + view_rd(g, V, 2, v, i, j - 1, 1) - view_rd(g, V, 2, v, i, j - 1, 0);
      twz[(size_t)j * g.s1 + i] = sqrt(0.5 * visc * (sqrt(t1 * t1 + t2 * t2) * dzci0));
    }
    if (zhi) {
      const int n3 = g.n3;
      const real t1 = view_rd(g, V, 1, u, i, j, n3) - view_rd(g, V, 1, u, i, j, n3 + 1) + view_rd(g, V, 1, u, im1, j, n3) - view_rd(g, V, 1, u, im1, j, n3 + 1);
      const real t2 = view_rd(g, V, 2, v, i, j, n3) - view_rd(g, V, 2, v, i, j, n3 + 1) + view_rd(g, V, 2, v, i, j - 1, n3) - view_rd(g, V, 2, v, i, j - 1, n3 + 1);
      twz[(size_t)(g.n2 + 2 + j) * g.s1 + i] = sqrt(0.5 * visc * (sqrt(t1 * t1 + t2 * t2) * dzcin));
    }
  } else {
    const int k = b, n2 = g.n2; if (k > g.n3) return;
    if (ylo) {
      const real t1 = view_rd(g, V, 1, u, i, 1, k) - view_rd(g, V, 1, u, i, 0, k) + view_rd(g, V, 1, u, im1, 1, k) - view_rd(g, V, 1, u, im1, 0, k);
      const real t2 = view_rd(g, V, 3, w, i, 1, k) - view_rd(g, V, 3, w, i, 0, k) + view_rd(g, V, 3, w, i, 1, k - 1) - view_rd(g, V, 3, w, i, 0, k - 1);
      twy[(size_t)k * g.s1 + i] = sqrt(0.5 * visc * (sqrt(t1 * t1 + t2 * t2) * dyi));
    }
    if (yhi) {
      const real t1 = view_rd(g, V, 1, u, i, n2, k) - view_rd(g, V, 1, u, i, n2 + 1, k) + view_rd(g, V, 1, u, im1, n2, k) - view_rd(g, V, 1, u, im1, n2 + 1, k);
      const real t2 = view_rd(g, V, 3, w, i, n2, k) - view_rd(g, V, 3, w, i, n2 + 1, k) + view_rd(g, V, 3, w, i, n2, k - 1) - view_rd(g, V, 3, w, i, n2 + 1, k - 1);
      twy[(size_t)(g.n3 + 2 + k) * g.s1 + i] = sqrt(0.5 * visc * (sqrt(t1 * t1 + t2 * t2) * dyi));
    }
  }
}

template <typename OFF, int TY, int YW>      // YW = 1: walls or wall-model faces in y (ducts); the channel instantiation carries none of that logic
__global__ __launch_bounds__(64 * (TY + 2)) void k_corr_smag_tile(Geom g, SmagFoldArgs A) {
  __shared__ real ring[3][3][TY + 2][66];      // rows: x-halo cell, 64 own cells, x-halo cell
  // pp of the tile's cells, planes k+1 / k+2 (by parity), column 64 = the x-halo cell right of the row (see k_corr_strain_tile)
  __shared__ real sP[2][TY + 2][65];
  const int tx = threadIdx.x, ty = threadIdx.y;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (A.bm.gx && !band_block(A.bm, bx, by, bz)) return;
  const int n1 = g.n1, n2 = g.n2, n3 = g.n3;      // (n1 a multiple of 64: every tile is full in x, smag_fold_ok)
  const int i = bx * 64 + tx + 1, j = by * TY + ty;
  const int kbeg = bz * A.kchunk + 1, kend = min(kbeg + A.kchunk - 1, n3);
  const bool outok = ty >= 1 && ty <= TY && j <= n2;
  const int jq = min(j, n2 + 1);                      // the row this thread loads (rows beyond the field: the last ghost row again)
  const bool rraw = jq == 0 || jq == n2 + 1;          // a ghost row: final values, taken as stored
  const OFF sk = (OFF)g.s12 * RSZ;
  const OFF cl = (OFF)g.ix(i, jq, 0) * RSZ, cly = (OFF)g.ix(i, min(jq + 1, n2 + 1), 0) * RSZ;
  const OFF cdump = (OFF)g.ix(0, jq, 0) * RSZ;      // the x ghost cell of the row: where lanes / planes without an output of their own store (rewritten by the ghost-cell updates that follow the pass)
  const OFF cst = outok ? cl : cdump;
  // side job of the y-halo waves: lane tx < TY+2 completes the x-halo cell of tile row tx (other lanes repeat their own cell: no branches around loads)
  const bool hwave = ty == 0 || ty == TY + 1;
  const int sside = ty == 0 ? 0 : 1, hxs = sside ? 65 : 0, si0 = bx * 64 + (sside ? 65 : 0), sj0 = by * TY + tx;
  const bool sok = hwave && tx < TY + 2 && sj0 <= n2 + 1;
  const int si = !sok ? i : si0 == 0 ? n1 : si0 == n1 + 1 ? 1 : si0, sjr = !sok ? jq : min(sj0, n2 + 1);
  const bool sraw = sjr == 0 || sjr == n2 + 1;
  const int sxr = si >= n1 ? 1 : si + 1;
  const OFF so = (OFF)g.ix(si, sjr, 0) * RSZ, soy = (OFF)g.ix(si, min(sjr + 1, n2 + 1), 0) * RSZ, sox = (OFF)g.ix(sxr, sjr, 0) * RSZ;
  const int srow = sok ? tx : 0;
  const real f0 = (A.fmask & 1) ? ldc(A.force, 0) : 0., f1 = (A.fmask & 2) ? ldc(A.force, 1) : 0., f2 = (A.fmask & 4) ? ldc(A.force, 2) : 0.;
  struct Raw { real q[3], pz; };
  struct RawS { real q[3], pz, px, py; };
  auto rawload = [&](int kk, Raw &r) __attribute__((always_inline)) {      // kk: a plane index in 0..n3+1
    const OFF a = cl + (OFF)kk * sk;
#pragma unroll
    for (int q = 0; q < 3; ++q) r.q[q] = ldb(A.u[q], a);
    r.pz = ldb(A.pp, cl + (OFF)min(kk + 1, n3 + 1) * sk);
  };
  auto rawloads = [&](int kk, RawS &r) __attribute__((always_inline)) {
    const OFF a = so + (OFF)kk * sk;
#pragma unroll
    for (int q = 0; q < 3; ++q) r.q[q] = ldb(A.u[q], a);
    r.pz = ldb(A.pp, so + (OFF)min(kk + 1, n3 + 1) * sk); r.px = ldb(A.pp, sox + (OFF)kk * sk); r.py = ldb(A.pp, soy + (OFF)kk * sk);
  };
  // interior cell of plane kq: (u* + f) - dtrk grad(pp), k_correc_cell's operations in their order; raw: a ghost cell, as stored
  auto fix = [&](const real *q, real P0, real px, real py, real pz, int kq, bool raw, real *o) __attribute__((always_inline)) {
    const real a = ((A.fmask & 1) ? q[0] + f0 : q[0]) - A.cfi * (px - P0);
    const real b = ((A.fmask & 2) ? q[1] + f1 : q[1]) - A.cfj * (py - P0);
    const real c = ((A.fmask & 4) ? q[2] + f2 : q[2]) - A.cdt * ldc(A.dzci, kq) * (pz - P0);
    o[0] = raw ? q[0] : a; o[1] = raw ? q[1] : b; o[2] = raw ? q[2] : c;
  };
  // p + pp of plane kk (updatep.f90:30-47): explicit diffusion, or with the z Laplacian of pp (z-implicit; k_correc_cell<2>'s expression)
#define PNEW(pold, pm, pc, pn, kk) (!imp2 ? (pold) + (pc) : (pold) + (pc) + alpha * ((((pn) - (pc)) * ldc(A.dzci, (kk)) - ((pc) - (pm)) * ldc(A.dzci, (kk) - 1)) * ldc(A.dzfi, (kk))))
  const bool imp2 = A.imp2 != 0; const real alpha = A.alpha;
  // van Driest damping (sgs.f90:108-145): the nearer y wall of this row is known before the loop (the first one wins a tie), its shear plane is read for every k
  const int jg = j + g.jlo;
  real dminy = YW && A.wylo ? A.dl2 * (jg - 0.5) : CALES_BIG; int locy = 2;
  { const real d = YW && A.wyhi ? A.dl2 * (g.ng2 - jg + 0.5) : CALES_BIG; if (d < dminy) { dminy = d; locy = 3; } }
  const real *twp = (YW && A.twy) ? A.twy + (size_t)(locy == 3 ? n3 + 2 : 0) * g.s1 + i : A.del;
  const int tws = (YW && A.twy) ? g.s1 : 0;
  const int jt = min(max(j, 1), n2);
  const real tw_lo = A.zlo ? A.twz[(size_t)jt * g.s1 + i] : 0., tw_hi = A.zhi ? A.twz[(size_t)(n2 + 2 + jt) * g.s1 + i] : 0.;
  const bool exlo = YW && A.wmylo && j == 1, exhi = YW && A.wmyhi && j == n2;
  Raw rn; RawS rh = {}; real p0n, p0h = 0., pyt = 0., ppm = 0.;      // in flight at the top of iteration k: plane k+1 of the own / the side job's column, pp of its cell; ppm: pp of plane k of the own cell (z-implicit)
  { // ---- planes kbeg-1 and kbeg complete, plane kbeg+1 in flight (pp's neighbours by direct loads here: sP serves the loop)
    real c1[3], c0v[3], h1[3] = {0., 0., 0.}, h0[3] = {0., 0., 0.};
    const OFF clx = (OFF)g.ix(i >= n1 ? 1 : i + 1, jq, 0) * RSZ;
    const real Pm = ldb(A.pp, cl + (OFF)(kbeg - 1) * sk);
    { const int kq = kbeg; Raw r; rawload(kq, r); const real P0 = ldb(A.pp, cl + (OFF)kq * sk);
      fix(r.q, P0, ldb(A.pp, clx + (OFF)kq * sk), ldb(A.pp, cly + (OFF)kq * sk), r.pz, kq, rraw, c1); p0n = r.pz;
      const OFF a = cst + (OFF)kbeg * sk; stb(A.p, a, PNEW(ldb(A.p, a), Pm, P0, r.pz, kq));      // lanes without output: their ghost cell
      ppm = P0;
      if (hwave) { RawS e; rawloads(kq, e); const real E0 = ldb(A.pp, so + (OFF)kq * sk); fix(e.q, E0, e.px, e.py, e.pz, kq, sraw, h1); p0h = e.pz; } }
    { const int kq = kbeg - 1; Raw r; rawload(kq, r);
      fix(r.q, Pm, ldb(A.pp, clx + (OFF)kq * sk), ldb(A.pp, cly + (OFF)kq * sk), r.pz, kq, rraw || kq == 0, c0v);
      if (hwave) { RawS e; rawloads(kq, e); const real E0 = ldb(A.pp, so + (OFF)kq * sk); fix(e.q, E0, e.px, e.py, e.pz, kq, sraw || kq == 0, h0); } }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      ring[(kbeg - 1) % 3][q][ty][tx + 1] = c0v[q]; ring[kbeg % 3][q][ty][tx + 1] = c1[q];
      if (sok) { ring[(kbeg - 1) % 3][q][srow][hxs] = h0[q]; ring[kbeg % 3][q][srow][hxs] = h1[q]; }
      stb(A.un[q], cst + (OFF)kbeg * sk, c1[q]);
    }
    sP[(kbeg + 1) & 1][ty][tx] = p0n;      // pp of plane kbeg+1
    if (sok && sside) sP[(kbeg + 1) & 1][srow][64] = p0h;
    rawload(min(kbeg + 1, n3 + 1), rn);
    if (hwave) { rawloads(min(kbeg + 1, n3 + 1), rh); pyt = ldb(A.pp, cly + (OFF)min(kbeg + 1, n3 + 1) * sk); }
    __syncthreads();
  }
  int km = (kbeg - 1) % 3, kc = kbeg % 3, kp = (kbeg + 1) % 3;
  real keep[2] = {0., 0.};
  real twyc = YW ? twp[(size_t)kbeg * tws] : 0.;
  // one plane. HALO: a y-halo wave (side job, no outputs); TOP: plane k+1 is the ghost plane n3+1 (k = n3): taken as stored
  // (tw_lo, tw_hi by value and twyc read into a local first: a conditional between variables captured by reference becomes a select between POINTERS
  //  into the closure, which keeps every captured variable of the kernel in scratch memory -- 1 KB per lane, measured)
  auto plane = [&, tw_lo, tw_hi](const int k, auto halo_c, auto top_c) __attribute__((always_inline)) {
    const real twy_now = twyc;
    constexpr bool HALO = decltype(halo_c)::value, TOP = decltype(top_c)::value;
    const OFF idx = cst + (OFF)k * sk;
    const real twyn = (YW && !HALO) ? twp[(size_t)min(k + 1, n3) * tws] : 0.;
    const int par = (k + 1) & 1, k2 = min(k + 2, n3 + 1);
    { // plane k+1, loaded during the last iteration, is completed, stored if it belongs to this chunk, and plane k+2 goes into flight
      real cc[3];
      real px = lane_next(p0n);
      if (tx == 63) px = sP[par][ty][64];
      const real py = (HALO && ty == TY + 1) ? pyt : sP[par][ty + (ty == TY + 1 ? 0 : 1)][tx];
      fix(rn.q, p0n, px, py, rn.pz, k + 1, TOP || rraw, cc);
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
      // (the data registers of the last plane's stores stay allocated up to here, see k_corr_strain_tile)
      if (!HALO) asm volatile("" :: "v"(keep[0]), "v"(keep[1]));
      if (HALO) {
        real hh[3];
        fix(rh.q, p0h, sside ? rh.px : sP[par][srow][0], rh.py, rh.pz, k + 1, TOP || sraw, hh);
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
    if (!HALO) {
      // p + pp of plane k+1: load, wait and store inside one iteration; pp(k+1) and pp(k+2) of the cell are still in sP
      const OFF dstp = (outok && k + 1 <= kend ? cst : cdump) + (OFF)(k + 1) * sk;
      const real pl = ldb(A.p, dstp);
#define RU(dk, dj, di) ring[dk][0][ty + (dj)][tx + 1 + (di)]
#define RV(dk, dj, di) ring[dk][1][ty + (dj)][tx + 1 + (di)]
#define RW(dk, dj, di) ring[dk][2][ty + (dj)][tx + 1 + (di)]
      // strain rate (sgs.f90:571-630), expressions and their order as in k_strain_tile; at wall-model faces the ghost values it sees are extrapolated
      // from the interior (extrapolate(...,lwm), sgs.f90:683-748: along y u and w by 2 Q(1) - Q(2), then along z u and v with the grid factor) --
      // formed here from the stencil's own values instead of rewriting ghost cells
      const bool zl = A.wmlo && k == 1, zh = A.wmhi && k == n3;
      const real dxi = A.dxi, dyi = A.dyi, zc = ldc(A.dzci, k), zm = ldc(A.dzci, k - 1);
      const real u_ccc = RU(kc, 0, 0), u_mcc = RU(kc, 0, -1), v_ccc = RV(kc, 0, 0), v_cmc = RV(kc, -1, 0), w_ccc = RW(kc, 0, 0), w_ccm = RW(km, 0, 0);
      const real s11 = (u_ccc - u_mcc) * dxi, s22 = (v_ccc - v_cmc) * dyi, s33 = (w_ccc - w_ccm) * ldc(A.dzfi, k);
      real s12, s13, s23;
      { real u_mmc = RU(kc, -1, -1), u_cmc = RU(kc, -1, 0), u_mpc = RU(kc, 1, -1), u_cpc = RU(kc, 1, 0);
        const real v_mmc = RV(kc, -1, -1), v_pmc = RV(kc, -1, 1), v_mcc = RV(kc, 0, -1), v_pcc = RV(kc, 0, 1);
        if (YW) {
          const real a0 = 2. * u_mcc - u_mpc, a1 = 2. * u_ccc - u_cpc, b0 = 2. * u_mcc - u_mmc, b1 = 2. * u_ccc - u_cmc;
          if (exlo) { u_mmc = a0; u_cmc = a1; }
          if (exhi) { u_mpc = b0; u_cpc = b1; }
        }
        s12 = .125 * ((u_cpc - u_ccc) * dyi + (v_pcc - v_ccc) * dxi + (u_ccc - u_cmc) * dyi + (v_pmc - v_cmc) * dxi +
                      (u_mpc - u_mcc) * dyi + (v_ccc - v_mcc) * dxi + (u_mcc - u_mmc) * dyi + (v_cmc - v_mmc) * dxi); }
      __builtin_amdgcn_sched_barrier(0);
      { real u_mcm = RU(km, 0, -1), u_ccm = RU(km, 0, 0), u_mcp = RU(kp, 0, -1), u_ccp = RU(kp, 0, 0);
        const real w_mcm = RW(km, 0, -1), w_pcm = RW(km, 0, 1), w_mcc = RW(kc, 0, -1), w_pcc = RW(kc, 0, 1);
        { const real a0 = (1. + A.flo) * u_mcc - A.flo * u_mcp, a1 = (1. + A.flo) * u_ccc - A.flo * u_ccp;
          const real b0 = (1. + A.fhi) * u_mcc - A.fhi * u_mcm, b1 = (1. + A.fhi) * u_ccc - A.fhi * u_ccm;
          if (zl) { u_mcm = a0; u_ccm = a1; }
          if (zh) { u_mcp = b0; u_ccp = b1; } }
        s13 = .125 * ((u_ccp - u_ccc) * zc + (w_pcc - w_ccc) * dxi + (u_ccc - u_ccm) * zm + (w_pcm - w_ccm) * dxi +
                      (u_mcp - u_mcc) * zc + (w_ccc - w_mcc) * dxi + (u_mcc - u_mcm) * zm + (w_ccm - w_mcm) * dxi); }
      __builtin_amdgcn_sched_barrier(0);
      { real v_cmm = RV(km, -1, 0), v_ccm = RV(km, 0, 0), v_cmp = RV(kp, -1, 0), v_ccp = RV(kp, 0, 0);
        real w_cmm = RW(km, -1, 0), w_cpm = RW(km, 1, 0), w_cmc = RW(kc, -1, 0), w_cpc = RW(kc, 1, 0);
        if (YW) {
          const real a0 = 2. * w_ccm - w_cpm, a1 = 2. * w_ccc - w_cpc, b0 = 2. * w_ccm - w_cmm, b1 = 2. * w_ccc - w_cmc;
          if (exlo) { w_cmm = a0; w_cmc = a1; }
          if (exhi) { w_cpm = b0; w_cpc = b1; }
        }
        { const real a0 = (1. + A.flo) * v_cmc - A.flo * v_cmp, a1 = (1. + A.flo) * v_ccc - A.flo * v_ccp;
          const real b0 = (1. + A.fhi) * v_cmc - A.fhi * v_cmm, b1 = (1. + A.fhi) * v_ccc - A.fhi * v_ccm;
          if (zl) { v_cmm = a0; v_ccm = a1; }
          if (zh) { v_cmp = b0; v_ccp = b1; } }
        s23 = .125 * ((v_ccp - v_ccc) * zc + (w_cpc - w_ccc) * dyi + (v_ccc - v_ccm) * zm + (w_cpm - w_ccm) * dyi +
                      (v_cmp - v_cmc) * zc + (w_ccc - w_cmc) * dyi + (v_cmc - v_cmm) * zm + (w_ccm - w_cmm) * dyi); }
#undef RU
#undef RV
#undef RW
      const real s0v = sqrt(2. * (s11 * s11 + s22 * s22 + s33 * s33 + 2. * (s12 * s12 + s13 * s13 + s23 * s23)));
      real fd = 1.;
      if (A.zlo || A.zhi || (YW && (A.wylo || A.wyhi))) {     // nearest wall in the order y-, y+, z-, z+: the first one wins a tie (minloc, sgs.f90:116)
        real dmin = dminy; int loc = locy;
        { const real d = A.zlo ? ldc(A.zc, k) : CALES_BIG; if (d < dmin) { dmin = d; loc = 4; } }
        { const real d = A.zhi ? A.l3 - ldc(A.zc, k) : CALES_BIG; if (d < dmin) { dmin = d; loc = 5; } }
        const real tw = loc < 4 ? twy_now : loc == 4 ? tw_lo : tw_hi;
        const real dw_plus = dmin * tw * (1. / A.visc);
        fd = 1. - exp(-dw_plus / 25.);
      }
      const real t = 0.11 * ldc(A.del, k) * fd;      // c_smag, src/param.f90:33
      keep[0] = (t * t) * s0v;
      stb(A.visct, idx, keep[0]);
      { const real pc_ = sP[par][ty][tx], pn_ = sP[par ^ 1][ty][tx]; keep[1] = PNEW(pl, ppm, pc_, pn_, k + 1); ppm = pc_; }
      stb(A.p, dstp, keep[1]);
      twyc = twyn;
    }
    __syncthreads();      // the ring slot of plane k-1 is overwritten by the next iteration
    const int t_ = km; km = kc; kc = kp; kp = t_;
  };
  const std::true_type T_; const std::false_type F_;
  const bool topw = kend == n3;      // the chunk's last plane sits under the ghost plane n3+1
  const int klast = topw ? kend - 1 : kend;
  // (a wave's row is uniform: scalar branches, each loop straight-line code; the first plane peeled off, see k_corr_strain_tile)
  if (__builtin_amdgcn_readfirstlane((int)hwave)) {
    if (kbeg <= klast) plane(kbeg, T_, F_);
    for (int k = kbeg + 1; k <= klast; ++k) plane(k, T_, F_);
    if (topw) plane(kend, T_, T_);
  } else {
    if (kbeg <= klast) plane(kbeg, F_, F_);
    for (int k = kbeg + 1; k <= klast; ++k) plane(k, F_, F_);
    if (topw) plane(kend, F_, T_);
  }
#undef PNEW
}

// the static conditions under which cales_step folds the projection into the Smagorinsky pass
bool smag_fold_ok(const cales_ctx *c) {
  if (c->C.sgstype != 1 || c->fl.smag_reference_sequence || c->fl.unfolded_correc || c->fl.unfused_correc || c->fl.wide_offsets) return false;
  if (!(c->C.impdiff == 0 || c->C.impdiff == 2) || c->P != 1 || c->n[0] % 64 != 0 || c->n[1] < 2 || c->n[2] < 3) return false;
  if ((c->ntot + 16) * sizeof(real) >= (1ull << 32)) return false;      // 32-bit byte offsets
  for (int q = 0; q < 2; ++q) if (c->is_wall[q] != 0. || c->C.lwm[q] != 0) return false;      // walls or wall model in x: the general path
  if (!(CBP(c, 0, 1) == 'P' && CBP(c, 1, 1) == 'P')) return false;
  for (int iv = 1; iv <= 3; ++iv) if (!(CBV(c, 0, 1, iv) == 'P' && CBV(c, 1, 1, iv) == 'P')) return false;
  // homogeneous pressure conditions: the projection then leaves the prescribed normal velocity of a wall alone and the ghost cells of pp carry no boundary term
  for (int d = 2; d <= 3; ++d) {
    const bool per = CBP(c, 0, d) == 'P' && CBP(c, 1, d) == 'P', neu = CBP(c, 0, d) == 'N' && CBP(c, 1, d) == 'N' && c->C.bcpre[2 * (d - 1)] == 0. && c->C.bcpre[2 * (d - 1) + 1] == 0.;
    if (!per && !neu) return false;
  }
  // scratch for the wall-shear planes
  if ((size_t)2 * (c->n[2] + 2 + c->n[1] + 2) * c->g.s1 > c->ntot) return false;
  return true;
}

int op_smag_fold(cales_ctx *c) {
  const int *n = c->n; real **f = c->f;
  if (c->fold_dtrk == 0.) { c->err = "smag_fold: no projection pending"; return 1; }
  if (!c->d_del) { c->err = "smag_fold: filter widths not set"; return 1; }
  SmagFoldArgs S = {};
  for (int q = 0; q < 3; ++q) { S.u[q] = f[CALES_U + q]; S.un[q] = c->f2[q]; }
  S.pp = f[CALES_PP]; S.p = f[CALES_P]; S.visct = f[CALES_VISCT];
  S.dzci = c->d_dzci; S.dzfi = c->d_dzfi; S.zc = c->d_zc; S.del = c->d_del; S.force = c->d_force;
  S.cdt = c->fold_dtrk; S.cfi = c->fold_dtrk * c->dli[0]; S.cfj = c->fold_dtrk * c->dli[1]; S.alpha = c->fold_alpha; S.imp2 = c->C.impdiff == 2 ? 1 : 0;
  S.fmask = c->fold_mom_fmask;
  S.dxi = c->dli[0]; S.dyi = c->dli[1]; S.l3 = c->C.l[2]; S.visc = c->visc; S.dl2 = c->dl[1];
  S.zlo = c->is_wall[4] != 0.; S.zhi = c->is_wall[5] != 0.;
  S.wmlo = ISB(c, 0, 3) && LWM(c, 0, 3) != 0; S.wmhi = ISB(c, 1, 3) && LWM(c, 1, 3) != 0;
  S.flo = (1. / c->dzci[0]) * c->dzci[1]; S.fhi = (1. / c->dzci[n[2]]) * c->dzci[n[2] - 1];
  S.wylo = c->is_wall[2] != 0.; S.wyhi = c->is_wall[3] != 0.;
  S.wmylo = ISB(c, 0, 2) && LWM(c, 0, 2) != 0; S.wmyhi = ISB(c, 1, 2) && LWM(c, 1, 2) != 0;
  const bool yw = S.wylo || S.wyhi || S.wmylo || S.wmyhi;
  // wall-shear planes of the projected velocity (van Driest), through the corrected view
  real *twy = c->wk[0], *twz = c->wk[0] + (size_t)2 * (n[2] + 2) * c->g.s1;
  S.twy = (S.wylo || S.wyhi) ? twy : nullptr; S.twz = twz;
  if (S.zlo || S.zhi || S.wylo || S.wyhi) {
    const real keep_view = c->bc_view_dtrk; c->bc_view_dtrk = c->fold_dtrk;
    const CorrView V = corr_view(c);
    c->bc_view_dtrk = keep_view;
    const int nb = std::max(n[1], n[2]);
    LAUNCH(c, k_wall_shear_view, dim3((n[0] + 63) / 64, (nb + 3) / 4, 2), dim3(64, 4), 0, c->stream, c->g, V, f[CALES_U], f[CALES_V], f[CALES_W], c->visc, c->dli[1],
           c->dzci[0], c->dzci[n[2]], S.zlo, S.zhi, S.wylo, S.wyhi, twz, twy);
  }
  { ProfScope ps(c, "correc_smag_fused");
    dim3 mb(64, TYSF + 2, 1), mg(n[0] / 64, (n[1] + TYSF - 1) / TYSF, 1);
    int kch = n[2];
    while ((long)mg.x * mg.y * ((n[2] + kch - 1) / kch) < tile_min_blocks(c) && kch > 32) kch = (kch + 1) / 2;
    while ((long)mg.x * mg.y * ((n[2] + kch - 1) / kch) < 256 && kch > 8) kch = (kch + 1) / 2;
    if (int fk = tile_kchunk(c, (long)mg.x * mg.y, n[2])) kch = fk;
    mg.z = (n[2] + kch - 1) / kch; S.kchunk = kch;
    S.bm = BandMap{0, 0, 0, 0};
    if (band_wanted(mg.x)) { S.bm = band_map(mg.x, mg.y, mg.z); mg = dim3(band_blocks(S.bm), 1, 1); }
    if (yw) LAUNCH(c, (k_corr_smag_tile<unsigned, TYSF, 1>), mg, mb, 0, c->stream, c->g, S);
    else LAUNCH(c, (k_corr_smag_tile<unsigned, TYSF, 0>), mg, mb, 0, c->stream, c->g, S); }
  LAUNCHCHK(c);
  // the projected velocity sits in the second buffers: swap, and give the new buffers the ghost layers -- final since the bounduvw through the view
  for (int q = 0; q < 3; ++q) std::swap(c->f[CALES_U + q], c->f2[q]);
  if (int e = op_copy_ghosts(c, c->f2, c->f + CALES_U)) return e;
  return op_boundp(c, c->f[CALES_P], 0);
}
