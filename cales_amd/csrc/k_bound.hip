// Ghost-cell kernels: bounduvw / boundp / set_bc (reference src/bound.f90:18-399), the single-rank
// image of updthalo (src/bound.f90:619-696), updt_rhs_b (src/bound.f90:562-617) and the log-law /
// laminar wall model (src/wmodel.f90:19-335).
//
// The reference issues one tiny `!$acc kernels` region per face and field (39 regions). Here all faces
// of one direction (up to 3 fields x 2 sides) go into ONE launch; the x -> y -> z order of the
// reference is kept because corner ghosts depend on it.
#include "common.hpp"

// Corrected view (cales_step, projection folded into the next momentum pass): a source cell that is an INTERIOR cell (1..n in all three directions) is
// read as the projected velocity (u* + f) - dtrk grad(pp) of component comp = 1..3 (the expressions of k_correc_cell), every other cell as stored --
// ghost cells written by an earlier direction of the same update are final already. comp = 0: plain reads.
struct CorrView { const real *pp, *dzci, *force; real cfi, cfj, dt; int fmask, perx; };      // force: the bulk-forcing increments of the substep (device), fmask: the forced components
__device__ inline real view_rd(const Geom &g, const CorrView &V, int comp, const real *p, int i, int j, int k) {
  const size_t c = g.ix(i, j, k);
  const real v = p[c];
  if (!comp || i < 1 || i > g.n1 || j < 1 || j > g.n2 || k < 1 || k > g.n3) return v;
  const real pc = V.pp[c];
  const real pb = comp == 1 ? V.pp[(V.perx && i == g.n1) ? g.ix(1, j, k) : c + 1] : comp == 2 ? V.pp[c + g.s1] : V.pp[c + g.s12];
  const real cf = comp == 1 ? V.cfi : comp == 2 ? V.cfj : V.dt * V.dzci[k];
  return ((V.fmask >> (comp - 1) & 1) ? v + V.force[comp - 1] : v) - cf * (pb - pc);
}
static CorrView corr_view(const cales_ctx *c) {
  CorrView V{};
  V.pp = c->f[CALES_PP]; V.dzci = c->d_dzci; V.force = c->d_force; V.cfi = c->bc_view_dtrk * c->dli[0]; V.cfj = c->bc_view_dtrk * c->dli[1]; V.dt = c->bc_view_dtrk;
  V.fmask = c->fold_mom_fmask; V.perx = c->step_xskip ? 1 : 0;
  return V;
}
struct BcJob {
  real *p;          // field
  const real *bc;   // plane of side `ibound` (already offset), (0:na+1,0:nb+1)
  real dr;
  char ctype;         // 'P','D','N'
  char centered;
  char ibound;
  char vcomp = 0;     // != 0: sources through the corrected view of this velocity component
};
struct BcJobs { int njobs, idir; BcJob job[6]; CorrView V; };

__global__ __launch_bounds__(256) void k_set_bc(Geom g, BcJobs J) {
  const BcJob jb = J.job[blockIdx.z];
  const int idir = J.idir;
  const int na = idir == 1 ? g.n2 : g.n1, nb = idir == 3 ? g.n2 : g.n3, n = idir == 1 ? g.n1 : idir == 2 ? g.n2 : g.n3;
  const int a = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y * 4 + threadIdx.y;
  if (a > na + 1 || b > nb + 1) return;
  const long st = idir == 1 ? 1 : idir == 2 ? (long)g.s1 : g.s12;
  const size_t base = idir == 1 ? g.ix(0, a, b) : idir == 2 ? g.ix(a, 0, b) : g.ix(a, b, 0);
  real *p = jb.p + base;
#define P(m) p[(long)(m)*st]
  // sources (cells 1, n-1, n of the line): through the corrected view when the job asks for it
  auto R = [&](int m) -> real {
    if (!jb.vcomp) return P(m);
    return idir == 1 ? view_rd(g, J.V, jb.vcomp, jb.p, m, a, b) : idir == 2 ? view_rd(g, J.V, jb.vcomp, jb.p, a, m, b) : view_rd(g, J.V, jb.vcomp, jb.p, a, b, m);
  };
  const real bcv = jb.bc ? jb.bc[a + (size_t)(na + 2) * b] : 0.;
  const real sgn = (jb.ctype == 'D' && jb.centered) ? -1. : 1.;
  switch (jb.ctype) {
  case 'P':   // bound.f90:220-248 (one job handles both ends)
    { const real lo = R(n), hi = R(1); P(0) = lo; P(n + 1) = hi; } break;
  case 'D':   // bound.f90:249-319
    if (jb.centered) { if (jb.ibound == 0) P(0) = 2. * bcv + sgn * R(1); else P(n + 1) = 2. * bcv + sgn * R(n); }
    else { if (jb.ibound == 0) P(0) = bcv; else { P(n + 1) = R(n - 1); P(n) = bcv; } }
    break;
  case 'N':   // bound.f90:320-396
    if (jb.centered) { if (jb.ibound == 0) P(0) = -jb.dr * bcv + sgn * R(1); else P(n + 1) = jb.dr * bcv + sgn * R(n); }
    else { if (jb.ibound == 0) P(0) = -jb.dr * bcv + R(1); else { P(n + 1) = R(n); P(n) = jb.dr * bcv + R(n - 1); } }
    break;
  }
#undef P
}

static int launch_jobs(cales_ctx *c, BcJobs &J) {
  if (J.njobs == 0) return 0;
  const int na = J.idir == 1 ? c->n[1] : c->n[0], nb = J.idir == 3 ? c->n[1] : c->n[2];
  dim3 b(64, 4, 1), gr((na + 2 + 63) / 64, (nb + 2 + 3) / 4, J.njobs);
  LAUNCH(c, k_set_bc, gr, b, 0, c->stream, c->g, J);
  LAUNCHCHK(c);
  return 0;
}
static inline const real *plane(const DBound &b, int idir, int ibound, const int *n) {
  const size_t pl = idir == 1 ? (size_t)(n[1] + 2) * (n[2] + 2) : idir == 2 ? (size_t)(n[0] + 2) * (n[2] + 2) : (size_t)(n[0] + 2) * (n[1] + 2);
  const real *base = idir == 1 ? b.x : idir == 2 ? b.y : b.z;
  return base + (size_t)ibound * pl;
}
static inline void add_job(BcJobs &J, real *p, char ctype, int ibound, int centered, const real *bc, real dr) {
  BcJob &j = J.job[J.njobs++];
  j.p = p; j.bc = bc; j.dr = dr; j.ctype = ctype; j.centered = (char)centered; j.ibound = (char)ibound; j.vcomp = 0;
}

// ---- all three directions in ONE launch, for the common BC sets: x periodic, y periodic (or exchanged between slabs), any
// pointwise z condition. The reference applies x, then y, then z, and later directions also fill the corner ghosts of earlier ones
// (bound.f90:158-199). With periodic x and y that sequence is a composition with a closed form -- every ghost cell is the z
// operation applied to the periodically wrapped interior cell -- so each thread computes its ghost cell from interior cells only
// and no ordering between the directions is left. Saves two of the three launches of every bounduvw / boundp (19 -> 7 per substep
// for a channel), which is what small grids are bound by.
struct MField { real *p; const real *bc0, *bc1; real dr0, dr1; char t0, t1, centered, vcomp = 0; };      // vcomp != 0: sources through the corrected view (CorrView) of this velocity component      // z sides: 'P','D','N' or 0 (leave z alone)
struct MJobs { int nf, do_x, wrap_y, do_z; MField f[8]; CorrView V; };
__global__ __launch_bounds__(256) void k_bc_merged(Geom g, MJobs J) {
  const int region = blockIdx.z % 3; const MField F = J.f[blockIdx.z / 3];
  const int n1 = g.n1, n2 = g.n2, n3 = g.n3;
  const int a = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y * 4 + threadIdx.y;
  auto wx = [&](int i) { return !J.do_x ? i : i == 0 ? n1 : i == n1 + 1 ? 1 : i; };
  auto wy = [&](int j) { return !J.wrap_y ? j : j == 0 ? n2 : j == n2 + 1 ? 1 : j; };
  const bool top_face = J.do_z && F.t1 == 'D' && !F.centered;      // plane n3 itself is boundary data (face-centred normal component)
  const bool zset0 = J.do_z && F.t0 != 0, zset1 = J.do_z && F.t1 != 0;      // the z condition sets the ghost plane below / above
  real *p = F.p;
  auto S = [&](int i, int j, int k) -> real { return F.vcomp ? view_rd(g, J.V, F.vcomp, F.p, i, j, k) : F.p[g.ix(i, j, k)]; };      // sources
  if (region == 0) {               // z ghost planes of the column (a, b), ghost columns included
    if (!J.do_z || a > n1 + 1 || b > n2 + 1) return;
    const int ia = wx(a), jb = wy(b);
    const size_t q2 = (size_t)a + (size_t)(n1 + 2) * b;
    if (F.t0 == 'P') { p[g.ix(a, b, 0)] = S(ia, jb, n3); p[g.ix(a, b, n3 + 1)] = S(ia, jb, 1); return; }
    if (F.t0 == 'D') p[g.ix(a, b, 0)] = F.centered ? 2. * F.bc0[q2] - S(ia, jb, 1) : F.bc0[q2];
    else if (F.t0 == 'N') p[g.ix(a, b, 0)] = -F.dr0 * F.bc0[q2] + S(ia, jb, 1);
    if (F.t1 == 'D') {
      if (F.centered) p[g.ix(a, b, n3 + 1)] = 2. * F.bc1[q2] - S(ia, jb, n3);
      else { p[g.ix(a, b, n3 + 1)] = S(ia, jb, n3 - 1); p[g.ix(a, b, n3)] = F.bc1[q2]; }
    } else if (F.t1 == 'N') p[g.ix(a, b, n3 + 1)] = F.dr1 * F.bc1[q2] + S(ia, jb, n3);
  } else if (region == 1) {        // x ghost columns of the rows (b, k) (ghost rows included), in every plane the z condition does not set -- the ghost planes too
    // where z is left alone (the corrected normal velocity, skipped directions): the reference's x and y copies run over whole planes (bound.f90:158-199)
    const int bb = a, k = b;       // lanes along y (rows 4 KB apart share DRAM pages; along z they would be a plane apart), blocks along z
    if (!J.do_x || k > n3 + 1 || bb > n2 + 1 || (top_face && k == n3) || (k == 0 && zset0) || (k == n3 + 1 && zset1)) return;
    const int jb = wy(bb);
    p[g.ix(0, bb, k)] = S(n1, jb, k); p[g.ix(n1 + 1, bb, k)] = S(1, jb, k);
  } else {                         // y ghost rows, i = 1..n1, the same planes
    const int i = a + 1, k = b;
    if (!J.wrap_y || i > n1 || k > n3 + 1 || (top_face && k == n3) || (k == 0 && zset0) || (k == n3 + 1 && zset1)) return;
    p[g.ix(i, 0, k)] = S(i, n2, k); p[g.ix(i, n2 + 1, k)] = S(i, 1, k);
  }
}
static int launch_merged(cales_ctx *c, MJobs &J, const Geom *gg = nullptr) {
  if (!J.nf) return 0;
  const Geom &G = gg ? *gg : c->g;
  const int n[3] = {G.n1, G.n2, G.n3};
  // one grid for the three regions: region 0 needs (n1+2) x (n2+2), region 1 (n2+2) x n3, region 2 n1 x n3 threads
  const int ex = std::max(n[0] + 2, n[1] + 2), ey = std::max(n[1] + 2, n[2] + 2);
  LAUNCH(c, k_bc_merged, dim3((ex + 63) / 64, (ey + 3) / 4, 3 * J.nf), dim3(64, 4, 1), 0, c->stream, G, J);
  LAUNCHCHK(c);
  return 0;
}
// ---- ANY set of pointwise conditions (periodic, Dirichlet, Neumann on cell-centred data, Dirichlet on face-centred data) in all three directions in ONE
// launch: ducts, cavities, half channels -- the cases k_bc_merged (periodic x and y) leaves to one k_set_bc launch per direction. The reference applies
// x, then y, then z over WHOLE planes (bound.f90:158-199, set_bc: ghost rows and planes of the other directions included), so a later direction also
// rewrites the edge and corner cells of an earlier one from the values that one left. Every rule is "target = c bc + s source" with ONE source cell
// (set_bc, bound.f90:202-399), so that sequence has a closed form: the value of a cell the z rule sets is the z rule applied to the value the x and y
// rules leave in its source cell, and so on down to a cell no rule sets -- Z(Y(X(stored))) with each rule's boundary value taken where the reference
// takes it (the plane entry of the cell being set, ghost positions included). A thread computes its cell from stored cells no thread writes: no
// ordering between the directions is left. Not served: Neumann on face-centred data (outflow: set_bc copies the OLD boundary value into the ghost
// cell while it rewrites the boundary value itself) -- such sets keep the launch per direction.
struct ADir { char t0, t1, cen; real dr0, dr1; const real *bc0, *bc1; };      // t = 0: this end is left alone; 'P': both ends; cen = 0: face-centred along this direction
struct AField { real *p; ADir d[3]; char vcomp; };      // vcomp != 0: stored cells through the corrected view of this velocity component
struct AJobs { int nf; AField f[6]; CorrView V; };
__device__ inline bool a_set(const ADir &R, int idx, int n) {
  if (idx == 0) return R.t0 != 0;
  if (idx == n + 1) return R.t1 != 0;
  return idx == n && !R.cen && R.t1 == 'D';
}
// the rule of direction R for the cell at index idx (in 0, n, n+1): value = c * bc(side) + s * value(src)
__device__ inline void a_rule(const ADir &R, int idx, int n, real &c, real &s, int &src, int &side) {
  if (idx == 0) {
    side = 0; src = 1;
    if (R.t0 == 'P') { c = 0.; s = 1.; src = n; }
    else if (R.t0 == 'D') { if (R.cen) { c = 2.; s = -1.; } else { c = 1.; s = 0.; } }
    else { c = -R.dr0; s = 1.; }                                              // 'N', cell-centred: bound.f90:320-348
  } else {
    side = 1; src = n;
    if (R.t1 == 'P') { c = 0.; s = 1.; src = 1; }
    else if (R.t1 == 'D') { if (R.cen) { c = 2.; s = -1.; } else if (idx == n) { c = 1.; s = 0.; } else { c = 0.; s = 1.; src = n - 1; } }
    else { c = R.dr1; s = 1.; }
  }
}
__global__ __launch_bounds__(256) void k_bc_all(Geom g, AJobs J) {
  const int region = blockIdx.z % 3; const AField F = J.f[blockIdx.z / 3];
  const int n1 = g.n1, n2 = g.n2, n3 = g.n3;
  const int a = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y * 4 + threadIdx.y;
  real *p = F.p;
  auto S = [&](int i, int j, int k) -> real { return F.vcomp ? view_rd(g, J.V, F.vcomp, F.p, i, j, k) : F.p[g.ix(i, j, k)]; };      // stored cells
  auto VX = [&](int i, int j, int k) -> real {      // what the x rules leave at (i, j, k)
    const ADir &R = F.d[0];
    if (!a_set(R, i, n1)) return S(i, j, k);
    real c, s_; int src, side; a_rule(R, i, n1, c, s_, src, side);
    const real *bc = side ? R.bc1 : R.bc0;
    real v = (c != 0. && bc) ? c * bc[j + (size_t)(n2 + 2) * k] : 0.;
    if (s_ != 0.) v += s_ * S(src, j, k);
    return v;
  };
  auto VY = [&](int i, int j, int k) -> real {      // ... the x and y rules
    const ADir &R = F.d[1];
    if (!a_set(R, j, n2)) return VX(i, j, k);
    real c, s_; int src, side; a_rule(R, j, n2, c, s_, src, side);
    const real *bc = side ? R.bc1 : R.bc0;
    real v = (c != 0. && bc) ? c * bc[i + (size_t)(n1 + 2) * k] : 0.;
    if (s_ != 0.) v += s_ * VX(i, src, k);
    return v;
  };
  auto VZ = [&](int i, int j, int k) -> real {      // ... all three
    const ADir &R = F.d[2];
    if (!a_set(R, k, n3)) return VY(i, j, k);
    real c, s_; int src, side; a_rule(R, k, n3, c, s_, src, side);
    const real *bc = side ? R.bc1 : R.bc0;
    real v = (c != 0. && bc) ? c * bc[i + (size_t)(n1 + 2) * j] : 0.;
    if (s_ != 0.) v += s_ * VY(i, j, src);
    return v;
  };
  // the (up to three) indices a direction sets: 0, n+1 and, for face-centred Dirichlet data, n
  if (region == 0) {               // planes the z rules set, every (i, j) of the plane
    if (a > n1 + 1 || b > n2 + 1) return;
    const int ks[3] = {0, n3 + 1, n3};
#pragma unroll
    for (int q = 0; q < 3; ++q) if (a_set(F.d[2], ks[q], n3)) p[g.ix(a, b, ks[q])] = VZ(a, b, ks[q]);
  } else if (region == 1) {        // rows the y rules set, in the planes the z rules leave alone
    if (a > n1 + 1 || b > n3 + 1 || a_set(F.d[2], b, n3)) return;
    const int js[3] = {0, n2 + 1, n2};
#pragma unroll
    for (int q = 0; q < 3; ++q) if (a_set(F.d[1], js[q], n2)) p[g.ix(a, js[q], b)] = VY(a, js[q], b);
  } else {                         // columns the x rules set, in the rows and planes the others leave alone
    if (a > n2 + 1 || b > n3 + 1 || a_set(F.d[1], a, n2) || a_set(F.d[2], b, n3)) return;
    const int is_[3] = {0, n1 + 1, n1};
#pragma unroll
    for (int q = 0; q < 3; ++q) if (a_set(F.d[0], is_[q], n1)) p[g.ix(is_[q], a, b)] = VX(is_[q], a, b);
  }
}
static int launch_all(cales_ctx *c, AJobs &J) {
  if (!J.nf) return 0;
  const int *n = c->n;
  const int ex = std::max(n[0] + 2, n[1] + 2), ey = std::max(n[1] + 2, n[2] + 2);
  LAUNCH(c, k_bc_all, dim3((ex + 63) / 64, (ey + 3) / 4, 3 * J.nf), dim3(64, 4, 1), 0, c->stream, c->g, J);
  LAUNCHCHK(c);
  return 0;
}
// one direction of one field: the types of its two ends as the reference's loops over idir / ibound would apply them on this rank
static void a_dir(cales_ctx *c, ADir &D, int idir, char c0, char c1, int centered, const real *bc0, const real *bc1, real dr0, real dr1) {
  D.t0 = D.t1 = 0; D.cen = (char)centered; D.dr0 = dr0; D.dr1 = dr1; D.bc0 = bc0; D.bc1 = bc1;
  if ((bc_skipped(c) >> (idir - 1) & 1) && !(idir == 3 && c0 == 'P' && c1 == 'P')) return;      // (a periodic z is copied even where z is "skipped": the skip is for wall planes nobody reads)
  if (c0 == 'P' && c1 == 'P') {
    if (idir == 2 && c->P > 1) return;                // rows exchanged between the slabs (halo_y_comm, before the launch)
    D.t0 = D.t1 = 'P'; return;                        // (x and z are never decomposed: a local copy)
  }
  if (ISB(c, 0, idir)) D.t0 = c0;
  if (ISB(c, 1, idir)) D.t1 = c1;
}
static bool a_served(const AJobs &J) {      // no Neumann condition on face-centred data
  for (int q = 0; q < J.nf; ++q) for (int d = 0; d < 3; ++d) { const ADir &D = J.f[q].d[d]; if (!D.cen && (D.t0 == 'N' || D.t1 == 'N')) return false; }
  return true;
}
// x periodic, y periodic on one rank or exchanged between slabs, z pointwise
static bool merged_ok(const cales_ctx *c, const char *cbx, const char *cby) {
  if (c->fl.unmerged_bc) return false;
  if (!(cbx[0] == 'P' && cbx[1] == 'P')) return false;
  if (!(cby[0] == 'P' && cby[1] == 'P')) return false;
  return true;
}

// y-slab neighbours (bound.f90:619-696 for idir = 2): pack the first/last interior rows of nf fields into the
// staging buffer A, let the host exchange them, unpack into the ghost rows. Planes include the x/z ghosts.
struct HaloFields { int nf; real *p[16]; unsigned char wide[16]; int off[16]; unsigned char vcomp[16]; unsigned char rofs[16]; real *dst[16]; CorrView V; };      // rofs / dst: rows 1 + rofs and n2 - rofs leave, into the ghost rows of dst (a companion field; rofs = 0: the field itself)      // vcomp != 0: the rows that leave are read through the corrected view      // wide: a pair field (rows twice as long); off: first staging plane of the field, in planes of s1 (n3+2) values
// x ghost columns of the two z ghost planes, rows 0..n2+1: the corners the velocity update after the projection leaves alone (bounduvw with
// is_correc does not touch the z ghost planes of w, and the periodic copies of the step's earlier calls were skipped: cales_step, step_xskip)
__global__ __launch_bounds__(256) void k_xwrap_zghost(Geom g, HaloFields H) {
  const int j = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y ? g.n3 + 1 : 0;
  if (j > g.n2 + 1) return;
  real *p = H.p[blockIdx.z];
  p[g.ix(0, j, k)] = p[g.ix(g.n1, j, k)]; p[g.ix(g.n1 + 1, j, k)] = p[g.ix(1, j, k)];
}
__global__ __launch_bounds__(256) void k_pack_y(Geom g, HaloFields H, real *__restrict__ lo, real *__restrict__ hi) {
  const int i = blockIdx.x * 64 + threadIdx.x, k = blockIdx.y * 4 + threadIdx.y, f = blockIdx.z, w = H.wide[f];
  if (i >= (g.n1 + 2) << w || k > g.n3 + 1) return;
  const size_t s1 = (size_t)g.s1 << w, s12 = (size_t)g.s12 << w;
  const size_t q = (size_t)i + s1 * k + (size_t)g.s1 * (g.n3 + 2) * H.off[f];
  if (H.vcomp[f]) { lo[q] = view_rd(g, H.V, H.vcomp[f], H.p[f], i, 1, k); hi[q] = view_rd(g, H.V, H.vcomp[f], H.p[f], i, g.n2, k); return; }      // (never a pair field)
  lo[q] = H.p[f][i + s1 * (1 + H.rofs[f]) + s12 * k]; hi[q] = H.p[f][i + s1 * (g.n2 - H.rofs[f]) + s12 * k];
}
__global__ __launch_bounds__(256) void k_unpack_y(Geom g, HaloFields H, const real *__restrict__ lo, const real *__restrict__ hi, int has_lo, int has_hi) {
  const int i = blockIdx.x * 64 + threadIdx.x, k = blockIdx.y * 4 + threadIdx.y, f = blockIdx.z, w = H.wide[f];
  if (i >= (g.n1 + 2) << w || k > g.n3 + 1) return;
  const size_t s1 = (size_t)g.s1 << w, s12 = (size_t)g.s12 << w;
  const size_t q = (size_t)i + s1 * k + (size_t)g.s1 * (g.n3 + 2) * H.off[f];
  if (has_lo) H.dst[f][i + s12 * k] = lo[q];
  if (has_hi) H.dst[f][i + s1 * (g.n2 + 1) + s12 * k] = hi[q];
}
// A field of PAIRS (two values per cell, k_sgs.hip dsmag_pairs) is, for every operation that copies whole rows or planes, a field of twice the width:
// 2 (n1 + 2) values per row, pitches doubled. Its x ghost "columns" mean nothing in that view -- the callers skip direction x.
static Geom wide_geom(const cales_ctx *c) { Geom g = c->g; g.n1 = 2 * c->g.n1 + 2; g.s1 = 2 * c->g.s1; g.s12 = 2 * c->g.s12; return g; }
// (wide[q] != 0: field q is a pair field; nullptr: none is)
static int halo_y_on(cales_ctx *c, int nf, real **flds, hipStream_t st, bool overlapped, const unsigned char *wide = nullptr) {
  const Geom &G = c->g;
  HaloFields H; H.nf = nf; int planes = 0, anyw = 0;
  // (wide[q]: 0 a field, 1 a pair field, 2 / 3 the second / third rows of a field into its first / second companion)
  for (int q = 0; q < nf; ++q) {
    const int kind = wide ? wide[q] : 0;
    H.p[q] = flds[q]; H.wide[q] = kind == 1; H.off[q] = planes; planes += 1 + H.wide[q]; anyw |= H.wide[q]; H.vcomp[q] = 0;
    H.rofs[q] = kind >= 2 ? (unsigned char)(kind - 1) : 0; H.dst[q] = kind >= 2 ? flds[q] + (size_t)(kind - 1) * c->comp_one : flds[q];
  }
  if (c->bc_view_dtrk != 0.) {      // op_bounduvw through the corrected view: the velocity rows that leave are those of the projected velocity
    H.V = corr_view(c);
    for (int q = 0; q < nf; ++q) for (int iv = 0; iv < 3; ++iv) if (flds[q] == c->f[CALES_U + iv] && !H.rofs[q]) H.vcomp[q] = (unsigned char)(iv + 1);
  }
  const int64_t cnt = (int64_t)G.s1 * (c->n[2] + 2) * planes;
  if (4 * cnt > c->comm.nbuf) { c->err = "halo staging buffer too small"; return 1; }
  dim3 b(64, 4, 1), gr(((G.s1 << anyw) + 63) / 64, (c->n[2] + 2 + 3) / 4, nf);
  LAUNCH(c, k_pack_y, gr, b, 0, st, G, H, c->comm.A, c->comm.A + cnt);
  { ProfScope ps(c, "halo_exchange", st);
    const int rc = overlapped ? c->comm.halo_s(c->comm.user, 0, cnt, 0, cnt, cnt, (void *)st) : c->comm.halo(c->comm.user, 0, cnt, 0, cnt, cnt);
    if (rc) { c->err = "halo callback failed"; return 1; } }
  const int has_lo = (c->per_y || c->rank > 0) ? 1 : 0, has_hi = (c->per_y || c->rank < c->P - 1) ? 1 : 0;
  LAUNCH(c, k_unpack_y, gr, b, 0, st, G, H, c->comm.B, c->comm.B + cnt, has_lo, has_hi);
  LAUNCHCHK(c);
  return 0;
}
int halo_y_rows(cales_ctx *c, int nf, real **flds, int kind) {
  if (!c->comm.on) { c->err = "nranks > 1 but no communication hooks registered (cales_set_comm)"; return 1; }
  if (!c->comp_one || kind < 2 || kind > 3) { c->err = "halo_y_rows: no companion fields"; return 1; }
  if (c->defer_halo) { for (int q = 0; q < nf; ++q) { c->deferred.push_back(flds[q]); c->deferred_wide.push_back((unsigned char)kind); } return 0; }
  unsigned char w[16]; for (int q = 0; q < nf && q < 16; ++q) w[q] = (unsigned char)kind;
  return halo_y_on(c, nf, flds, c->stream, false, w);
}
static int halo_y_comm(cales_ctx *c, int nf, real **flds, bool wide = false) {
  if (!c->comm.on) { c->err = "nranks > 1 but no communication hooks registered (cales_set_comm)"; return 1; }
  if (c->bc_no_halo) return 0;      // the ghost rows already hold the neighbours' rows (end-of-step refresh of the x ghost columns, cales_step)
  if (c->defer_halo) { for (int q = 0; q < nf; ++q) { c->deferred.push_back(flds[q]); c->deferred_wide.push_back(wide ? 1 : 0); } return 0; }      // exchanged later (halo_flush_deferred)
  unsigned char w[16]; for (int q = 0; q < nf && q < 16; ++q) w[q] = wide ? 1 : 0;
  return halo_y_on(c, nf, flds, c->stream, false, w);
}
// The y-halo rows of the fields collected while c->defer_halo was set travel on the second stream, after everything queued on the
// context's stream so far (their ghost-cell kernels included: what those wrote into the ghost rows is overwritten by the rows
// that arrive, whose own x/z ghost cells the neighbour has already set -- the same values the in-order sequence produces).
// The caller makes the context's stream wait (stream_after) before the first kernel that reads those ghost rows.
// overlapped = false: the same batching in order on the context's stream -- one exchange for up to twelve fields instead of one per call.
int halo_flush_deferred(cales_ctx *c, bool overlapped) {
  if (c->deferred.empty()) return 0;
  if (overlapped) { if (int e = stream_after(c, c->comm_stream, c->stream)) return e; }
  for (size_t q0 = 0; q0 < c->deferred.size();) {      // as many fields per exchange as the staging buffers hold: sixteen planes, a pair field takes two
    int nf = 0, planes = 0;
    while (q0 + nf < c->deferred.size() && nf < 16 && planes + 1 + (c->deferred_wide[q0 + nf] == 1) <= 16) { planes += 1 + (c->deferred_wide[q0 + nf] == 1); ++nf; }
    if (int e = halo_y_on(c, nf, c->deferred.data() + q0, overlapped ? c->comm_stream : c->stream, overlapped, c->deferred_wide.data() + q0)) { c->deferred.clear(); c->deferred_wide.clear(); return e; }
    q0 += nf;
  }
  c->deferred_wide.clear();
  c->deferred.clear();
  return 0;
}
// halo exchange in the non-pencil directions: y across slabs (or a periodic copy on one rank), z always local
static int halo_self(cales_ctx *c, int nf, real **flds) {
  if (c->P > 1) { if (int e = halo_y_comm(c, nf, flds)) return e; }
  for (int idir = (c->P > 1 ? 3 : 2); idir <= 3; ++idir) {
    const bool periodic = idir == 2 ? c->per_y : !ISB(c, 0, 3);
    if (!periodic) continue;             // not periodic: neighbours are MPI_PROC_NULL
    BcJobs J; J.njobs = 0; J.idir = idir;
    const bool view = c->bc_view_dtrk != 0.;      // op_bounduvw through the corrected view: the periodic copies are those of the projected velocity
    if (view) J.V = corr_view(c);
    for (int q = 0; q < nf; ++q) {
      if (J.njobs == 6) { if (int e = launch_jobs(c, J)) return e; J.njobs = 0; }
      add_job(J, flds[q], 'P', 0, 1, nullptr, 0.);
      if (view) for (int iv = 0; iv < 3; ++iv) if (flds[q] == c->f[CALES_U + iv]) J.job[J.njobs - 1].vcomp = (char)(iv + 1);
    }
    if (int e = launch_jobs(c, J)) return e;
  }
  return 0;
}

// a cell-centred field with the pressure (which = 0) or the sgs (1) BC set as one entry of the one-launch kernel
static void merged_pfield(cales_ctx *c, MField &F, real *p, int which) {
  const char *cbc = which == 0 ? c->C.cbcpre : c->C.cbcsgs; const DBound &bc = which == 0 ? c->bcp : c->bcs;
  const bool per_z = cbc[4] == 'P' && cbc[5] == 'P';
  F.p = p; F.centered = 1;
  F.t0 = per_z ? 'P' : cbc[4]; F.t1 = per_z ? 'P' : cbc[5];
  F.bc0 = plane(bc, 3, 0, c->n); F.bc1 = plane(bc, 3, 1, c->n); F.dr0 = c->dzc[0]; F.dr1 = c->dzc[c->n[2]];
}
// a cell-centred field with the pressure (which = 0) or the sgs (1) BC set as one entry of the all-directions kernel
static void all_pfield(cales_ctx *c, AField &F, real *p, int which) {
  const char *cbc = which == 0 ? c->C.cbcpre : c->C.cbcsgs; const DBound &bc = which == 0 ? c->bcp : c->bcs;
  F.p = p; F.vcomp = 0;
  for (int idir = 1; idir <= 3; ++idir) {
    const real dr0 = idir < 3 ? c->dl[idir - 1] : c->dzc[0], dr1 = idir < 3 ? c->dl[idir - 1] : c->dzc[c->n[2]];
    a_dir(c, F.d[idir - 1], idir, cbc[2 * (idir - 1)], cbc[2 * (idir - 1) + 1], 1, plane(bc, idir, 0, c->n), plane(bc, idir, 1, c->n), dr0, dr1);
  }
}
// ------------------------------------------------------------------------------------------ boundp (bound.f90:156-200)
// pair fields (x and y periodic only: dsmag_pairs): the y rows (wrapped on one rank, exchanged between slabs) and the z ghost planes through the
// one-launch kernel in the doubled-width view; direction x is the consumers' business (they wrap around)
int op_boundp_wide(cales_ctx *c, int nf, real **p2, int which) {
  ProfScope ps(c, "boundp");
  const char *cbc = which == 0 ? c->C.cbcpre : c->C.cbcsgs;
  if (!merged_ok(c, cbc, cbc + 2) || nf > 8 || !(bc_skipped(c) & 1)) { c->err = "pair fields need periodic x and y and a caller that skips direction x"; return 1; }
  if (c->P > 1) { if (int e = halo_y_comm(c, nf, p2, true)) return e; }
  const bool per_z = cbc[4] == 'P' && cbc[5] == 'P';
  const Geom G = wide_geom(c);
  MJobs J; J.nf = nf; J.do_x = 0; J.wrap_y = c->P == 1; J.do_z = per_z || !(bc_skipped(c) & 4);
  if (!J.wrap_y && !J.do_z) return 0;
  for (int q = 0; q < nf; ++q) {
    merged_pfield(c, J.f[q], p2[q], which);
    if (!per_z) { J.f[q].t0 = 0; J.f[q].t1 = 0; }      // (pointwise z conditions would need the BC planes in the doubled view: the callers skip z at walls)
  }
  if (!per_z) J.do_z = 0;
  if (!J.wrap_y && !J.do_z) return 0;
  return launch_merged(c, J, &G);
}
// nf <= 8 fields with the same BC set in one halo exchange and as few launches as the job table allows
int op_boundp_multi(cales_ctx *c, int nf, real **p, int which) {
  ProfScope ps(c, "boundp");
  const char *cbc = which == 0 ? c->C.cbcpre : c->C.cbcsgs; const DBound &bc = which == 0 ? c->bcp : c->bcs;
  if (merged_ok(c, cbc, cbc + 2) && nf <= 8) {      // x, y periodic: all three directions in one launch (k_bc_merged)
    if (c->P > 1) { if (int e = halo_y_comm(c, nf, p)) return e; }
    const bool per_z = cbc[4] == 'P' && cbc[5] == 'P';
    MJobs J; J.nf = nf; J.do_x = !(bc_skipped(c) & 1); J.wrap_y = c->P == 1; J.do_z = per_z || !(bc_skipped(c) & 4);
    for (int q = 0; q < nf; ++q) merged_pfield(c, J.f[q], p[q], which);
    return launch_merged(c, J);
  }
  if (!c->fl.unmerged_bc) {      // every other set of a cell-centred field: all directions in one launch (k_bc_all), six fields at a time
    if (c->P > 1) { if (int e = halo_y_comm(c, nf, p)) return e; }
    for (int q0 = 0; q0 < nf; q0 += 6) {
      AJobs J; J.nf = std::min(6, nf - q0); J.V = CorrView{};
      for (int q = 0; q < J.nf; ++q) all_pfield(c, J.f[q], p[q0 + q], which);
      if (int e = launch_all(c, J)) return e;
    }
    return 0;
  }
  if (int e = halo_self(c, nf, p)) return e;
  for (int idir = 1; idir <= 3; ++idir) {
    if (!ISB(c, 0, idir) && !ISB(c, 1, idir)) continue;
    if (bc_skipped(c) >> (idir - 1) & 1) continue;
    BcJobs J; J.njobs = 0; J.idir = idir;
    const real dr0 = idir < 3 ? c->dl[idir - 1] : c->dzc[0], dr1 = idir < 3 ? c->dl[idir - 1] : c->dzc[c->n[2]];
    const char c0 = cbc[0 + 2 * (idir - 1)], c1 = cbc[1 + 2 * (idir - 1)];
    for (int q = 0; q < nf; ++q) {
      if (J.njobs + 2 > 6) { if (int e = launch_jobs(c, J)) return e; J.njobs = 0; }
      if (c0 == 'P') add_job(J, p[q], 'P', 0, 1, nullptr, 0.);      // both ends in one job (identical result to the two calls)
      else {
        if (ISB(c, 0, idir)) add_job(J, p[q], c0, 0, 1, plane(bc, idir, 0, c->n), dr0);
        if (ISB(c, 1, idir)) add_job(J, p[q], c1, 1, 1, plane(bc, idir, 1, c->n), dr1);
      }
    }
    if (J.njobs) { if (int e = launch_jobs(c, J)) return e; }
  }
  return 0;
}
int op_boundp(cales_ctx *c, real *p, int which) { real *fl[1] = {p}; return op_boundp_multi(c, 1, fl, which); }

// ------------------------------------------------------------------------------------------ wall model (wmodel.f90:65-335)
__device__ inline real vel_relative(real v1, real v2, real coef, real mag) {
  real r = (1. - coef) * v1 + coef * v2;
  return r - mag;
}
__device__ inline void wallmodel(int mtype, real uh, real vh, real h, real l1d, real visc, real &t1, real &t2) {
  const real kap = 0.41, blog = 5.20;
  real upar = sqrt(uh * uh + vh * vh), tauw_tot;
  if (mtype == 1) {
    real conv = 1., utau = fmax(sqrt(upar / h * visc), visc / h * exp(-kap * blog));
    while (conv > 0.5e-4) {
      const real utau_old = utau;
      const real f = upar / utau - 1. / kap * log(h * utau / visc) - blog;
      const real fp = -1. / utau * (upar / utau + 1. / kap);
      utau = fabs(utau - f / fp);
      conv = fabs(utau / utau_old - 1.);
    }
    tauw_tot = utau * utau;
  } else {
    const real del = 0.5 * l1d, umax = upar / (h / del * (2. - h / del));
    tauw_tot = 2. / del * umax * visc;
  }
  t1 = tauw_tot * uh / (upar + CALES_EPS); t2 = tauw_tot * vh / (upar + CALES_EPS);
}

struct WmArgs {
  int idir, ibound, mtype, i1, i2;   // i1/i2: near / far interpolation index along idir
  real coef, sgn, h, l1d, visc;
  const real *u, *v, *w;           // velocity fields
  real *bc_a, *bc_b;               // planes (side ibound) receiving the first / second tangential component
  const real *mag_a, *mag_b;       // *_mag planes (side ibound)
  const real *zc, *zf, *dzc;
};
// All wall-model faces of a bounduvw in one launch (a channel has two, a duct four): blockIdx.z = 2 face + component; the grid spans the
// largest face and the blocks beyond a smaller one return.
// component 0: first tangential component loop, 1: second (wmodel.f90:138-153/154-170, 189-204/205-221, 240-255/256-271)
struct WmJobs { WmArgs a[6]; int n; };
__global__ __launch_bounds__(256) void k_wallmodel(Geom g, WmJobs J) {
  const WmArgs &A = J.a[blockIdx.z >> 1];
  const int na = A.idir == 1 ? g.n2 : g.n1, nb = A.idir == 3 ? g.n2 : g.n3;
  const int a = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y * 4 + threadIdx.y, comp = blockIdx.z & 1;
  if (a > na + 1 || b > nb + 1) return;
  const size_t ld = na + 2;
  const real visci = 1. / A.visc;
  real t1, t2;
#define M(pl, a_, b_) pl[(a_) + ld * (b_)]
  if (A.idir == 1) {          // wall normal x; a = j, b = k; tangential: v (first), w (second)
    const int i1 = A.i1, i2 = A.i2;
    if (comp == 0) {
      if (a > na || b < 1 || b > nb) return;           // j = 0..n2, k = 1..n3
      const int j = a, k = b;
      const real v1 = A.v[g.ix(i1, j, k)], v2 = A.v[g.ix(i2, j, k)];
      const real w1 = 0.25 * (A.w[g.ix(i1, j, k)] + A.w[g.ix(i1, j + 1, k)] + A.w[g.ix(i1, j, k - 1)] + A.w[g.ix(i1, j + 1, k - 1)]);
      const real w2 = 0.25 * (A.w[g.ix(i2, j, k)] + A.w[g.ix(i2, j + 1, k)] + A.w[g.ix(i2, j, k - 1)] + A.w[g.ix(i2, j + 1, k - 1)]);
      const real v_mag = M(A.mag_a, j, k), w_mag = 0.25 * (M(A.mag_b, j, k) + M(A.mag_b, j + 1, k) + M(A.mag_b, j, k - 1) + M(A.mag_b, j + 1, k - 1));
      wallmodel(A.mtype, vel_relative(v1, v2, A.coef, v_mag), vel_relative(w1, w2, A.coef, w_mag), A.h, A.l1d, A.visc, t1, t2);
      M(A.bc_a, j, k) = A.sgn * visci * t1;
    } else {
      if (a < 1 || a > na || b > nb) return;           // j = 1..n2, k = 0..n3
      const int j = a, k = b;
      const real wei = (A.zf[k] - A.zc[k]) / A.dzc[k];
      const real v1 = 0.5 * ((1. - wei) * (A.v[g.ix(i1, j - 1, k)] + A.v[g.ix(i1, j, k)]) + wei * (A.v[g.ix(i1, j - 1, k + 1)] + A.v[g.ix(i1, j, k + 1)]));
      const real v2 = 0.5 * ((1. - wei) * (A.v[g.ix(i2, j - 1, k)] + A.v[g.ix(i2, j, k)]) + wei * (A.v[g.ix(i2, j - 1, k + 1)] + A.v[g.ix(i2, j, k + 1)]));
      const real w1 = A.w[g.ix(i1, j, k)], w2 = A.w[g.ix(i2, j, k)];
      const real v_mag = 0.5 * ((1. - wei) * (M(A.mag_a, j - 1, k) + M(A.mag_a, j, k)) + wei * (M(A.mag_a, j - 1, k + 1) + M(A.mag_a, j, k + 1)));
      const real w_mag = M(A.mag_b, j, k);
      wallmodel(A.mtype, vel_relative(v1, v2, A.coef, v_mag), vel_relative(w1, w2, A.coef, w_mag), A.h, A.l1d, A.visc, t1, t2);
      M(A.bc_b, j, k) = A.sgn * visci * t2;
    }
  } else if (A.idir == 2) {   // wall normal y; a = i, b = k; tangential: u (first), w (second)
    const int j1 = A.i1, j2 = A.i2;
    if (comp == 0) {
      if (a > na || b < 1 || b > nb) return;           // i = 0..n1, k = 1..n3
      const int i = a, k = b;
      const real u1 = A.u[g.ix(i, j1, k)], u2 = A.u[g.ix(i, j2, k)];
      const real w1 = 0.25 * (A.w[g.ix(i, j1, k)] + A.w[g.ix(i + 1, j1, k)] + A.w[g.ix(i, j1, k - 1)] + A.w[g.ix(i + 1, j1, k - 1)]);
      const real w2 = 0.25 * (A.w[g.ix(i, j2, k)] + A.w[g.ix(i + 1, j2, k)] + A.w[g.ix(i, j2, k - 1)] + A.w[g.ix(i + 1, j2, k - 1)]);
      const real u_mag = M(A.mag_a, i, k), w_mag = 0.25 * (M(A.mag_b, i, k) + M(A.mag_b, i + 1, k) + M(A.mag_b, i, k - 1) + M(A.mag_b, i + 1, k - 1));
      wallmodel(A.mtype, vel_relative(u1, u2, A.coef, u_mag), vel_relative(w1, w2, A.coef, w_mag), A.h, A.l1d, A.visc, t1, t2);
      M(A.bc_a, i, k) = A.sgn * visci * t1;
    } else {
      if (a < 1 || a > na || b > nb) return;           // i = 1..n1, k = 0..n3
      const int i = a, k = b;
      const real wei = (A.zf[k] - A.zc[k]) / A.dzc[k];
      const real u1 = 0.5 * ((1. - wei) * (A.u[g.ix(i - 1, j1, k)] + A.u[g.ix(i, j1, k)]) + wei * (A.u[g.ix(i - 1, j1, k + 1)] + A.u[g.ix(i, j1, k + 1)]));
      const real u2 = 0.5 * ((1. - wei) * (A.u[g.ix(i - 1, j2, k)] + A.u[g.ix(i, j2, k)]) + wei * (A.u[g.ix(i - 1, j2, k + 1)] + A.u[g.ix(i, j2, k + 1)]));
      const real w1 = A.w[g.ix(i, j1, k)], w2 = A.w[g.ix(i, j2, k)];
      const real u_mag = 0.5 * ((1. - wei) * (M(A.mag_a, i - 1, k) + M(A.mag_a, i, k)) + wei * (M(A.mag_a, i - 1, k + 1) + M(A.mag_a, i, k + 1)));
      const real w_mag = M(A.mag_b, i, k);
      wallmodel(A.mtype, vel_relative(u1, u2, A.coef, u_mag), vel_relative(w1, w2, A.coef, w_mag), A.h, A.l1d, A.visc, t1, t2);
      M(A.bc_b, i, k) = A.sgn * visci * t2;
    }
  } else {                    // wall normal z; a = i, b = j; tangential: u (first), v (second)
    const int k1 = A.i1, k2 = A.i2;
    if (comp == 0) {
      if (a > na || b < 1 || b > nb) return;           // i = 0..n1, j = 1..n2
      const int i = a, j = b;
      const real u1 = A.u[g.ix(i, j, k1)], u2 = A.u[g.ix(i, j, k2)];
      const real v1 = 0.25 * (A.v[g.ix(i, j, k1)] + A.v[g.ix(i + 1, j, k1)] + A.v[g.ix(i, j - 1, k1)] + A.v[g.ix(i + 1, j - 1, k1)]);
      const real v2 = 0.25 * (A.v[g.ix(i, j, k2)] + A.v[g.ix(i + 1, j, k2)] + A.v[g.ix(i, j - 1, k2)] + A.v[g.ix(i + 1, j - 1, k2)]);
      const real u_mag = M(A.mag_a, i, j), v_mag = 0.25 * (M(A.mag_b, i, j) + M(A.mag_b, i + 1, j) + M(A.mag_b, i, j - 1) + M(A.mag_b, i + 1, j - 1));
      wallmodel(A.mtype, vel_relative(u1, u2, A.coef, u_mag), vel_relative(v1, v2, A.coef, v_mag), A.h, A.l1d, A.visc, t1, t2);
      M(A.bc_a, i, j) = A.sgn * visci * t1;
    } else {
      if (a < 1 || a > na || b > nb) return;           // i = 1..n1, j = 0..n2
      const int i = a, j = b;
      const real u1 = 0.25 * (A.u[g.ix(i - 1, j, k1)] + A.u[g.ix(i, j, k1)] + A.u[g.ix(i - 1, j + 1, k1)] + A.u[g.ix(i, j + 1, k1)]);
      const real u2 = 0.25 * (A.u[g.ix(i - 1, j, k2)] + A.u[g.ix(i, j, k2)] + A.u[g.ix(i - 1, j + 1, k2)] + A.u[g.ix(i, j + 1, k2)]);
      const real v1 = A.v[g.ix(i, j, k1)], v2 = A.v[g.ix(i, j, k2)];
      const real u_mag = 0.25 * (M(A.mag_a, i - 1, j) + M(A.mag_a, i, j) + M(A.mag_a, i - 1, j + 1) + M(A.mag_a, i, j + 1));
      const real v_mag = M(A.mag_b, i, j);
      wallmodel(A.mtype, vel_relative(u1, u2, A.coef, u_mag), vel_relative(v1, v2, A.coef, v_mag), A.h, A.l1d, A.visc, t1, t2);
      M(A.bc_b, i, j) = A.sgn * visci * t2;
    }
  }
#undef M
}

static int updt_wallmodelbc(cales_ctx *c, DBound &bu, DBound &bv, DBound &bw, const real *u, const real *v, const real *w) {
  const int *n = c->n; const real h = c->C.hwm; const real *dl = c->dl;
  WmJobs J; J.n = 0; int gx = 0, gy = 0;
  for (int idir = 1; idir <= 3; ++idir) for (int ib = 0; ib <= 1; ++ib) {
    if (!(ISB(c, ib, idir) && LWM(c, ib, idir) != 0)) continue;
    WmArgs &A = J.a[J.n++]; A.idir = idir; A.ibound = ib; A.mtype = LWM(c, ib, idir); A.h = h; A.visc = c->visc; A.l1d = c->C.l[idir - 1];
    A.u = u; A.v = v; A.w = w; A.zc = c->d_zc; A.zf = c->d_zf; A.dzc = c->d_dzc;
    const int index = IWM(c, ib, idir);
    A.i2 = index; A.i1 = ib == 0 ? index - 1 : index + 1; A.sgn = ib == 0 ? 1. : -1.;
    if (idir == 1) { A.coef = ib == 0 ? (h - (A.i1 - 0.5) * dl[0]) / dl[0] : (h - (n[0] - A.i1 + 0.5) * dl[0]) / dl[0]; }
    else if (idir == 2) { A.coef = ib == 0 ? (h - (A.i1 - 0.5) * dl[1]) / dl[1] : (h - (n[1] - A.i1 + 0.5) * dl[1]) / dl[1]; }
    else { A.coef = ib == 0 ? (h - c->zc[A.i1]) / c->dzc[A.i1] : (h - (c->C.l[2] - c->zc[A.i1])) / (c->dzc[A.i2]); }
    DBound *ba = idir == 1 ? &bv : &bu, *bb = idir == 3 ? &bv : &bw;
    const DBound *ma = idir == 1 ? &c->bcv_mag : &c->bcu_mag, *mb = idir == 3 ? &c->bcv_mag : &c->bcw_mag;
    A.bc_a = const_cast<real *>(plane(*ba, idir, ib, n)); A.bc_b = const_cast<real *>(plane(*bb, idir, ib, n));
    A.mag_a = plane(*ma, idir, ib, n); A.mag_b = plane(*mb, idir, ib, n);
    const int na = idir == 1 ? n[1] : n[0], nb = idir == 3 ? n[1] : n[2];
    gx = std::max(gx, (na + 2 + 63) / 64); gy = std::max(gy, (nb + 2 + 3) / 4);
  }
  if (J.n) LAUNCH(c, k_wallmodel, dim3(gx, gy, 2 * J.n), dim3(64, 4, 1), 0, c->stream, c->g, J);
  LAUNCHCHK(c);
  return 0;
}

// ------------------------------------------------------------------------------------------ bounduvw (bound.f90:18-154)
int op_bounduvw(cales_ctx *c, DBound &bu, DBound &bv, DBound &bw, int is_updt_wm, int is_correc, real *u, real *v, real *w) {
  ProfScope ps(c, "bounduvw");
  const int *n = c->n;
  real *fl[3] = {u, v, w};
  DBound *bnd[3] = {&bu, &bv, &bw};
  // corrected view (cales_step, fold_mom): the fields hold the prediction, the ghost cells receive the values of the projected velocity
  const bool view = c->bc_view_dtrk != 0.;
  const CorrView V = view ? corr_view(c) : CorrView{};
  bool merged = merged_ok(c, c->C.cbcpre, c->C.cbcpre + 2);       // velocity and pressure are periodic together (sanity.f90:163-175)
  for (int ivel = 1; ivel <= 3 && merged; ++ivel) {
    for (int d = 1; d <= 2; ++d) merged = merged && CBV(c, 0, d, ivel) == 'P' && CBV(c, 1, d, ivel) == 'P';
    if (ivel == 3 && (CBV(c, 0, 3, 3) == 'N' || CBV(c, 1, 3, 3) == 'N')) merged = false;      // face-centred Neumann reads the plane it rewrites
  }
  // the all-directions kernel for the sets the periodic one does not serve: types of every (component, direction, end) as the loops below would apply them
  AJobs JA; JA.nf = 0; JA.V = V; bool allv = false; int nra = 0; real *alla[6] = {fl[0], fl[1], fl[2], nullptr, nullptr, nullptr};
  if (!merged && !c->fl.unmerged_bc) {
    JA.nf = 3;
    for (int ivel = 1; ivel <= 3; ++ivel) {
      AField &F = JA.f[ivel - 1]; F.p = fl[ivel - 1]; F.vcomp = view ? (char)ivel : 0;
      for (int idir = 1; idir <= 3; ++idir) {
        const bool periodic = CBV(c, 0, idir, idir) == 'P' && CBV(c, 1, idir, idir) == 'P', normal = ivel == idir;
        const real dr0 = idir < 3 ? c->dl[idir - 1] : (normal ? c->dzf[0] : c->dzc[0]), dr1 = idir < 3 ? c->dl[idir - 1] : (normal ? c->dzf[n[2]] : c->dzc[n[2]]);
        char c0 = CBV(c, 0, idir, ivel), c1 = CBV(c, 1, idir, ivel);
        if (normal && is_correc && !periodic) c0 = c1 = 0;      // the corrected normal velocity keeps its wall value (bound.f90:60-75)
        a_dir(c, F.d[idir - 1], idir, c0, c1, normal ? 0 : 1, plane(*bnd[ivel - 1], idir, 0, n), plane(*bnd[ivel - 1], idir, 1, n), dr0, dr1);
        if (!c0 && !c1) F.d[idir - 1].t0 = F.d[idir - 1].t1 = 0;
        if (!normal) { if (LWM(c, 0, idir) != 0) F.d[idir - 1].t0 = 0; if (LWM(c, 1, idir) != 0) F.d[idir - 1].t1 = 0; }      // set below from the wall-model stress
      }
    }
    allv = a_served(JA);
    if (allv && c->bc_nride > 0 && c->bc_nride <= 3 && !(bc_skipped(c) & 4)) {      // riders (cales_step): cell-centred fields join this launch -- and this slab exchange
      nra = c->bc_nride;
      for (int q = 0; q < nra; ++q) { all_pfield(c, JA.f[3 + q], c->bc_ride[q], c->bc_ride_which[q]); alla[3 + q] = c->bc_ride[q]; }
      JA.nf = 3 + nra;
    }
  }
  if (merged) {
    // riders (cales_step): cell-centred fields whose BC sets take the one-launch kernel too join this launch -- and this slab exchange
    int nr = 0;
    if (c->bc_nride > 0 && !(bc_skipped(c) & 4)) {
      bool ok = true;
      for (int q = 0; q < c->bc_nride; ++q) { const char *cb = c->bc_ride_which[q] == 0 ? c->C.cbcpre : c->C.cbcsgs; ok = ok && merged_ok(c, cb, cb + 2); }
      if (ok) nr = c->bc_nride;
    }
    real *all[8] = {fl[0], fl[1], fl[2]};
    for (int q = 0; q < nr; ++q) all[3 + q] = c->bc_ride[q];
    if (c->P > 1) { if (int e = halo_y_comm(c, 3 + nr, all)) return e; }
    const bool per_z = CBV(c, 0, 3, 3) == 'P' && CBV(c, 1, 3, 3) == 'P';
    MJobs J; J.nf = 3 + nr; J.do_x = !(bc_skipped(c) & 1); J.wrap_y = c->P == 1; J.do_z = per_z || !(bc_skipped(c) & 4);
    for (int q = 0; q < nr; ++q) merged_pfield(c, J.f[3 + q], c->bc_ride[q], c->bc_ride_which[q]);
    if (nr) c->bc_nride = 0;      // taken
    J.V = V;
    for (int ivel = 1; ivel <= 3; ++ivel) {
      MField &F = J.f[ivel - 1]; F.p = fl[ivel - 1]; F.vcomp = view ? (char)ivel : 0;
      const bool normal = ivel == 3;
      F.centered = normal ? 0 : 1;
      F.bc0 = plane(*bnd[ivel - 1], 3, 0, n); F.bc1 = plane(*bnd[ivel - 1], 3, 1, n);
      F.dr0 = normal ? c->dzf[0] : c->dzc[0]; F.dr1 = normal ? c->dzf[n[2]] : c->dzc[n[2]];
      if (per_z) { F.t0 = F.t1 = 'P'; continue; }
      F.t0 = CBV(c, 0, 3, ivel); F.t1 = CBV(c, 1, 3, ivel);
      if (normal && is_correc) F.t0 = F.t1 = 0;                                 // the corrected normal velocity keeps its wall value (bound.f90:60-75)
      if (!normal) { if (LWM(c, 0, 3) != 0) F.t0 = 0; if (LWM(c, 1, 3) != 0) F.t1 = 0; }      // set below from the wall-model stress
    }
    if (int e = launch_merged(c, J)) return e;
  } else if (allv) {
    // every other pointwise set: the three directions of the three components (and of the riders) in ONE launch (k_bc_all)
    if (c->P > 1) { if (int e = halo_y_comm(c, 3 + nra, alla)) return e; }
    if (nra) c->bc_nride = 0;      // taken
    if (int e = launch_all(c, JA)) return e;
  } else {
  if (int e = halo_self(c, 3, fl)) return e;
  for (int idir = 1; idir <= 3; ++idir) {
    if (!ISB(c, 0, idir) && !ISB(c, 1, idir)) continue;
    if (bc_skipped(c) >> (idir - 1) & 1) continue;
    BcJobs J; J.njobs = 0; J.idir = idir;
    const bool periodic = CBV(c, 0, idir, idir) == 'P' && CBV(c, 1, idir, idir) == 'P';
    const bool impose_norm = (!is_correc) || periodic;
    const real drn0 = idir < 3 ? c->dl[idir - 1] : c->dzf[0], drn1 = idir < 3 ? c->dl[idir - 1] : c->dzf[n[2]];
    const real drt0 = idir < 3 ? c->dl[idir - 1] : c->dzc[0], drt1 = idir < 3 ? c->dl[idir - 1] : c->dzc[n[2]];
    J.V = V;
    for (int ivel = 1; ivel <= 3; ++ivel) {
      real *p = fl[ivel - 1];
      const bool normal = ivel == idir;
      const char c0 = CBV(c, 0, idir, ivel), c1 = CBV(c, 1, idir, ivel);
      const int first = J.njobs;
      struct SetView { BcJobs &J; int first, comp; ~SetView() { for (int q = first; q < J.njobs; ++q) J.job[q].vcomp = (char)comp; } } setview{J, first, view ? ivel : 0};
      if (normal) {
        if (!impose_norm) continue;
        if (c0 == 'P') add_job(J, p, 'P', 0, 0, nullptr, 0.);
        else {
          if (ISB(c, 0, idir)) add_job(J, p, c0, 0, 0, plane(*bnd[ivel - 1], idir, 0, n), drn0);
          if (ISB(c, 1, idir)) add_job(J, p, c1, 1, 0, plane(*bnd[ivel - 1], idir, 1, n), drn1);
        }
      } else {
        if (c0 == 'P' && LWM(c, 0, idir) == 0) { add_job(J, p, 'P', 0, 1, nullptr, 0.); continue; }
        if (ISB(c, 0, idir) && LWM(c, 0, idir) == 0) add_job(J, p, c0, 0, 1, plane(*bnd[ivel - 1], idir, 0, n), drt0);
        if (ISB(c, 1, idir) && LWM(c, 1, idir) == 0) add_job(J, p, c1, 1, 1, plane(*bnd[ivel - 1], idir, 1, n), drt1);
      }
    }
    if (int e = launch_jobs(c, J)) return e;
  }
  }
  // cales_step, first bounduvw of a substep (bc_skip_wm): the wall-model planes and the tangential ghost cells they set are rewritten by the bounduvw that
  // follows the correction before anything reads them (fillps, the solver and correc do not) -- see step_body
  if (c->bc_skip_wm) return 0;
  if (is_updt_wm) if (int e = updt_wallmodelbc(c, bu, bv, bw, u, v, w)) return e;
  if (!c->fl.unmerged_bc) {      // tangential Neumann BCs carrying the wall-model stress (bound.f90:125-148): every wall-model face in one launch
    AJobs JW; JW.nf = 3; JW.V = V; bool any = false;
    for (int ivel = 1; ivel <= 3; ++ivel) {
      AField &F = JW.f[ivel - 1]; F.p = fl[ivel - 1]; F.vcomp = view ? (char)ivel : 0;
      for (int idir = 1; idir <= 3; ++idir) {
        ADir &D = F.d[idir - 1]; D.t0 = D.t1 = 0; D.cen = 1; D.dr0 = idir < 3 ? c->dl[idir - 1] : c->dzc[0]; D.dr1 = idir < 3 ? c->dl[idir - 1] : c->dzc[n[2]];
        D.bc0 = plane(*bnd[ivel - 1], idir, 0, n); D.bc1 = plane(*bnd[ivel - 1], idir, 1, n);
        if (ivel == idir) continue;
        if (ISB(c, 0, idir) && LWM(c, 0, idir) != 0) { D.t0 = CBV(c, 0, idir, ivel); any = true; }
        if (ISB(c, 1, idir) && LWM(c, 1, idir) != 0) { D.t1 = CBV(c, 1, idir, ivel); any = true; }
      }
    }
    if (!any) return 0;
    if (a_served(JW)) return launch_all(c, JW);
  }
  for (int idir = 1; idir <= 3; ++idir) {   // ... or one launch per direction
    BcJobs J; J.njobs = 0; J.idir = idir;
    const real drt0 = idir < 3 ? c->dl[idir - 1] : c->dzc[0], drt1 = idir < 3 ? c->dl[idir - 1] : c->dzc[n[2]];
    for (int ib = 0; ib <= 1; ++ib) {
      if (!(ISB(c, ib, idir) && LWM(c, ib, idir) != 0)) continue;
      for (int ivel = 1; ivel <= 3; ++ivel) if (ivel != idir)
        add_job(J, fl[ivel - 1], CBV(c, ib, idir, ivel), ib, 1, plane(*bnd[ivel - 1], idir, ib, n), ib ? drt1 : drt0);
    }
    if (int e = launch_jobs(c, J)) return e;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------ updt_rhs_b (bound.f90:562-617)
struct RhsJob { real *p; const real *rhs; int idir, pos, na, nb; };
__global__ __launch_bounds__(256) void k_updt_rhs_b(Geom g, RhsJob J) {
  const int a = blockIdx.x * 64 + threadIdx.x + 1, b = blockIdx.y * 4 + threadIdx.y + 1;
  if (a > J.na || b > J.nb) return;
  const size_t q = J.idir == 1 ? g.ix(J.pos, a, b) : J.idir == 2 ? g.ix(a, J.pos, b) : g.ix(a, b, J.pos);
  J.p[q] += J.rhs[(a - 1) + (size_t)J.na * (b - 1)];
}
static int rhs_b_dir(cales_ctx *c, real *p, int idir, const char *cbc6, const char *cf, const real *rhs, real scale_unused) {
  const int *n = c->n;
  const int na = idir == 1 ? n[1] : n[0], nb = idir == 3 ? n[1] : n[2];
  const int q = (cf[idir - 1] == 'f' && cbc6[1 + 2 * (idir - 1)] == 'D') ? 1 : 0;
  for (int ib = 0; ib <= 1; ++ib) {
    if (!ISB(c, ib, idir)) continue;
    RhsJob J; J.p = p; J.rhs = rhs + (size_t)ib * na * nb; J.idir = idir; J.pos = ib ? n[idir - 1] - q : 1; J.na = na; J.nb = nb;
    LAUNCH(c, k_updt_rhs_b, dim3((na + 63) / 64, (nb + 3) / 4), dim3(64, 4), 0, c->stream, c->g, J);
  }
  LAUNCHCHK(c);
  return 0;
}
static bool plane_all_zero(const cales_ctx *c, int idir) { (void)c; (void)idir; return false; }

int op_updt_rhs_b(cales_ctx *c) {
  ProfScope ps(c, "updt_rhs_b");
  // the planes are zero for periodic and homogeneous BCs; skip those launches (adding 0. is the identity)
  for (int idir = 1; idir <= 3; ++idir) {
    const bool zero = (c->C.bcpre[0 + 2 * (idir - 1)] == 0. && c->C.bcpre[1 + 2 * (idir - 1)] == 0.) ||
                      (CBP(c, 0, idir) == 'P');
    if (zero || plane_all_zero(c, idir)) continue;
    if (int e = rhs_b_dir(c, c->f[CALES_PP], idir, c->C.cbcpre, "ccc", c->rhsbp[idir - 1], 1.)) return e;
  }
  return 0;
}

// z part of the Helmholtz boundary r.h.s. for one velocity component (main.f90:425-433): rhsbz computed on the
// host from the CURRENT bc planes would need a download; the planes of wall-model faces change every substep,
// so the term is evaluated on the device from the plane directly.
__global__ __launch_bounds__(256) void k_rhs_b_velz(Geom g, real *p, const real *bcplane, int ib, char ctype, char c_or_f, real dlc,
                                                    real dlf, real alpha, int pos, real *plane_out) {
  const int i = blockIdx.x * 64 + threadIdx.x + 1, j = blockIdx.y * 4 + threadIdx.y + 1;
  if (i > g.n1 || j > g.n2) return;
  const RhsBz R{bcplane, ctype, c_or_f, dlc, dlf, ib == 0 ? (real)1. : (real)-1., alpha};
  const real r = R.at(i, j, g.n1);
  if (plane_out) plane_out[(size_t)(i - 1) + (size_t)g.n1 * (j - 1)] = r;      // added by the column form of the fused Helmholtz sweep, in the reference's order
  else p[g.ix(i, j, pos)] += r;
}
// The same term on the x (idir = 1) and y (idir = 2) faces for the 3-D implicit step (main.f90:424-431: rhsbx, rhsby times alpha, added by
// updt_rhs_b to the first and the last unknown plane of the direction, bound.f90:578-603): inflow profiles, moving side walls.
__global__ __launch_bounds__(256) void k_rhs_b_velxy(Geom g, real *p, const real *bcplane, int idir, int ib, char ctype, char c_or_f, real dl,
                                                     real alpha, int pos) {
  const int a = blockIdx.x * 64 + threadIdx.x + 1, k = blockIdx.y * 4 + threadIdx.y + 1;      // a = j (x faces) or i (y faces)
  const int na = idir == 1 ? g.n2 : g.n1;
  if (a > na || k > g.n3) return;
  const real bcv = bcplane[a + (size_t)(na + 2) * k];
  const real sgn = ib == 0 ? 1. : -1.;
  real r = 0.;
  if (c_or_f == 'c') { if (ctype == 'D') r = -2. * bcv / dl / dl; else if (ctype == 'N') r = sgn * bcv / dl; }
  else               { if (ctype == 'D') r = -bcv / dl / dl;      else if (ctype == 'N') r = sgn * bcv / dl; }
  p[idir == 1 ? g.ix(pos, a, k) : g.ix(a, pos, k)] += r * alpha;
}
int op_rhs_b_velxy(cales_ctx *c, int ivel, real alpha) {
  const int *n = c->n;
  const DBound &bc = ivel == 1 ? c->bcu : ivel == 2 ? c->bcv : c->bcw;
  for (int idir = 1; idir <= 2; ++idir) {
    const char cf = ivel == idir ? 'f' : 'c';
    const char *cbc = &c->cbcvel[6 * (ivel - 1) + 2 * (idir - 1)];
    const int q = (cf == 'f' && cbc[1] == 'D') ? 1 : 0;
    const int na = idir == 1 ? n[1] : n[0];
    for (int ib = 0; ib <= 1; ++ib) {
      if (!ISB(c, ib, idir) || cbc[ib] == 'P') continue;
      LAUNCH(c, k_rhs_b_velxy, dim3((na + 63) / 64, (n[2] + 3) / 4), dim3(64, 4), 0, c->stream, c->g, c->f[CALES_U + ivel - 1],
                         plane(bc, idir, ib, n), idir, ib, cbc[ib], cf, c->dl[idir - 1], alpha, ib ? n[idir - 1] - q : 1);
    }
  }
  LAUNCHCHK(c);
  return 0;
}
// planes != nullptr: the two contributions go to planes[0 / n1*n2] instead of being added to the field; has[ib] tells which exist
// the two sides as arguments of the in-LDS sweep (no launch): has[ib] as below
void rhs_b_velz_args(cales_ctx *c, int ivel, real alpha, RhsBz *R, int *has) {
  const int *n = c->n; const int n3 = n[2];
  const char cf = ivel == 3 ? 'f' : 'c';
  const DBound &bc = ivel == 1 ? c->bcu : ivel == 2 ? c->bcv : c->bcw;
  const char *cbc = &c->cbcvel[6 * (ivel - 1) + 4];
  has[0] = has[1] = 0;
  for (int ib = 0; ib <= 1; ++ib) {
    R[ib] = RhsBz{nullptr, 0, cf, 1., 1., 1., 0.};
    if (!ISB(c, ib, 3) || cbc[ib] == 'P') continue;
    has[ib] = 1;
    const real dlc = cf == 'c' ? (ib ? c->dzc[n3] : c->dzc[0]) : (ib ? c->dzc[n3 - 1] : c->dzc[1]);
    const real dlf = ib ? c->dzf[n3] : c->dzf[1];
    R[ib] = RhsBz{plane(bc, 3, ib, n), cbc[ib], cf, dlc, dlf, ib == 0 ? (real)1. : (real)-1., alpha};
  }
}
int op_rhs_b_velz(cales_ctx *c, int ivel, real alpha, real *planes, int *has) {
  const int *n = c->n; const int n3 = n[2];
  const char cf = ivel == 3 ? 'f' : 'c';
  const DBound &bc = ivel == 1 ? c->bcu : ivel == 2 ? c->bcv : c->bcw;
  const char *cbc = &c->cbcvel[6 * (ivel - 1) + 4];
  const int q = (cf == 'f' && cbc[1] == 'D') ? 1 : 0;
  // bound.f90:479-482: dzc01_c=[dzc(0),dzc(n)], dzf01_c=[dzf(1),dzf(n)]; dzc01_f=[dzc(1),dzc(n-1)], dzf01_f=[dzf(1),dzf(n)]
  if (has) has[0] = has[1] = 0;
  for (int ib = 0; ib <= 1; ++ib) {
    if (!ISB(c, ib, 3) || cbc[ib] == 'P') continue;
    if (has) has[ib] = 1;
    const real dlc = cf == 'c' ? (ib ? c->dzc[n3] : c->dzc[0]) : (ib ? c->dzc[n3 - 1] : c->dzc[1]);
    const real dlf = ib ? c->dzf[n3] : c->dzf[1];
    LAUNCH(c, k_rhs_b_velz, dim3((n[0] + 63) / 64, (n[1] + 3) / 4), dim3(64, 4), 0, c->stream, c->g, c->f[CALES_U + ivel - 1],
                       plane(bc, 3, ib, n), ib, cbc[ib], cf, dlc, dlf, alpha, ib ? n3 - q : 1, planes ? planes + (size_t)ib * n[0] * n[1] : nullptr);
  }
  LAUNCHCHK(c);
  return 0;
}

int op_xwrap_zghost(cales_ctx *c, int nf, real **f) {
  HaloFields H; H.nf = nf; for (int q = 0; q < nf; ++q) H.p[q] = f[q];
  LAUNCH(c, k_xwrap_zghost, dim3((c->n[1] + 2 + 255) / 256, 2, nf), dim3(256), 0, c->stream, c->g, H);
  LAUNCHCHK(c);
  return 0;
}
