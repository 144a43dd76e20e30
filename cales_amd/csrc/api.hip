// extern "C" entry points of libcales_hip.so (include/cales.h): context life cycle, host<->device copies,
// one entry per reference operator, the fused time step (reference src/main.f90:417-508) and the
// HIP-event kernel timers used by bench.py's roofline line.
#include "common.hpp"

static thread_local std::string g_create_err;

// ------------------------------------------------------------------------------------------ launch check (common.hpp, LAUNCH)
void launch_failed(cales_ctx *c, const char *what, hipError_t e) {
  if (c->launch_err.empty()) c->launch_err = std::string("kernel launch failed: ") + what + ": " + hipGetErrorString(e) + " (the context is unusable from here on)";
}
// every operator entry: refuse a failed context, run, and report a launch that failed on the way
static int finish_pending(cales_ctx *c);
// ... and first of all completes a projection that cales_step left to its successor (common.hpp, pend_xskip)
#define ENTER(c) do { LAUNCHCHK(c); if (const int pe_ = finish_pending(c)) return pe_; } while (0)
#define ENTRY(c, expr) do { ENTER(c); const int e_ = (expr); if (e_) return e_; LAUNCHCHK(c); return 0; } while (0)

// ------------------------------------------------------------------------------------------ profiling
int stream_after(cales_ctx *c, hipStream_t later, hipStream_t earlier) {
  if (c->sync_ev.size() < 64) { hipEvent_t e; HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->sync_ev.push_back(e); c->sync_next = c->sync_ev.size() - 1; }
  hipEvent_t e = c->sync_ev[c->sync_next]; c->sync_next = (c->sync_next + 1) % c->sync_ev.size();
  HIPCHK(c, hipEventRecord(e, earlier));
  HIPCHK(c, hipStreamWaitEvent(later, e, 0));
  return 0;
}
int prof_begin(cales_ctx *c, const char *name, hipStream_t s) {
  int slot = -1;
  for (size_t q = 0; q < c->stats.size(); ++q) if (c->stats[q].name == name) { slot = (int)q; break; }
  if (slot < 0) { KernelStat s; s.name = name; c->stats.push_back(s); slot = (int)c->stats.size() - 1; }
  std::pair<hipEvent_t, hipEvent_t> ev;
  if (!c->evpool.empty()) { ev = c->evpool.back(); c->evpool.pop_back(); }
  else { hipEventCreate(&ev.first); hipEventCreate(&ev.second); }
  hipEventRecord(ev.first, s ? s : c->stream);
  c->pending.push_back({slot, ev});
  return (int)c->pending.size() - 1;
}
void prof_end(cales_ctx *c, int idx, hipStream_t s) { hipEventRecord(c->pending[idx].second.second, s ? s : c->stream); }
void prof_flush(cales_ctx *c) {
  if (c->pending.empty()) return;
  hipStreamSynchronize(c->stream);
  if (c->comm_stream) hipStreamSynchronize(c->comm_stream);
  for (auto &pe : c->pending) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, pe.second.first, pe.second.second);
    c->stats[pe.first].calls += 1; c->stats[pe.first].ms += ms;
    c->evpool.push_back(pe.second);
  }
  c->pending.clear();
}

// ------------------------------------------------------------------------------------------ helpers
static int dev_alloc(cales_ctx *c, real **p, size_t n, bool zero = true) {
  HIPCHK(c, hipMalloc(p, (n ? n : 1) * sizeof(real)));
  if (zero) HIPCHK(c, hipMemset(*p, 0, (n ? n : 1) * sizeof(real)));
  return 0;
}
// 3-D fields: pitch-padded rows, shifted so that element (1,j,k) sits on a 128-B boundary (see cales_create). (Skewing the start of
// every field by a different number of cache lines, so that the same cell of different fields does not fall on the same L2 set,
// was measured at 512^3 with six strides and changed no kernel by more than the run-to-run noise.)
static int field_alloc(cales_ctx *c, real **p) {
  real *base = nullptr;
  if (dev_alloc(c, &base, c->ntot + LINE_REALS)) return 1;
  *p = base + c->field_ofs;
  return 0;
}
static void field_free(cales_ctx *c, real *p) { if (p) hipFree(p - c->field_ofs); }
// k fields in ONE allocation, each right behind the one before: a kernel that addresses the first with 32-bit byte offsets reaches the others with
// multiples of the constant c->pp_companion_bytes added (the correction pressure and its companions, the velocity and its companion: the second /
// third ghost rows of k_corr_strain_tile on several slabs). p[0] is what field_free takes.
static int field_alloc_multi(cales_ctx *c, int k, real **p) {
  real *base = nullptr;
  const size_t one = c->ntot + LINE_REALS;
  if (dev_alloc(c, &base, (size_t)k * one)) return 1;
  for (int q = 0; q < k; ++q) p[q] = base + (size_t)q * one + c->field_ofs;
  c->pp_companion_bytes = one * sizeof(real); c->comp_one = one;
  return 0;
}
// host layout (0:n1+1,0:n2+1,0:n3+1), x contiguous  <->  device layout with row pitch s1
__global__ __launch_bounds__(256) void k_repack(Geom g, int to_device, real *__restrict__ dev, real *__restrict__ packed) {
  const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
  if (i > g.n1 + 1 || j > g.n2 + 1) return;
  const size_t h = (size_t)i + (size_t)(g.n1 + 2) * ((size_t)j + (size_t)(g.n2 + 2) * k);
  if (to_device) dev[g.ix(i, j, k)] = packed[h]; else packed[h] = dev[g.ix(i, j, k)];
}
// Same-box calibration (bench.py "calibration"): three streams over the rows of the context's own fields in the library's layout
// (padded rows, element 1 of every row on a 128-B boundary, 8 B per lane -- what the FP64 passes do), timed with HIP events on the context's stream.
// MODE 0: read-only, 1: write-only, 2: copy. Rows are dealt to the blocks round-robin (the order of tools/micro/calib.hip's calib_copy8_rows).
template <int MODE>
__global__ __launch_bounds__(256) void k_calib_rows(Geom g, long nrows, const real *__restrict__ a, real *__restrict__ b, real *__restrict__ sink) {
  real acc = 0.;
  for (long r = blockIdx.x; r < nrows; r += gridDim.x) {
    const size_t o = (size_t)r * g.s1 + 1;
    for (int i = threadIdx.x; i < g.n1; i += 256) {
      if (MODE == 0) acc += a[o + i];
      else if (MODE == 1) b[o + i] = (real)1.5;
      else b[o + i] = a[o + i];
    }
  }
  if (MODE == 0 && acc == (real)1.2345e30) sink[blockIdx.x] = acc;      // never true: no write traffic
}
static int upload_vec(cales_ctx *c, real **p, const std::vector<real> &v) {
  if (dev_alloc(c, p, v.size(), false)) return 1;
  HIPCHK(c, hipMemcpy(*p, v.data(), v.size() * sizeof(real), hipMemcpyHostToDevice));
  return 0;
}
static int upload_bound(cales_ctx *c, DBound &b, std::vector<real> h[3]) {
  return upload_vec(c, &b.x, h[0]) || upload_vec(c, &b.y, h[1]) || upload_vec(c, &b.z, h[2]);
}
static void free_bound(DBound &b) { hipFree(b.x); hipFree(b.y); hipFree(b.z); }

extern "C" {

// ------------------------------------------------------------------------------------------ host-only helpers
int cales_initgrid(int gtype, int n, real gr, real lz, real *dzc, real *dzf, real *zc, real *zf) {
  if (n < 1 || !dzc || !dzf || !zc || !zf) return 1;
  hs_initgrid(gtype, n, gr, lz, dzc, dzf, zc, zf);
  return 0;
}
int cales_initflow(const cales_case *cs, const char *inivel, int is_wallturb, real *u, real *v, real *w, real *p) {
  if (!cs || !inivel || !u || !v || !w || !p) return 1;
  return hs_initflow(cs, inivel, is_wallturb, u, v, w, p, 0, 1);
}
int cales_check_case(const cales_case *cs, char *msg, int msglen) {
  std::string m;
  const int rc = hs_check_case(cs, m);
  if (msg && msglen > 0) { std::snprintf(msg, msglen, "%s", m.c_str()); }
  return rc;
}

int cales_real_size(void) { return (int)sizeof(real); }
int cales_device_count(int *ndev) { if (!ndev) return 1; *ndev = 0; return hipGetDeviceCount(ndev) == hipSuccess ? 0 : 2; }
int cales_set_device(int dev) { return hipSetDevice(dev) == hipSuccess ? 0 : 1; }

// ------------------------------------------------------------------------------------------ context
const char *cales_last_error(const cales_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

void cales_destroy(cales_ctx *c) {
  if (!c) return;
  hipStreamSynchronize(c->stream);
  prof_flush(c);
  for (auto &ev : c->evpool) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); }
  if (c->comm_stream) { hipStreamSynchronize(c->comm_stream); hipStreamDestroy(c->comm_stream); }
  for (auto &e : c->sync_ev) hipEventDestroy(e);
  solver_teardown(c);
  for (int q = 0; q < CALES_NFIELDS; ++q) field_free(c, c->f[q]);
  for (int q = 0; q < 3; ++q) field_free(c, c->f2[q]);
  hipFree(c->d_dzc); hipFree(c->d_dzf); hipFree(c->d_zc); hipFree(c->d_zf); hipFree(c->d_dzci); hipFree(c->d_dzfi); hipFree(c->d_gvr_c); hipFree(c->d_gvr_f);
  DBound *bs[11] = {&c->bcu, &c->bcv, &c->bcw, &c->bcp, &c->bcs, &c->bcuf, &c->bcvf, &c->bcwf, &c->bcu_mag, &c->bcv_mag, &c->bcw_mag};
  for (auto *b : bs) free_bound(*b);
  for (int d = 0; d < 3; ++d) hipFree(c->rhsbp[d]);
  field_free(c, c->scr1); hipFree(c->d_red); hipFree(c->d_force); if (c->d_mpart) hipFree(c->d_mpart); if (c->d_abct) hipFree(c->d_abct); if (c->d_hztab) hipFree(c->d_hztab); if (c->d_cs) hipFree(c->d_cs); if (c->d_nullw) hipFree(c->d_nullw); if (c->d_stat) hipFree(c->d_stat); if (c->d_stat2) hipFree(c->d_stat2); hipHostFree(c->h_red);
  field_free(c, c->s0); field_free(c, c->uc); field_free(c, c->vc); field_free(c, c->wc); field_free(c, c->uf); field_free(c, c->vf); field_free(c, c->wf); field_free(c, c->alph2); if (!c->p1d_in_comm) hipFree(c->d_p1d);
  for (int m = 0; m < 6; ++m) { field_free(c, c->wk[m]); field_free(c, c->sij[m]); field_free(c, c->mij[m]); }
  for (int m = 0; m < 3; ++m) if (c->ss2[m]) hipFree(c->ss2[m] - 2 * c->field_ofs);
  cales_comm_release_native(c);
  hipFree(c->d_del);
  if (c->own_stream) hipStreamDestroy(c->stream);
  delete c;
}

int cales_create(const cales_case *cs, void *stream, cales_ctx **out) {
  if (!cs || !out) { g_create_err = "null argument"; return 1; }
  *out = nullptr;
  std::string msg;
  if (hs_check_case(cs, msg)) { g_create_err = msg; return 2; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_create_err = "no HIP device: the CaLES hot path has no CPU fallback"; return 3; }
  cales_ctx *c = new cales_ctx();
  c->C = *cs;
  c->fl.read_env();      // the CALES_* switches are fixed for the life of the context
  // zero all device pointers
  for (auto &p : c->f) p = nullptr;
  c->d_dzc = c->d_dzf = c->d_zc = c->d_zf = c->d_dzci = c->d_dzfi = c->d_gvr_c = c->d_gvr_f = nullptr;
  DBound *bs[11] = {&c->bcu, &c->bcv, &c->bcw, &c->bcp, &c->bcs, &c->bcuf, &c->bcvf, &c->bcwf, &c->bcu_mag, &c->bcv_mag, &c->bcw_mag};
  for (auto *b : bs) b->x = b->y = b->z = nullptr;
  for (int d = 0; d < 3; ++d) { c->rhsbp[d] = nullptr; c->d_av[d] = c->d_bv[d] = c->d_cv[d] = nullptr; }
  c->rhsbz_vel = c->d_lamx = c->d_lamy = c->d_a = c->d_b = c->d_c = c->d_twx = c->d_twy = c->d_twx_post = c->d_twy_post = nullptr;
  c->scr1 = c->scr2 = c->d_red = c->h_red = c->d_force = nullptr;
  c->s0 = c->uc = c->vc = c->wc = c->uf = c->vf = c->wf = c->alph2 = c->d_p1d = nullptr;
  for (int m = 0; m < 6; ++m) c->wk[m] = c->sij[m] = c->mij[m] = nullptr;
  c->sgs_first = true;
  auto fail = [&](int rc) { g_create_err = c->err; cales_destroy(c); return rc; };
  if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
  else { c->own_stream = true; if (hipStreamCreate(&c->stream) != hipSuccess) { c->own_stream = false; c->stream = 0; c->err = "hipStreamCreate failed"; return fail(4); } }
  // geometry: y-slab of rank `rank`
  const int P = cs->nranks, r = cs->rank;
  c->P = P; c->rank = r; c->per_y = cs->cbcpre[2] == 'P' && cs->cbcpre[3] == 'P';
  c->n[0] = cs->ng[0]; c->n[1] = cs->ng[1] / P; c->n[2] = cs->ng[2];
  c->lo[0] = 1; c->lo[1] = r * c->n[1] + 1; c->lo[2] = 1;
  for (int d = 0; d < 3; ++d) { c->dl[d] = cs->l[d] / (real)(1.f * (float)cs->ng[d]); c->dli[d] = 1. / c->dl[d]; }   // param.f90:153-154
  c->visc = 1. / cs->visci;
  Geom &g = c->g;
  g.n1 = c->n[0]; g.n2 = c->n[1]; g.n3 = c->n[2];
  // Row pitch: a multiple of one cache line (128 B = 16 doubles / 32 floats) with room for the n1/2+1 complex modes of a row stored from i = 1; with the
  // one-line-minus-one-element offset of dev_alloc, element i = 1 of every row is 128-B aligned, so kernels whose waves handle 64 consecutive
  // cells from i = 1 read and write whole cache lines (partial-line writes cost ~1.5x, tools/micro/wrtile.hip).
  g.s1 = (g.n1 + 3 + LINE_REALS - 1) / LINE_REALS * LINE_REALS;
  c->field_ofs = LINE_REALS - 1;
  g.s12 = (long)g.s1 * (g.n2 + 2); g.jlo = c->lo[1] - 1; g.ng2 = cs->ng[1];
  c->ntot = (size_t)g.s12 * (g.n3 + 2);
  const int n3 = c->n[2];
  // z grid and metrics (main.f90:246-285)
  c->dzc.resize(n3 + 2); c->dzf.resize(n3 + 2); c->zc.resize(n3 + 2); c->zf.resize(n3 + 2);
  c->dzci.resize(n3 + 2); c->dzfi.resize(n3 + 2); c->gvr_c.resize(n3 + 2); c->gvr_f.resize(n3 + 2);
  hs_initgrid(cs->gtype, n3, cs->gr, cs->l[2], c->dzc.data(), c->dzf.data(), c->zc.data(), c->zf.data());
  for (int k = 0; k <= n3 + 1; ++k) {
    c->dzci[k] = 1. / c->dzc[k]; c->dzfi[k] = 1. / c->dzf[k];
    c->gvr_c[k] = c->dl[0] * c->dl[1] * c->dzc[k] / (cs->l[0] * cs->l[1] * cs->l[2]);
    c->gvr_f[k] = c->dl[0] * c->dl[1] * c->dzf[k] / (cs->l[0] * cs->l[1] * cs->l[2]);
  }
  if (upload_vec(c, &c->d_dzc, c->dzc) || upload_vec(c, &c->d_dzf, c->dzf) || upload_vec(c, &c->d_zc, c->zc) || upload_vec(c, &c->d_zf, c->zf) ||
      upload_vec(c, &c->d_dzci, c->dzci) || upload_vec(c, &c->d_dzfi, c->dzfi) || upload_vec(c, &c->d_gvr_c, c->gvr_c) || upload_vec(c, &c->d_gvr_f, c->gvr_f))
    return fail(5);
  // which faces are physical boundaries of this slab (initmpi.f90:201-204 for x-pencils; y is the decomposed direction)
  c->is_bound[0] = c->is_bound[1] = 1;
  { const bool per_y = cs->cbcpre[2] == 'P' && cs->cbcpre[3] == 'P';
    c->is_bound[2] = (!per_y && r == 0) ? 1 : 0; c->is_bound[3] = (!per_y && r == P - 1) ? 1 : 0;
    const bool per_z = cs->cbcpre[4] == 'P' && cs->cbcpre[5] == 'P';
    c->is_bound[4] = c->is_bound[5] = per_z ? 0 : 1; }
  // boundary-condition tables (bound.f90:726-867)
  std::vector<real> hb[11][3];
  hs_initbc(c, hb);
  for (int q = 0; q < 11; ++q) if (upload_bound(c, *bs[q], hb[q])) return fail(6);
  // pressure boundary r.h.s. (main.f90:317, bound.f90:447-499)
  { const int *n = c->n;
    const real dx01[2] = {c->dl[0], c->dl[0]}, dy01[2] = {c->dl[1], c->dl[1]};
    const real dzc01[2] = {c->dzc[0], c->dzc[n3]}, dzf01[2] = {c->dzf[1], c->dzf[n3]};
    std::vector<real> rx((size_t)n[1] * n[2] * 2), ry((size_t)n[0] * n[2] * 2), rz((size_t)n[0] * n[1] * 2);
    hs_bc_rhs(&cs->cbcpre[0], hb[3][0].data(), n[1], n[2], dx01, dx01, 'c', rx.data());
    hs_bc_rhs(&cs->cbcpre[2], hb[3][1].data(), n[0], n[2], dy01, dy01, 'c', ry.data());
    hs_bc_rhs(&cs->cbcpre[4], hb[3][2].data(), n[0], n[1], dzc01, dzf01, 'c', rz.data());
    if (upload_vec(c, &c->rhsbp[0], rx) || upload_vec(c, &c->rhsbp[1], ry) || upload_vec(c, &c->rhsbp[2], rz)) return fail(7); }
  // fields (haloed); r.h.s. buffers use the same layout so every kernel shares one index
  const int nfields = cs->impdiff ? CALES_NFIELDS : CALES_DUDTD;
  // several slabs with the dynamic model: companions behind u, v, w (both buffer sets) and two behind pp (common.hpp, vel_comp)
  c->vel_comp = P > 1 && cs->sgstype == 2;
  for (int q = 0; q < nfields; ++q) {
    if (q == CALES_PP) continue;
    if (c->vel_comp && q <= CALES_W) { real *two[2]; if (field_alloc_multi(c, 2, two)) return fail(8); c->f[q] = two[0]; }
    else if (field_alloc(c, &c->f[q])) return fail(8);
  }
  { real *three[3]; if (field_alloc(c, &c->scr1) || field_alloc_multi(c, c->vel_comp ? 3 : 2, three)) return fail(9);      // (scr2, scr3 live in pp's allocation: freed with it)
    c->f[CALES_PP] = three[0]; c->scr2 = three[1]; c->scr3 = c->vel_comp ? three[2] : nullptr; }
  for (int q = 0; q < 3; ++q) {
    if (c->vel_comp) { real *two[2]; if (field_alloc_multi(c, 2, two)) return fail(9); c->f2[q] = two[0]; }
    else if (field_alloc(c, &c->f2[q])) return fail(9);
  }
  c->red_blocks = 8;
  if (dev_alloc(c, &c->d_red, 64 + 16 * (size_t)(n3 + 2) + 6 * (size_t)(n3 + 2)) || dev_alloc(c, &c->d_force, 8)) return fail(10);
  c->res = c->d_red;
  if (hipHostMalloc((void **)&c->h_red, 64 * sizeof(real)) != hipSuccess) { c->err = "hipHostMalloc failed"; return fail(11); }
  // sgs scratch (sgs.f90:70-83,154-171)
  for (int d = 1; d <= 3; ++d) for (int s = 0; s <= 1; ++s) c->is_wall[s + 2 * (d - 1)] = (ISB(c, s, d) && CBV(c, s, d, d) == 'D') ? 1. : 0.;
  // wall flags are global properties of the case, not of the slab (distances use global indices)
  for (int s = 0; s <= 1; ++s) c->is_wall[s + 2] = (!(cs->cbcpre[2] == 'P' && cs->cbcpre[3] == 'P') && CBV(c, s, 2, 2) == 'D') ? 1. : 0.;
  if (cs->sgstype >= 1) {
    if (field_alloc(c, &c->s0)) return fail(12);
    const int nw = cs->sgstype == 1 ? 3 : 6;
    for (int m = 0; m < nw; ++m) if (field_alloc(c, &c->wk[m])) return fail(12);
  }
  if (cs->sgstype == 2) {
    if (field_alloc(c, &c->uc) || field_alloc(c, &c->vc) || field_alloc(c, &c->wc) || field_alloc(c, &c->uf) ||
        field_alloc(c, &c->vf) || field_alloc(c, &c->wf) || field_alloc(c, &c->alph2) || dev_alloc(c, &c->d_p1d, 2 * (size_t)n3 + 2))
      return fail(13);
    if (dsmag_pairs(c)) {      // |S|Sij as three fields of pairs between K_AC and the fused last pass: the twelve scalar scratch fields of the other forms are not needed
      for (int m = 0; m < 3; ++m) { real *b = nullptr; if (dev_alloc(c, &b, 2 * c->ntot + 2 * LINE_REALS)) return fail(13); c->ss2[m] = b + 2 * c->field_ofs; }
    } else
    for (int m = 0; m < 6; ++m) if (field_alloc(c, &c->sij[m]) || field_alloc(c, &c->mij[m])) return fail(13);
  }
  if (solver_setup(c)) return fail(14);
  { hipDeviceProp_t pr; int dev = 0; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) c->ncu = pr.multiProcessorCount; }
  if (hipStreamSynchronize(c->stream) != hipSuccess) { c->err = "sync failed"; return fail(15); }
  if (!c->launch_err.empty()) { c->err = c->launch_err; return fail(16); }      // a set-up kernel (twiddles, tables) or attribute call failed
  *out = c;
  return 0;
}

int cales_sync(cales_ctx *c) { ENTER(c); HIPCHK(c, hipStreamSynchronize(c->stream)); return 0; }      // (what has been asked for is done when this returns: a pending projection too)
int cales_local_size(const cales_ctx *c, int32_t n[3], int32_t lo[3]) { for (int d = 0; d < 3; ++d) { n[d] = c->n[d]; lo[d] = c->lo[d]; } return 0; }

// ------------------------------------------------------------------------------------------ copies
int cales_set_field(cales_ctx *c, int field, const real *host) {
  ENTER(c);
  if (field < 0 || field >= CALES_NFIELDS || !c->f[field]) { c->err = "bad field id"; return 1; }
  if (field == CALES_VISCT) { c->visct_zero = false; c->visct_lazy = false; }
  const size_t nh = (size_t)(c->n[0] + 2) * (c->n[1] + 2) * (c->n[2] + 2);
  const dim3 b(64, 4, 1), gr((c->n[0] + 2 + 63) / 64, (c->n[1] + 2 + 3) / 4, c->n[2] + 2);
  HIPCHK(c, hipMemcpyAsync(c->scr1, host, nh * sizeof(real), hipMemcpyHostToDevice, c->stream));     // scr1: scratch between operators
  LAUNCH(c, k_repack, gr, b, 0, c->stream, c->g, 1, c->f[field], c->scr1);
  LAUNCHCHK(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
int cales_get_field(cales_ctx *c, int field, real *host) {
  ENTER(c);
  if (field < 0 || field >= CALES_NFIELDS || !c->f[field]) { c->err = "bad field id"; return 1; }
  if (field == CALES_VISCT) if (int e = materialize_visct(c)) return e;
  const size_t nh = (size_t)(c->n[0] + 2) * (c->n[1] + 2) * (c->n[2] + 2);
  const dim3 b(64, 4, 1), gr((c->n[0] + 2 + 63) / 64, (c->n[1] + 2 + 3) / 4, c->n[2] + 2);
  LAUNCH(c, k_repack, gr, b, 0, c->stream, c->g, 0, c->f[field], c->scr1);
  LAUNCHCHK(c);
  HIPCHK(c, hipMemcpyAsync(host, c->scr1, nh * sizeof(real), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
int cales_upload_state(cales_ctx *c, const real *u, const real *v, const real *w, const real *p) {
  return cales_set_field(c, CALES_U, u) || cales_set_field(c, CALES_V, v) || cales_set_field(c, CALES_W, w) || cales_set_field(c, CALES_P, p);
}
int cales_download_state(cales_ctx *c, real *u, real *v, real *w, real *p, real *visct) {
  if (u && cales_get_field(c, CALES_U, u)) return 1;
  if (v && cales_get_field(c, CALES_V, v)) return 1;
  if (w && cales_get_field(c, CALES_W, w)) return 1;
  if (p && cales_get_field(c, CALES_P, p)) return 1;
  if (visct && cales_get_field(c, CALES_VISCT, visct)) return 1;
  return 0;
}
int cales_get_bcvel(cales_ctx *c, int ivel, real *x, real *y, real *z) {
  const DBound &b = ivel == 1 ? c->bcu : ivel == 2 ? c->bcv : c->bcw; const int *n = c->n;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(x, b.x, sizeof(real) * (size_t)(n[1] + 2) * (n[2] + 2) * 2, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(y, b.y, sizeof(real) * (size_t)(n[0] + 2) * (n[2] + 2) * 2, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(z, b.z, sizeof(real) * (size_t)(n[0] + 2) * (n[1] + 2) * 2, hipMemcpyDeviceToHost));
  return 0;
}

// ------------------------------------------------------------------------------------------ operators
int cales_bounduvw(cales_ctx *c, int is_updt_wm, int is_correc) {
  ENTRY(c, op_bounduvw(c, c->bcu, c->bcv, c->bcw, is_updt_wm, is_correc, c->f[CALES_U], c->f[CALES_V], c->f[CALES_W]));
}
int cales_boundp(cales_ctx *c, int field, int which) {
  if (field < 0 || field >= CALES_NFIELDS || !c->f[field]) { c->err = "bad field id"; return 1; }
  ENTRY(c, op_boundp(c, c->f[field], which));
}
int cales_get_forcing(cales_ctx *c, real f[3]);
int cales_mom(cales_ctx *c) { ENTRY(c, op_mom(c)); }
int cales_rk(cales_ctx *c, int irk, real dt) { if (irk < 1 || irk > 3) { c->err = "irk must be 1..3"; return 1; } ENTRY(c, op_rk(c, irk, dt)); }
int cales_rk_par(cales_ctx *c, const real rkpar[2], real dt, real f_out[3]) {
  if (!rkpar) { c->err = "cales_rk_par: rkpar is NULL"; return 1; }
  ENTER(c);
  if (int e = op_rk_par(c, rkpar[0], rkpar[1], dt)) return e;
  LAUNCHCHK(c);
  return f_out ? cales_get_forcing(c, f_out) : 0;
}
int cales_bulk_forcing(cales_ctx *c) { ENTRY(c, op_bulk_forcing(c)); }
int cales_get_forcing(cales_ctx *c, real f[3]) {
  HIPCHK(c, hipMemcpyAsync(c->h_red + 32, c->d_force, 3 * sizeof(real), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int q = 0; q < 3; ++q) f[q] = c->h_red[32 + q];
  return 0;
}
int cales_bulk_mean(cales_ctx *c, int field, int c_or_f, real *mean) {
  if (field < 0 || field >= CALES_NFIELDS || !c->f[field]) { c->err = "bad field id"; return 1; }
  ENTER(c);
  if (int e = op_bulk_mean_dev(c, c->f[field], c_or_f, nullptr)) return e;
  LAUNCHCHK(c);
  HIPCHK(c, hipMemcpyAsync(c->h_red + 16, c->res + 16, sizeof(real), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *mean = c->h_red[16];
  return 0;
}
int cales_fillps(cales_ctx *c, real dtrki) { ENTRY(c, op_fillps(c, dtrki)); }
int cales_updt_rhs_b(cales_ctx *c) { ENTRY(c, op_updt_rhs_b(c)); }
int cales_solver(cales_ctx *c) { ENTRY(c, op_solver(c)); }
int cales_helmholtz(cales_ctx *c, int ivel, real alpha) { ENTRY(c, op_helmholtz(c, ivel, alpha)); }
int cales_helmholtz_z(cales_ctx *c, int ivel, real alpha) { if (ivel < 1 || ivel > 3) { c->err = "ivel must be 1..3"; return 1; } ENTRY(c, op_helmholtz_z(c, ivel, alpha)); }
int cales_correc(cales_ctx *c, real dtrk) { ENTRY(c, op_correc(c, dtrk)); }
int cales_updatep(cales_ctx *c, real alpha) { ENTRY(c, op_updatep(c, alpha)); }
int cales_cmpt_sgs(cales_ctx *c) { ENTRY(c, op_cmpt_sgs(c)); }
int cales_chkdt(cales_ctx *c, real *dtmax) { ENTRY(c, op_chkdt(c, dtmax)); }
int cales_chkdiv(cales_ctx *c, real *divtot, real *divmax) { ENTRY(c, op_chkdiv(c, divtot, divmax)); }
int cales_out1d_single_point_chan(cales_ctx *c, real *buf) { if (!c || !buf) return 1; ENTRY(c, op_stats_chan(c, buf)); }
int cales_out1d_chan_budgets(cales_ctx *c, real *budget, real *leakage) { if (!c) return 1; ENTRY(c, op_stats_chan_budget(c, budget, leakage)); }
int cales_out1d(cales_ctx *c, int field, int idir, int use_dzc, real *buf) { if (!c || !buf) return 1; ENTRY(c, op_out1d(c, field, idir, use_dzc, buf)); }
int cales_out1d_chan(cales_ctx *c, real *buf) { if (!c || !buf) return 1; ENTRY(c, op_out1d_chan(c, buf)); }
int cales_out2d_duct(cales_ctx *c, real *buf) { if (!c || !buf) return 1; ENTRY(c, op_out2d_duct(c, buf)); }

// ------------------------------------------------------------------------------------------ time step (main.f90:412-508)
__global__ void k_zero6(real *f, int first) { if ((int)threadIdx.x >= first && threadIdx.x < 6) f[threadIdx.x] = 0.; }      // first = 3: dpdl only (the increments f(0:2) of a pending projection are still needed)
__global__ __launch_bounds__(256) void k_row2_to_companion(Geom g, const real *__restrict__ pp, real *__restrict__ comp) {
  const int i = blockIdx.x * 64 + threadIdx.x, k = blockIdx.y * 4 + threadIdx.y;
  if (i > g.n1 + 1 || k > g.n3 + 1) return;
  comp[g.ix(i, 1, k)] = pp[g.ix(i, 2, k)];
}

// the projection of a substep may be left to the next momentum pass (step_body): static conditions of the case and the switches
static bool fold_mom_ok(const cales_ctx *c) {
  bool ok = !c->fl.unfolded_mom && c->C.sgstype == 0 && c->visct_zero && !c->sgs_first && (c->C.impdiff == 0 || c->C.impdiff == 2) && (c->P == 1 || c->comm.on) && c->n[2] >= 3 &&
            !c->fl.unfused_rk && !c->fl.unfused_correc;
  for (int q = 0; q < 6; ++q) ok = ok && c->C.lwm[q] == 0;
  for (int d = 1; d <= 3 && ok; ++d) {
    bool per = CBP(c, 0, d) == 'P' && CBP(c, 1, d) == 'P', walls = CBP(c, 0, d) == 'N' && CBP(c, 1, d) == 'N' && c->C.bcpre[2 * (d - 1)] == 0. && c->C.bcpre[2 * (d - 1) + 1] == 0.;
    // (walls: the normal component is prescribed -- with the homogeneous Neumann pressure its face values do not change in the projection --, the tangential
    //  ones no-slip or free-slip: either rule is an affine map of the mirrored interior cell, which the corrected view reads projected)
    for (int iv = 1; iv <= 3; ++iv) for (int sd = 0; sd <= 1; ++sd) { per = per && CBV(c, sd, d, iv) == 'P'; walls = walls && (CBV(c, sd, d, iv) == 'D' || (iv != d && CBV(c, sd, d, iv) == 'N')); }
    ok = per || walls;
  }
  return ok;
}
// correction + pressure update of a substep as passes of their own, and the ghost cells of what they produced (main.f90:498-504)
static int project_now(cales_ctx *c, real dtrk, real alpha) {
  const bool fuse_cu = !c->fl.unfused_correc && c->C.impdiff != 1;     // updatep only needs pp: one pass with correc
  { const int e = fuse_cu ? op_correc_updatep(c, dtrk, alpha, 1) : op_correc(c, dtrk); c->defer_force = false; if (e) return e; }
  // the pressure is final once the fused correction has run: its ghost cells ride along with those of the velocity (one launch, one slab exchange)
  if (fuse_cu && !c->fl.unmerged_bc) { c->bc_nride = 1; c->bc_ride[0] = c->f[CALES_P]; c->bc_ride_which[0] = 0; }
  const int e = op_bounduvw(c, c->bcu, c->bcv, c->bcw, 1, 1, c->f[CALES_U], c->f[CALES_V], c->f[CALES_W]);
  const bool rode = fuse_cu && !c->fl.unmerged_bc && c->bc_nride == 0; c->bc_nride = 0;
  if (e) return e;
  if (!fuse_cu) { if (int e2 = op_updatep(c, alpha)) return e2; }
  if (!rode) { if (int e2 = op_boundp(c, c->f[CALES_P], 0)) return e2; }
  return 0;
}
// the ghost cells of everything a caller may look at, all directions (the corners of the x ghost columns with the z ghost planes included), when the
// step left the x ghost columns alone (step_xskip)
static int end_of_step_refresh(cales_ctx *c) {
  if (!c->step_xskip) return 0;
  c->step_xskip = false;
  c->bc_nride = 3; c->bc_ride[0] = c->f[CALES_P]; c->bc_ride[1] = c->f[CALES_PP]; c->bc_ride[2] = c->f[CALES_VISCT];
  c->bc_ride_which[0] = 0; c->bc_ride_which[1] = 0; c->bc_ride_which[2] = 1;
  if (c->fl.unmerged_bc) c->bc_nride = 0;
  c->bc_no_halo = true;      // (only the x ghost columns are stale: the rows the neighbours sent are complete but for their two ends, which the local copies fill)
  struct NoHalo { cales_ctx *c; ~NoHalo() { c->bc_no_halo = false; } } nohalo{c};
  const int e = op_bounduvw(c, c->bcu, c->bcv, c->bcw, 0, 1, c->f[CALES_U], c->f[CALES_V], c->f[CALES_W]);
  const bool rode = !c->fl.unmerged_bc && c->bc_nride == 0; c->bc_nride = 0;
  if (e) return e;
  if (!(CBV(c, 0, 3, 3) == 'P' && CBV(c, 1, 3, 3) == 'P')) { if (int e2 = op_xwrap_zghost(c, 3, c->f + CALES_U)) return e2; }      // (periodic z: the z copies of the launch above cover the corners)
  if (!rode) {
    real *pq[2] = {c->f[CALES_P], c->f[CALES_PP]};
    if (int e2 = op_boundp_multi(c, 2, pq, 0)) return e2;
    if (int e2 = op_boundp(c, c->f[CALES_VISCT], 1)) return e2;
  }
  return 0;
}
// A projection that cales_step left to its successor (fold_mom, third substep) is completed here -- correction pass, ghost cells, and the refresh of
// the x ghost columns -- before anything else looks at the fields. Collective over the ranks like every entry of the C-ABI.
static int finish_pending(cales_ctx *c) {
  if (c->in_step) return 0;
  if (c->fold_mom_dtrk == 0.) {
    if (!c->pend_xrefresh) return 0;
    // only the refresh of the x ghost columns is due (cales_step with step_xskip, common.hpp)
    c->pend_xrefresh = false;
    c->in_step = true; c->step_xskip = true;
    struct Restore { cales_ctx *c; ~Restore() { c->in_step = false; c->step_xskip = false; c->bc_nride = 0; } } restore{c};
    if (int e = end_of_step_refresh(c)) { c->launch_err = "refreshing the x ghost columns failed (" + c->err + "): the context is unusable"; return e; }
    LAUNCHCHK(c);
    return 0;
  }
  c->pend_xrefresh = false;      // (the completion below ends with the refresh)
  const real dtrk = c->fold_mom_dtrk;
  c->fold_mom_dtrk = 0.;
  c->in_step = true; c->step_xskip = c->pend_xskip; c->defer_force = c->fold_mom_fmask != 0;
  struct Restore { cales_ctx *c; ~Restore() { c->in_step = false; c->step_xskip = false; c->defer_force = false; c->bc_nride = 0; } } restore{c};
  if (c->fold_mom_pdone) {      // (z-implicit diffusion: the pressure is up to date, ghost cells included)
    c->fold_mom_pdone = false;
    int e = op_correc(c, dtrk);
    if (!e) e = op_bounduvw(c, c->bcu, c->bcv, c->bcw, 1, 1, c->f[CALES_U], c->f[CALES_V], c->f[CALES_W]);
    if (e) { c->launch_err = "completing a pending projection failed (" + c->err + "): the context is unusable"; return e; }
  } else
  if (int e = project_now(c, dtrk, 0.)) { c->launch_err = "completing a pending projection failed (" + c->err + "): the context is unusable"; return e; }
  if (int e = end_of_step_refresh(c)) { c->launch_err = "completing a pending projection failed (" + c->err + "): the context is unusable"; return e; }
  LAUNCHCHK(c);
  return 0;
}
// ---- the plan of a step (StepPlan, common.hpp): every decision about WHICH form of an operator a substep takes is made here, from the case, the
// switches and the state recorded in the plan's in_* fields -- step_body below only reads the result.
static void make_plan(cales_ctx *c) {
  StepPlan pl;
  pl.valid = true;
  pl.in_visct_zero = c->visct_zero; pl.in_sgs_first = c->sgs_first; pl.in_comm_on = c->comm.on; pl.in_overlap = c->comm_stream != nullptr;
  const Flags &fl = c->fl;
  for (int q = 0; q < 6; ++q) if (c->C.lwm[q] != 0) pl.any_wm = true;      // (of the case: a face owned by another slab counts)
  // fillps inside the forward x transform: homogeneous pressure BCs (no boundary r.h.s.) and a radix-8 x plan; the transform then also sums the bulk
  // means of the forced components (their increment is only needed by the correction kernel)
  pl.fuse_fill = !fl.unfused_fillps && solver_can_fuse_fillps(c);
  for (int d = 0; d < 3; ++d) pl.fuse_fill = pl.fuse_fill && ((c->C.bcpre[2 * d] == 0. && c->C.bcpre[2 * d + 1] == 0.) || c->C.cbcpre[2 * d] == 'P');
  // periodic x, explicit diffusion, no wall model, the fused passes everywhere: every kernel of the step wraps around instead of reading x ghost
  // columns, which are then left alone until the step returns (common.hpp, step_xskip)
  pl.xskip = !fl.xghosts_in_step && CBP(c, 0, 1) == 'P' && CBP(c, 1, 1) == 'P' && c->C.impdiff == 0 && !fl.unfused_rk && !fl.unfused_correc &&
             pl.fuse_fill && c->xkind == 0 && sgs_wraps_x(c) && !pl.any_wm;
  // dynamic model, x and y periodic (|S|Sij as pair fields), z periodic or two no-slip walls, explicit diffusion, no wall model: the projection
  // u = u* - dtrk grad(pp) (+ the deferred forcing) and p += pp are folded into the strain-rate pass of cmpt_sgs, which reads the velocity anyway --
  // the correction pass (9 words per cell) disappears (dsmag_fast, k_corr_strain_tile)
  pl.fold_correc = pl.xskip && c->C.sgstype == 2 && c->C.impdiff == 0 && dsmag_pairs(c) && !fl.unfolded_correc && !fl.unfused_correc && c->n[0] % 64 == 0;      // (whole 64-cell tiles in x)
  { const bool perz = CBV(c, 0, 3, 3) == 'P' && CBV(c, 1, 3, 3) == 'P';
    bool walls = true;
    for (int iv = 1; iv <= 3; ++iv) for (int sd = 0; sd <= 1; ++sd) walls = walls && CBV(c, sd, 3, iv) == 'D';
    walls = walls && CBP(c, 0, 3) == 'N' && CBP(c, 1, 3) == 'N';
    pl.fold_correc = pl.fold_correc && (perz || walls);
    // several slabs: the pass reaches the companion field of pp with 32-bit offsets (two fields under 4 GB), exchanges through the slab hooks
    // (at least two rows per slab: row 2 goes to the companion field BEFORE the exchange, and with one row per slab "row 2" is the stale ghost row n2+1)
    if (c->P > 1) pl.fold_correc = pl.fold_correc && c->comm.on && c->n[1] >= 2 && 2 * (c->ntot + 2 * LINE_REALS) * sizeof(real) < (1ull << 32);
    // ... with a second ghost row of the prediction and a third of pp (companion fields) the pass forms the ghost rows of all its outputs itself
    pl.fold_rows2 = pl.fold_correc && c->P > 1 && c->vel_comp && c->n[1] >= 4 && 3 * (c->ntot + 2 * LINE_REALS) * sizeof(real) < (1ull << 32) && !fl.unmerged_bc; }
  // no subgrid model, explicit or z-implicit diffusion, no wall model, every direction periodic or between walls with homogeneous Neumann pressure
  // (Taylor-Green, channels, cavities without a model): the projection and pressure update of substeps 1 and 2 are applied by the momentum pass of the
  // NEXT substep while it loads its planes (k_momrk<.., CORR = 1>) -- between the two the fields hold the prediction, whose ghost cells receive the
  // projected values through the corrected view of the ghost-cell kernels. The correction pass (9 words per cell) runs once per step instead of three times.
  pl.fold_mom = !pl.fold_correc && fold_mom_ok(c);
  // ... and the THIRD substep's projection may be left to the next step's first momentum pass (finish_pending for every other entry of the C-ABI). One rank
  // only: on several slabs completing it moves slab rows, which would turn every rank-local entry of the C-ABI into a hidden collective
  pl.lazy_last = pl.fold_mom && !fl.eager_projection && c->P == 1 && (fl.lazy_projection || (size_t)c->n[0] * c->n[1] * c->n[2] >= ((size_t)1 << 22));
  // z-implicit diffusion: the Helmholtz sweeps form their r.h.s. themselves (k_gaussel_cols_rhs); needs the shared-pivot form
  { const char *bz = &c->cbcvel[4];
    pl.defer_imp_rhs = c->C.impdiff == 2 && !fl.helmholtz_z_per_column && !fl.unfused_imp_rhs &&
                       !(bz[0] == 'P' && bz[1] == 'P') && !(bz[6] == 'P' && bz[7] == 'P') && !(bz[12] == 'P' && bz[13] == 'P'); }
  // Wall models: the bounduvw between bulk_forcing and fillps (main.f90:492-494) updates the wall-model planes and sets the tangential ghost cells of the
  // wall-model faces from them -- and nothing reads either before the bounduvw after correc (main.f90:500-501) has rewritten both: fillps differences the
  // normal components, the solver and boundp see pp, correc only adds to the cells. The one exception is a sampling height inside the first cell
  // (index_wm = 1 / n): that second wall-model update then interpolates with the ghost cell the first one left. Everywhere else the first update is
  // skipped (two launches per substep), with results identical to the last bit.
  { bool dead = !wm_samples_ghost(c);
    for (int sd = 0; sd <= 1; ++sd) if (LWM(c, sd, 1) != 0) dead = false;      // (wall-model faces in x: no reference-made vector holds this shortcut to account there)
    // (from the case alone, the same on every rank: the deferred forcing moves the all-reduce of the bulk means; CALES_UNMERGED_BC keeps the reference's
    //  full sequence: the A/B of the tests)
    pl.skip_first_wm = dead && pl.any_wm && !fl.unmerged_bc; }
  // explicit step, forced directions periodic: the velocity between the forcing and the correction is only differenced along the forced direction
  // (fillps) -- the increment is added by the correction kernel, one pass less. With a wall model only where its first update is skipped (above):
  // k_wallmodel would otherwise sample the velocity without the increment
  pl.fuse_cu = !fl.unfused_correc && c->C.impdiff != 1;     // updatep only needs pp: one pass with correc
  { bool ok = c->C.impdiff == 0 && pl.fuse_cu && !fl.unfused_forcing && (!pl.any_wm || pl.skip_first_wm);
    for (int d = 0; d < 3; ++d) if (c->C.is_forced[d]) ok = ok && c->cbcvel[6 * d + 2 * d] == 'P' && c->cbcvel[6 * d + 2 * d + 1] == 'P';
    pl.force_mask = (c->C.is_forced[0] ? 1 : 0) | (c->C.is_forced[1] ? 2 : 0) | (c->C.is_forced[2] ? 4 : 0);
    pl.defer_force = ok && pl.force_mask != 0; }
  pl.mean_mask = (pl.fuse_fill && pl.defer_force && !fl.unfused_mean) ? pl.force_mask : 0;
  pl.keep_last_rhs = fl.keep_last_rhs;
  // no subgrid model and homogeneous sgs BC values: the eddy viscosity is zero, ghost cells included, since start-up (sgs.f90:62-68; the first
  // cmpt_sgs of a context zeroes the whole field)
  pl.visct_ghosts = !(c->C.sgstype == 0 && (c->visct_zero || c->sgs_first));
  for (int q = 0; q < 6; ++q) if (c->C.bcsgs[q] != 0.) pl.visct_ghosts = true;
  c->plan = pl;
}
static const StepPlan &current_plan(cales_ctx *c) {
  const StepPlan &pl = c->plan;
  if (!pl.valid || pl.in_visct_zero != c->visct_zero || pl.in_sgs_first != c->sgs_first || pl.in_comm_on != c->comm.on || pl.in_overlap != (c->comm_stream != nullptr)) make_plan(c);
  return c->plan;
}
int cales_describe_plan(cales_ctx *c, char *buf, int buflen) {
  if (!c || !buf || buflen < 1) return 1;
  const StepPlan &pl = current_plan(c);      // (reads no field: a pending projection stays pending)
  std::string s;
  s += std::string("projection=") + (pl.fold_rows2 ? "in_strain_rate_pass(ghost_rows_local)" : pl.fold_correc ? "in_strain_rate_pass" : pl.fold_mom ? (pl.lazy_last ? "in_next_momentum_pass(all_substeps)" : "in_next_momentum_pass(substeps_1_2)") : (pl.fuse_cu ? "own_pass(correc+updatep)" : "own_passes"));
  s += std::string(";x_ghost_columns=") + (pl.xskip ? "wrapped" : "maintained");
  s += std::string(";fillps=") + (pl.fuse_fill ? "in_x_transform" : "own_pass");
  s += std::string(";bulk_forcing=") + (pl.force_mask == 0 ? "none" : pl.defer_imp_rhs ? "in_helmholtz_sweep" : pl.defer_force ? (pl.mean_mask ? "in_correction(means_in_x_transform)" : "in_correction(means_own_pass)") : "own_pass");
  if (c->C.impdiff == 2) s += std::string(";implicit_rhs=") + (pl.defer_imp_rhs ? "in_helmholtz_sweep" : "own_pass");
  if (pl.any_wm) s += std::string(";first_wall_model_update=") + (pl.skip_first_wm ? "skipped" : "kept");
  s += std::string(";ghost_cells=") + (c->fl.unmerged_bc ? "by_direction" : "one_launch");
  s += std::string(";visct_ghost_cells=") + (pl.visct_ghosts ? "updated" : "zero_field");
  s += std::string(";momentum=") + (c->fl.unfused_rk ? "mom+rk_update" : "fused_mom_rk");
  s += std::string(";sgs=") + sgs_path_name(c);
  s += std::string(";solver=") + solver_path_name(c);
  if (c->P > 1) s += ";mode_columns_per_rank=" + std::to_string(c->nyq_ok ? c->cw_nyq : c->cw);      // of the pressure solve (padded to whole 128-B lines where that costs 6 % or less: solver_setup)
  s += ";ranks=" + std::to_string(c->P) + ";exchanges=" + (c->P == 1 ? "none" : !c->comm.on ? "unset" : (c->comm_stream && (c->comm.halo_s || c->comm.a2a_part)) ? "second_stream" : "in_order");
  std::snprintf(buf, buflen, "%s", s.c_str());
  return (int)s.size() < buflen ? 0 : 2;      // 2: truncated
}

static int step_body(cales_ctx *c, real dt);
int cales_step(cales_ctx *c, real dt) {
  LAUNCHCHK(c);
  const int e = step_body(c, dt);
  LAUNCHCHK(c);
  // an error in the middle of a step leaves the fields between two operators (x ghost columns stale with step_xskip, a half-applied substep): the
  // context is marked failed so that operator-level calls cannot read that state as if it were a finished step
  if (e) { c->launch_err = "an earlier cales_step failed (" + c->err + "): the fields are in an intermediate state, the context is unusable"; return e; }
  return 0;
}
// One time step, src/main.f90:417-508. WHICH form every operator takes is the plan's (make_plan); what is left here is the sequence and the
// hand-over of the plan's decisions to the operators through the context's per-call fields (reset on every return by `reset`).
static int step_body(cales_ctx *c, real dt) {
  static const real rk[3][2] = {{32. / 60., 0.}, {25. / 60., -17. / 60.}, {45. / 60., -25. / 60.}};
  const StepPlan pl = current_plan(c);      // (a copy: the plan of THIS step, whatever the step does to the state it was made from)
  if (c->fold_mom_dtrk != 0. && !pl.fold_mom) { if (int e = finish_pending(c)) return e; }      // (the conditions changed between two steps: a field was set by hand)
  const bool pending_in = c->fold_mom_dtrk != 0.;      // the step before left its last projection to this step's first momentum pass
  LAUNCH(c, k_zero6, dim3(1), dim3(64), 0, c->stream, c->d_force, pending_in ? 3 : 0);     // dpdl(:) = 0
  c->in_step = true;
  struct Reset { cales_ctx *c; bool keep = false; ~Reset() { c->in_step = false; c->step_xskip = false; c->bc_nride = 0; c->fold_dtrk = 0.; c->fold_rows2 = false; if (!keep) { c->fold_mom_dtrk = 0.; c->fold_mom_pdone = false; } c->bc_view_dtrk = 0.; c->defer_force = false; c->defer_imp_rhs = false; c->fuse_fillps_dti = 0.; c->fuse_mean_mask = 0; c->bc_skip_wm = false; c->skip_rhs_store = false; c->defer_halo = false; c->bc_no_halo = false; } } reset{c};      // also on the error returns
  if (c->pend_xrefresh && !pl.xskip) {      // the step before left the x ghost columns stale and this one reads them
    c->pend_xrefresh = false; c->step_xskip = true;
    if (int e = end_of_step_refresh(c)) return e;
  }
  c->step_xskip = pl.xskip;
  for (int irk = 1; irk <= 3; ++irk) {
    const real dtrk = (rk[irk - 1][0] + rk[irk - 1][1]) * dt, dtrki = 1. / dtrk;
    real alpha = 0.;
    c->defer_imp_rhs = pl.defer_imp_rhs;
    c->defer_force = pl.defer_force;
    c->fuse_mean_mask = pl.mean_mask;
    c->skip_rhs_store = irk == 3 && !pl.keep_last_rhs;
    const bool p_ghosts_due = c->fold_mom_dtrk != 0. && !c->fold_mom_pdone;      // the momentum pass below stores p + pp of the interior cells: its ghost cells ride along with those of the prediction
    { const int e = op_rk(c, irk, dt); c->skip_rhs_store = false; if (e) return e; }
    if (int e = op_bulk_forcing(c)) return e;
    if (c->C.impdiff == 2) {
      alpha = -.5 * c->visc * dtrk;
      for (int iv = 1; iv <= 3; ++iv) if (int e = op_helmholtz_z(c, iv, alpha)) return e;
    } else if (c->C.impdiff == 1) {
      alpha = -.5 * c->visc * dtrk;
      for (int iv = 1; iv <= 3; ++iv) if (int e = op_helmholtz(c, iv, alpha)) return e;
    }
    c->defer_imp_rhs = false;
    if (p_ghosts_due && !c->fl.unmerged_bc) { c->bc_nride = 1; c->bc_ride[0] = c->f[CALES_P]; c->bc_ride_which[0] = 0; }
    { c->bc_skip_wm = pl.skip_first_wm;
      // (two ghost rows of the prediction, fold_rows2: the rows 2 / n2-1 travel to the neighbours' companion fields in the same message as the rows 1 / n2)
      c->defer_halo = pl.fold_rows2;
      int e = op_bounduvw(c, c->bcu, c->bcv, c->bcw, 1, 0, c->f[CALES_U], c->f[CALES_V], c->f[CALES_W]);
      if (pl.fold_rows2) {
        if (!e) e = halo_y_rows(c, 3, c->f + CALES_U, 2);
        c->defer_halo = false;
        if (!e) e = halo_flush_deferred(c, false); else { c->deferred.clear(); c->deferred_wide.clear(); }
      }
      c->bc_skip_wm = false;
      const bool rode = p_ghosts_due && !c->fl.unmerged_bc && c->bc_nride == 0; c->bc_nride = 0;
      if (e) return e;
      if (p_ghosts_due && !rode) { if (int e2 = op_boundp(c, c->f[CALES_P], 0)) return e2; } }
    if (pl.fuse_fill) c->fuse_fillps_dti = dtrki;
    else { if (int e = op_fillps(c, dtrki)) return e; if (int e = op_updt_rhs_b(c)) return e; }
    { const int e = op_solver(c); c->fuse_fillps_dti = 0.; if (e) return e; }
    if (pl.fold_rows2) {
      // three ghost rows of pp above (n2+1, and n2+2, n2+3 in the ghost rows n2+1 of its two companions), two below (0, and -1 in the first companion's row 0):
      // ONE exchange, behind the ghost-cell kernel (the rows that travel carry their x and z ghost cells)
      c->defer_halo = true;
      real *one[1] = {c->f[CALES_PP]};
      int e = op_boundp(c, c->f[CALES_PP], 0);
      if (!e) e = halo_y_rows(c, 1, one, 2);
      if (!e) e = halo_y_rows(c, 1, one, 3);
      c->defer_halo = false;
      if (!e) e = halo_flush_deferred(c, false); else { c->deferred.clear(); c->deferred_wide.clear(); }      // (whole rows travel: the x and z ghost cells the kernel above gave them included)
      if (e) return e;
    } else
    if (pl.fold_correc && c->P > 1) {
      // the folded projection corrects v in the ghost row n2+1 too and needs pp one row further out: row 2 of every slab goes to row 1 of pp's companion
      // field and both fields take the ghost-cell update -- ONE exchange; the upper neighbour's row 2 arrives in the companion's ghost row n2+1
      LAUNCH(c, k_row2_to_companion, dim3((c->n[0] + 2 + 63) / 64, (c->n[2] + 2 + 3) / 4), dim3(64, 4), 0, c->stream, c->g, c->f[CALES_PP], c->scr2);
      real *two[2] = {c->f[CALES_PP], c->scr2};
      if (int e = op_boundp_multi(c, 2, two, 0)) return e;
    } else
    if (int e = op_boundp(c, c->f[CALES_PP], 0)) return e;
    if (pl.fold_correc) { c->fold_dtrk = dtrk; c->fold_rows2 = pl.fold_rows2; }      // correc, bounduvw, updatep, boundp(p): inside the cmpt_sgs below (dsmag_fast)
    else if (pl.fold_mom && (irk < 3 || pl.lazy_last)) {
      // the ghost cells of the projected velocity now (through the corrected view), its interior cells and p + pp in the next momentum pass -- the next
      // substep's, or after the third substep the next step's (finish_pending for every other entry of the C-ABI)
      c->fold_mom_fmask = c->defer_force ? pl.force_mask : 0;
      c->defer_force = false;
      if (c->C.impdiff == 2) {      // z-implicit diffusion: the pressure update keeps its own pass (updatep.f90:40-46) -- the z Laplacian of pp has no values in ghost cells
        if (int e = op_updatep(c, alpha)) return e;
        if (int e = op_boundp(c, c->f[CALES_P], 0)) return e;
        c->fold_mom_pdone = true;
      }
      c->bc_view_dtrk = dtrk;
      const int e = op_bounduvw(c, c->bcu, c->bcv, c->bcw, 1, 1, c->f[CALES_U], c->f[CALES_V], c->f[CALES_W]);
      c->bc_view_dtrk = 0.;
      if (e) return e;
      c->fold_mom_dtrk = dtrk;
    } else if (int e = project_now(c, dtrk, alpha)) return e;
    c->visct_bc_done = false;
    { const int e = op_cmpt_sgs(c); c->fold_dtrk = 0.; c->fold_rows2 = false; c->defer_force = false; if (e) return e; }
    if (pl.visct_ghosts && !c->visct_bc_done) { if (int e = op_boundp(c, c->f[CALES_VISCT], 1)) return e; }
    c->visct_bc_done = false;
  }
  if (c->fold_mom_dtrk != 0.) { c->pend_xskip = c->step_xskip; reset.keep = true; }      // the last projection is the next step's (or finish_pending's), the refresh with it
  else if (c->step_xskip && !c->fl.eager_projection) c->pend_xrefresh = true;      // the x ghost columns wait for the first caller that is not the next step (finish_pending)
  else if (int e = end_of_step_refresh(c)) return e;
  c->h_red[40] = dt;
  return 0;
}
int cales_get_dpdl(cales_ctx *c, real dpdl[3]) {
  HIPCHK(c, hipMemcpyAsync(c->h_red + 32, c->d_force, 6 * sizeof(real), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const real dti = 1. / c->h_red[40];
  for (int q = 0; q < 3; ++q) dpdl[q] = -c->h_red[35 + q] * dti;
  return 0;
}

// ------------------------------------------------------------------------------------------ multi-GPU hooks
#define CALES_RES_TAIL 4096
int cales_comm_buffer_doubles(const cales_ctx *c, int64_t *n) {
  const int64_t a2a = 2 * (int64_t)c->P * c->n[2] * c->n[1] * c->cw;                 // [peer][k][jl][mm] complex
  const int64_t halo = 4 * 16 * (int64_t)c->g.s1 * (c->n[2] + 2);                    // lo|hi x up to 16 field planes (send in A, recv in B)
  const int64_t tail = CALES_RES_TAIL + 2 * (int64_t)(c->n[2] + 2);
  *n = std::max(a2a, halo) + tail;
  return 0;
}
int cales_set_comm(cales_ctx *c, cales_halo_cb halo, cales_alltoall_cb a2a, cales_allreduce_cb allred, void *user,
                   real *bufA, real *bufB, int64_t nbuf) {
  int64_t need = 0; cales_comm_buffer_doubles(c, &need);
  if (!halo || !a2a || !allred || !bufA || !bufB || nbuf < need) { c->err = "cales_set_comm: missing callback/buffer or buffers too small"; return 1; }
  c->comm.halo = halo; c->comm.a2a = a2a; c->comm.allred = allred; c->comm.user = user;
  c->comm.A = bufA; c->comm.B = bufB; c->comm.nbuf = nbuf; c->comm.on = true;
  HIPCHK(c, hipMemsetAsync(bufA, 0, nbuf * sizeof(real), c->stream));
  HIPCHK(c, hipMemsetAsync(bufB, 0, nbuf * sizeof(real), c->stream));
  // reduction results live in the tail of A so that the host can all-reduce them in place
  const int64_t tail = CALES_RES_TAIL + 2 * (int64_t)(c->n[2] + 2);
  c->res = bufA + (nbuf - tail);
  if (c->d_p1d && !c->p1d_in_comm) hipFree(c->d_p1d);      // a second call must not free the interior pointer set by the first
  c->d_p1d = c->res + 64;
  c->p1d_in_comm = true;
  return 0;
}
int cales_set_comm_overlap(cales_ctx *c, cales_halo_s_cb halo_s, cales_alltoall_part_cb a2a_part) {
  if (!c->comm.on) { c->err = "cales_set_comm_overlap: call cales_set_comm first"; return 1; }
  if (!c->fl.overlap) return 0;      // default: every exchange in order on the one stream; CALES_OVERLAP=1 opts into the second stream (Flags::read_env)
  if (!c->comm_stream) {      // highest priority: the send/receive kernels of an exchange should get their few workgroups at once, beside a full grid
    int least = 0, greatest = 0;
    HIPCHK(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIPCHK(c, hipStreamCreateWithPriority(&c->comm_stream, hipStreamNonBlocking, greatest));
  }
  c->comm.halo_s = halo_s; c->comm.a2a_part = a2a_part;
  return 0;
}
int cales_initflow_slab(const cales_case *cs, const char *inivel, int is_wallturb, real *u, real *v, real *w, real *p) {
  if (!cs || !inivel || !u || !v || !w || !p) return 1;
  return hs_initflow(cs, inivel, is_wallturb, u, v, w, p, cs->rank, cs->nranks);
}

// ------------------------------------------------------------------------------------------ measurement
int cales_profile_enable(cales_ctx *c, int on) { prof_flush(c); c->prof = on != 0; return 0; }
int cales_profile_reset(cales_ctx *c) { prof_flush(c); c->stats.clear(); return 0; }
int cales_profile_count(cales_ctx *c) { prof_flush(c); return (int)c->stats.size(); }
int cales_profile_get(cales_ctx *c, int idx, char *name, int namelen, int64_t *calls, real *total_ms) {
  prof_flush(c);
  if (idx < 0 || idx >= (int)c->stats.size()) return 1;
  if (name && namelen > 0) std::snprintf(name, namelen, "%s", c->stats[idx].name.c_str());
  if (calls) *calls = c->stats[idx].calls;
  if (total_ms) *total_ms = c->stats[idx].ms;
  return 0;
}
int cales_calibrate(cales_ctx *c, int reps, real gbps[3], int64_t *bytes_per_stream) {
  if (!c || !gbps || reps < 1) return 1;
  ENTER(c);      // (completes a pending projection: the scratch field written below is then dead)
  const long nrows = (long)(c->n[1] + 2) * (c->n[2] + 2);
  const double bytes = (double)nrows * c->n[0] * sizeof(real);
  if (bytes_per_stream) *bytes_per_stream = (int64_t)bytes;
  hipEvent_t e0, e1;
  HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1));
  const dim3 gr(256 * 16), bl(256);
  const real *a = c->f[CALES_U]; real *b = c->scr1;      // scr1: scratch between operators
  auto run = [&](int mode) {
    if (mode == 0) LAUNCH(c, k_calib_rows<0>, gr, bl, 0, c->stream, c->g, nrows, a, b, c->d_red);
    else if (mode == 1) LAUNCH(c, k_calib_rows<1>, gr, bl, 0, c->stream, c->g, nrows, a, b, c->d_red);
    else LAUNCH(c, k_calib_rows<2>, gr, bl, 0, c->stream, c->g, nrows, a, b, c->d_red);
  };
  int rc = 0;
  for (int mode = 0; mode < 3 && !rc; ++mode) {
    run(mode);
    if (hipEventRecord(e0, c->stream) != hipSuccess) { rc = 1; break; }
    for (int r = 0; r < reps; ++r) run(mode);
    if (hipEventRecord(e1, c->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) { rc = 1; break; }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess || !(ms > 0.f)) { rc = 1; break; }
    gbps[mode] = (real)((mode == 2 ? 2. : 1.) * bytes * reps / (ms * 1e-3) / 1e9);
  }
  hipEventDestroy(e0); hipEventDestroy(e1);
  if (rc) { c->err = "cales_calibrate: event timing failed"; return rc; }
  LAUNCHCHK(c);
  return 0;
}
int cales_device_info(cales_ctx *c, char *name, int namelen, int64_t *hbm_bytes) {
  hipDeviceProp_t p; int dev = 0;
  HIPCHK(c, hipGetDevice(&dev)); HIPCHK(c, hipGetDeviceProperties(&p, dev));
  if (name && namelen > 0) std::snprintf(name, namelen, "%s (%s)", p.name, p.gcnArchName);
  if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
  return 0;
}

}  // extern "C"
