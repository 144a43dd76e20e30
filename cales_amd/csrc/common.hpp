// Internal definitions shared by the HIP translation units of libcales_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include "../../include/cales.h"

// The working precision rp of the reference (src/precision.f90:11-20): FP64, or FP32 when the library is built with -DCALES_SINGLE
// (the reference's -D_SINGLE_PRECISION; libcales_hip_sp.so). `real` is cales_real of include/cales.h.
typedef cales_real real;
#ifdef CALES_SINGLE
typedef float2 real2;
#define CALES_EPS 1.1920929e-07f            /* epsilon(1._rp), reference src/param.f90:20 */
#define CALES_BIG 3.4028235e38f             /* huge(1._rp), src/param.f90:25 */
#else
typedef double2 real2;
#define CALES_EPS 2.220446049250313e-16     /* epsilon(1._rp), reference src/param.f90:20 */
#define CALES_BIG 1.7976931348623157e308    /* huge(1._rp), src/param.f90:25 */
#endif
constexpr int RSZ = (int)sizeof(real);          // bytes per element: the byte offsets of ldb/stb
constexpr int LINE_REALS = 128 / RSZ;           // elements per 128-B cache line (16 / 32): row pitch and field offset are laid out in lines
__host__ __device__ inline real2 make_real2(real x, real y) { real2 r; r.x = x; r.y = y; return r; }

struct Geom {              // passed by value to kernels
  int n1, n2, n3;          // local interior sizes
  int s1;                  // row pitch >= n1+3, a multiple of 16 doubles
  long s12;                // s1*(n2+2)
  int jlo;                 // global index offset of local row j=1 is jlo+1 (y-slab)
  int ng2;                 // global n2

  __host__ __device__ inline size_t ix(int i, int j, int k) const { return (size_t)i + (size_t)s1 * (size_t)j + (size_t)s12 * (size_t)k; }
};

struct DBound { real *x, *y, *z; };
// Boundary term of an inhomogeneous z condition in the r.h.s. of a velocity Helmholtz solve (cmpt_rhs_b / bc_rhs, bound.f90:447-560, times alpha, main.f90:432):
// one side of one component; `at` evaluates it for column (i, j) with the operations of the reference in their order. Used by k_rhs_b_velz (the term as a
// pass of its own / as a plane) and by the in-LDS Helmholtz sweep, which adds it while it loads plane 1 / n.
struct RhsBz {
  const real *bc; char ctype, cf; real dlc, dlf, sgn, alpha;
  __host__ __device__ inline real at(int i, int j, int n1) const {
    const real bcv = bc[i + (size_t)(n1 + 2) * j];
    real r = 0.;
    if (cf == 'c') { if (ctype == 'D') r = -2. * bcv / dlc / dlf; else if (ctype == 'N') r = sgn * bcv / dlf; }
    else           { if (ctype == 'D') r = -bcv / dlc / dlf;      else if (ctype == 'N') r = sgn * bcv / dlc; }
    return r * alpha;
  }
};   // device BC planes (0:na+1,0:nb+1,0:1), reference src/typedef.f90:10-14

// Communication hooks (include/cales.h): y-slab neighbours, slab<->mode-block all-to-all, all-reduce.
struct Comm {
  cales_halo_cb halo = nullptr; cales_alltoall_cb a2a = nullptr; cales_allreduce_cb allred = nullptr; void *user = nullptr;
  real *A = nullptr, *B = nullptr; int64_t nbuf = 0;   // two device staging buffers owned by the host (doubles)
  bool on = false;
  cales_halo_s_cb halo_s = nullptr; cales_alltoall_part_cb a2a_part = nullptr;   // exchanges on the second stream (cales_set_comm_overlap)
};

// Where the spectral (after the x transform) data of the Poisson solve lives.
//  blocked = 0: in place in the haloed array: complex (m, j, k) at (ix(0,j,k)>>1) + m
//  blocked = 1: staging buffer laid out [peer][k][jl][mm] (complex), peer block = n3*n2l*cw:
//               on the slab side peer = m/cw, on the mode-block side peer = (j-1)/n2l.
struct Spec {
  int blocked, cw, n2l, n3;
  int nyq = 0;      // periodic x (periodic or Neumann y): the two REAL modes 0 and n1/2 of a row share complex slot 0 (re, im) and the row has n1/2 complex slots -- whole 128-B lines (k_solver.hip, "Nyquist packing")
  // slab side: global mode m, LOCAL row jl (1-based), plane k (1-based)
  __host__ __device__ inline size_t at_slab(const Geom &g, int m, int jl, int k) const {
    if (!blocked) return (g.ix(0, jl, k) >> 1) + (size_t)m;
    const int peer = m / cw, mm = m - peer * cw;
    return (size_t)mm + (size_t)cw * ((size_t)(jl - 1) + (size_t)n2l * ((size_t)(k - 1) + (size_t)n3 * peer));
  }
  // mode-block side: LOCAL mode mm, GLOBAL row j (1-based), plane k (1-based)
  __host__ __device__ inline size_t at_mode(const Geom &g, int mm, int j, int k) const {
    if (!blocked) return (g.ix(0, j, k) >> 1) + (size_t)mm;
    const int peer = (j - 1) / n2l, jl = (j - 1) - peer * n2l;
    return (size_t)mm + (size_t)cw * ((size_t)jl + (size_t)n2l * ((size_t)(k - 1) + (size_t)n3 * peer));
  }
};

// Run-time switches: read from the environment ONCE, by cales_create; the launch path only looks at these fields.
struct Flags {
  bool unfolded_correc = false, unfolded_mom = false, eager_projection = false, lazy_projection = false, helmholtz_z_per_column = false, unfused_imp_rhs = false, unfused_correc = false, unfused_forcing = false, unfused_fillps = false, unfused_mean = false, keep_last_rhs = false, wide_offsets = false, dsmag_reference_sequence = false, dsmag_xghosts = false, smag_reference_sequence = false, gaussel_march = false, fft_generic = false, keep_null_mode = false, unfused_rk = false, overlap = false, xghosts_in_step = false, unmerged_bc = false, no_nyquist_packing = false;
  int kchunk = 0; long tile_min_blocks = 2048;
  std::string test_bad_launch;      // CALES_TEST_BAD_LAUNCH: test hook of the launch check (LAUNCH below)
  void read_env() {
    test_bad_launch = getenv("CALES_TEST_BAD_LAUNCH") ? getenv("CALES_TEST_BAD_LAUNCH") : "";
    unfolded_mom = getenv("CALES_UNFOLDED_MOM") != nullptr;      // no subgrid model in cales_step: the projection as a pass of its own (k_correc_cell) in every substep instead of inside the next momentum pass
    lazy_projection = getenv("CALES_LAZY_PROJECTION") != nullptr;      // ... the third substep's projection left to the next step on grids of any size (default: one rank with 4M cells or more -- measured: 1024^3 166.0 -> 161.5 ms/step, 64^3 0.235 -> 0.242)
    eager_projection = getenv("CALES_EAGER_PROJECTION") != nullptr;      // ... folded, but the third substep's projection done before cales_step returns instead of by the next step's first momentum pass (or the first call that looks at the fields)
    unfolded_correc = getenv("CALES_UNFOLDED_CORREC") != nullptr;      // dynamic model in cales_step: the projection as a pass of its own (k_correc_cell) instead of inside the strain-rate pass
    helmholtz_z_per_column = getenv("CALES_HELMHOLTZ_Z_PER_COLUMN") != nullptr;
    unfused_imp_rhs = getenv("CALES_UNFUSED_IMP_RHS") != nullptr;
    unfused_correc = getenv("CALES_UNFUSED_CORREC") != nullptr;
    unfused_forcing = getenv("CALES_UNFUSED_FORCING") != nullptr;
    unfused_fillps = getenv("CALES_UNFUSED_FILLPS") != nullptr;
    unfused_mean = getenv("CALES_UNFUSED_MEAN") != nullptr;
    keep_last_rhs = getenv("CALES_KEEP_LAST_RHS") != nullptr;
    wide_offsets = getenv("CALES_WIDE_OFFSETS") != nullptr;
    dsmag_reference_sequence = getenv("CALES_DSMAG_REFERENCE_SEQUENCE") != nullptr;
    dsmag_xghosts = getenv("CALES_DSMAG_XGHOSTS") != nullptr;
    smag_reference_sequence = getenv("CALES_SMAG_REFERENCE_SEQUENCE") != nullptr;
    gaussel_march = getenv("CALES_GAUSSEL_MARCH") != nullptr;
    no_nyquist_packing = getenv("CALES_NO_NYQUIST_PACKING") != nullptr;      // periodic x, periodic or Neumann y: the real modes 0 and n1/2 in columns of their own (n1/2 + 1 mode columns) instead of sharing column 0
    fft_generic = getenv("CALES_FFT_GENERIC") != nullptr;
    xghosts_in_step = getenv("CALES_XGHOSTS_IN_STEP") != nullptr;      // keep the x ghost columns up to date after every operator of cales_step
    keep_null_mode = getenv("CALES_KEEP_NULL_MODE") != nullptr;
    unfused_rk = getenv("CALES_UNFUSED_RK") != nullptr;
    // exchanges beside the kernels on a second stream: opt-in (CALES_OVERLAP=1) until a run with real peers has confirmed it -- the emulated
    // ranks of the tests cannot show a gain, and the in-order exchanges are the form with the fewest assumptions (CALES_NO_OVERLAP wins)
    overlap = getenv("CALES_OVERLAP") != nullptr && atoi(getenv("CALES_OVERLAP")) != 0 && getenv("CALES_NO_OVERLAP") == nullptr;
    unmerged_bc = getenv("CALES_UNMERGED_BC") != nullptr;
    kchunk = getenv("CALES_KCHUNK") ? atoi(getenv("CALES_KCHUNK")) : 0;
    tile_min_blocks = getenv("CALES_TILE_MIN_BLOCKS") ? atol(getenv("CALES_TILE_MIN_BLOCKS")) : 2048;
  }
};

// The path ONE cales_step takes through the fused / folded forms of its operators (reference sequence src/main.f90:417-508): decided once by
// make_plan (api.hip) from the case, the switches (Flags) and the few pieces of state listed under `in_*`, re-made only when one of those changes, and
// READ -- never re-derived -- by step_body. cales_describe_plan (include/cales.h) prints it: bench.py puts the string into its line (config.path), the
// golden tests assert it next to the profile counters.
struct StepPlan {
  bool valid = false;
  bool in_visct_zero = false, in_sgs_first = false, in_comm_on = false, in_overlap = false;      // the state the plan was made from
  bool xskip = false;            // periodic x: the step's kernels wrap around, the x ghost columns are left alone until somebody else reads them
  bool fold_correc = false;      // dynamic model: projection + pressure update inside the strain-rate pass of the substep's cmpt_sgs (k_corr_strain_tile)
  bool fold_rows2 = false;       // ... on several slabs with TWO ghost rows of the prediction (three of pp): the pass also forms the ghost rows of everything it writes,
                                 // so no row of u, v, w, p, |S|, |S|Sij, the filtered velocity or v_c travels after it (27 field planes per step instead of 48)
  bool fold_mom = false;         // no subgrid model: projection of substeps 1, 2 (and 3: lazy_last) inside the NEXT momentum pass (k_momrk<CORR>)
  bool lazy_last = false;        // ... the third substep's projection stays pending across the return of cales_step (one rank only)
  bool defer_imp_rhs = false;    // z-implicit: the Helmholtz sweeps form their r.h.s. (implicit part of rk, forcing, boundary terms) while loading
  bool any_wm = false, skip_first_wm = false;   // wall model: the update between bulk_forcing and fillps is dead work (main.f90:492-501) and skipped
  bool fuse_cu = false;          // correc + updatep in one pass
  bool defer_force = false;      // bulk-forcing increment added by the correction pass (means summed by the forward x transform: mean_mask)
  bool fuse_fill = false;        // fillps inside the forward x transform of the pressure solve
  bool visct_ghosts = false;     // the eddy viscosity's ghost cells are updated after cmpt_sgs (not needed while the field is identically zero)
  bool keep_last_rhs = false;
  int mean_mask = 0, force_mask = 0;
};

struct KernelStat { std::string name; int64_t calls = 0; real ms = 0.; };

struct cales_ctx {
  cales_case C;
  Flags fl;
  StepPlan plan;      // see StepPlan above
  Geom g;
  int n[3], lo[3];
  real dl[3], dli[3], visc;
  hipStream_t stream; bool own_stream;
  hipStream_t comm_stream = nullptr;      // exchanges that overlap kernels (created by cales_set_comm_overlap)
  std::vector<hipEvent_t> sync_ev; size_t sync_next = 0;      // ordering events between the two streams (no timing), reused round-robin
  std::string err;
  std::string launch_err;      // first failed kernel launch / attribute call (LAUNCH below): the context is failed from then on
  // host copies of the grid
  std::vector<real> dzc, dzf, zc, zf, dzci, dzfi, gvr_c, gvr_f;
  char cbcvel[18];
  int is_bound[6], index_wm[6];
  // device grid (0:n3+1)
  real *d_dzc, *d_dzf, *d_zc, *d_zf, *d_dzci, *d_dzfi, *d_gvr_c, *d_gvr_f;
  // fields
  real *f[CALES_NFIELDS];
  real *f2[3] = {nullptr, nullptr, nullptr};   // second velocity buffers of the fused mom+RK kernel (pointers are swapped with f[U..W])
  size_t ntot;
  // BC planes
  DBound bcu, bcv, bcw, bcp, bcs, bcuf, bcvf, bcwf, bcu_mag, bcv_mag, bcw_mag;
  real *rhsbp[3];        // (na,nb,0:1)
  real *rhsbz_vel;       // scratch (n1,n2,0:1) for z-implicit Helmholtz r.h.s.
  // solver
  real *d_lamx, *d_lamy;     // eigenvalues per stored spectral index
  real *d_a, *d_b, *d_c;     // tridiagonal (n3)
  real *d_av[3], *d_bv[3], *d_cv[3];
  real normfft;
  int xkind, ykind;            // 0: periodic (r2c / c2c), 1: Neumann-Neumann cell-centred (DCT-II/III)
  bool nyq_ok = false; int cw_nyq = 0;      // the pressure solve packs the modes 0 and n1/2 into one column (solver_setup: periodic x and y, walls in z, radix-8 transforms, the z tile); mode columns per rank then
  real *d_twx, *d_twy;       // twiddle tables
  real *d_twx_post, *d_twy_post;    // d_twy_post: DCT weights of the x direction
  real *scr_twyd = nullptr;           // DCT weights of the y direction
  real *scr1, *scr2;         // solver scratch (haloed size)
  // reductions
  real *d_red; real *h_red;       // partial sums / results (pinned host)
  real *d_force;                    // f(3) + dpdl(3) accumulators on device
  int red_blocks;
  // sgs scratch
  real *ss2[3] = {nullptr, nullptr, nullptr};      // |S|Sij as three pair fields (2 ntot reals each; dsmag_pairs, k_sgs.hip) instead of sij / mij
  real *s0, *wk[6], *sij[6], *mij[6], *uc, *vc, *wc, *uf, *vf, *wf, *alph2, *d_p1d;
  real is_wall[6];
  bool sgs_first;
  // decomposition
  int P = 1, rank = 0; bool per_y = true; int cw = 0;      // cw: complex mode columns per rank (padded)
  Comm comm;
  bool p1d_in_comm = false;
  real *res = nullptr;                // reduction results (inside comm.A when comm is on, so they can be all-reduced)
  // profiling
  bool prof = false;
  std::vector<KernelStat> stats;
  std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> pending;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> evpool;
  real *d_tw4x = nullptr;  // DCT-IV weights of the x transform (pressure ND / DN)
  real *d_tw4y = nullptr, *d_twy4 = nullptr;      // the same for y, and the twiddles of its N/2-point lines
  real *d_del = nullptr;   // Smagorinsky filter width per plane (fast path)
  void *native_comm = nullptr;   // RCCL communicator + staging buffers when the library does the exchanges itself (comm_rccl.cpp)
  bool visct_zero = true;  // CALES_VISCT still holds the zeros it was created / reset with (no SGS model: lets kernels skip it)
  // dynamic model, fast path: the eddy-viscosity field holds |S| and d_cs(0:n3+1) the clipped plane coefficients <LM>/<MM> until somebody
  // other than the fused momentum kernel reads it (materialize_visct); visct = |S| * cs(k) is the same product either way
  bool visct_lazy = false; real *d_cs = nullptr;
  void *cur_velset = nullptr;       // k_solver.hip: transform set of the velocity component being solved by op_helmholtz
  bool in_step = false;             // inside cales_step: the operator order is known, dead ghost work can be dropped
  bool skip_rhs_store = false;      // cales_step, third substep: see MomRkArgs::wr_new
  real *d_stat2 = nullptr;
  real *d_stat = nullptr;      // partial sums and result of the plane statistics
  bool abct_ready = false, force_zeroed = false;
  real *d_abct = nullptr;      // tridiagonal coefficients in the chunked order of k_gaussel_tile
  // z-only Helmholtz sweeps: the chunked tables of (ivel, alpha) pairs already seen -- three alphas per step, the same every step while dt stays (four slots per
  // component, round robin); a hit saves the scaling and the table kernel of that sweep
  struct HzTab { real alpha = 0.; int nz = 0; bool ok = false; } hz_tab[3][4]; int hz_next[3] = {0, 0, 0}; real *d_hztab = nullptr;
  int ncu = 0;      // compute units of the device (balanced_kchunk)
  int fuse_mean_mask = 0; real *d_mpart = nullptr; size_t n_mpart = 0;      // bulk means of the forced components are summed by that pass too
  real fuse_fillps_dti = 0.;   // != 0: the forward x transform of the next pressure solve forms pp = div(u*)/dtrk itself (cales_step)
  bool defer_force = false;      // explicit step, forced directions periodic, no wall model: u += f is applied by the correction kernel
  bool defer_imp_rhs = false; real hf12 = 0.;   // z-implicit step: u -= hf12*dudtd and u += f are applied inside the Helmholtz sweep
  bool bc_no_halo = false;      // ghost-cell operators skip the slab exchange (the ghost rows are up to date)
  bool visct_bc_done = false;   // cmpt_sgs has already updated the ghost cells of the eddy-viscosity field (dsmag, lazy form: with the scratch fields' exchange)
  bool defer_halo = false; std::vector<real *> deferred; std::vector<unsigned char> deferred_wide;      // (wide: the field is a pair field, rows twice as long)      // y-halo exchanges collected for halo_flush_deferred (k_bound.hip)
  // cell-centred fields whose ghost-cell update rides along with the next bounduvw that takes the one-launch path (cales_step: the pressure after the
  // fused correction + pressure update; p, pp and the eddy viscosity at the end of the step); bounduvw clears the count when it has taken them
  int bc_nride = 0; real *bc_ride[4] = {nullptr, nullptr, nullptr, nullptr}; int bc_ride_which[4] = {0, 0, 0, 0};
  int bc_skip = 0;         // bit d-1: boundp/bounduvw leave direction d alone (set around calls whose consumers do not need it)
  bool bc_skip_wm = false; // op_bounduvw leaves the wall-model update and the tangential ghost cells of wall-model faces alone (cales_step, see step_body)
  // cales_step with periodic x: the x ghost columns are not maintained between the operators of a step -- every kernel of the step reads the wrapped
  // interior column instead (a ghost-column update touches two cache lines per row and field for two values: 1.2 of 45 ms per step at 512^3) -- and
  // are brought up to date once, when the step returns
  bool step_xskip = false;
  // cales_step, dynamic model on one rank with x and y periodic: the velocity correction and the pressure update of the substep are done by the
  // strain-rate pass of the cmpt_sgs that follows (k_strain_tile<.., CORR = 1>, k_sgs.hip) -- != 0: the dtrk of the pending projection
  real fold_dtrk = 0.; bool fold_rows2 = false;      // (fold_rows2: with two ghost rows of the prediction, StepPlan)
  // cales_step without subgrid model (explicit diffusion, one rank, every direction periodic or between no-slip walls with Neumann pressure): the
  // projection of substeps 1 and 2 is applied by the momentum pass of the NEXT substep while it loads its planes (k_momrk<.., CORR = 1>); the ghost
  // cells of the prediction receive their final values through a corrected view in the ghost-cell kernels (bc_view_dtrk). != 0: the dtrk of the
  // pending projection, with the mask of the components whose bulk-forcing increment it adds
  real fold_mom_dtrk = 0.; int fold_mom_fmask = 0;
  bool fold_mom_pdone = false;      // z-implicit diffusion: the pressure update ran as a pass of its own (its z Laplacian of pp cannot be formed in ghost cells), only the velocity is pending
  // The third substep's projection stays pending ACROSS the return of cales_step (fold_mom_dtrk != 0 outside a step): the next step's first momentum
  // pass applies it, or -- finish_pending in api.hip -- the first other entry of the C-ABI that reads or writes a field (every one of them calls it,
  // cales_sync included: a caller never sees the prediction). pend_xskip: the x ghost columns were left alone by that step.
  bool pend_xskip = false;
  // cales_step with step_xskip returns with the x ghost columns stale and THIS set: the next cales_step does not read them, every other entry of the C-ABI
  // brings them up to date first (finish_pending; local copies, no exchange: safe on several slabs) -- the refresh is 0.2 ms of strided accesses at 512^3
  bool pend_xrefresh = false;
  real bc_view_dtrk = 0.;      // op_bounduvw: sources are read as (u* + f) - dtrk grad(pp) wherever they are interior cells
  size_t pp_companion_bytes = 0;      // scr2 sits this many bytes behind CALES_PP in one allocation (api.hip field_alloc_multi)
  // several slabs with the dynamic model: u, v, w (both buffer sets) and pp carry COMPANION fields right behind them in their allocations (the same
  // distance for all: comp_one reals) -- the second ghost rows of the folded strain-rate pass (rows -1 and n2+2 in the companion's ghost rows 0 and
  // n2+1; pp a second companion for row n2+3), exchanged from the rows 2 / n2-1 (3) of the neighbours (k_bound.hip, halo kinds 2 and 3)
  bool vel_comp = false; size_t comp_one = 0; real *scr3 = nullptr;
  int field_ofs = 0;       // doubles between a field's allocation and its element (0,0,0)
  real *d_nullw = nullptr; // work space of k_null_column (CALES_KEEP_NULL_MODE)
};

#define CBV(c, side, dir, vel) ((c)->cbcvel[(side) + 2 * ((dir) - 1) + 6 * ((vel) - 1)])
#define CBP(c, side, dir) ((c)->C.cbcpre[(side) + 2 * ((dir) - 1)])
#define CBS(c, side, dir) ((c)->C.cbcsgs[(side) + 2 * ((dir) - 1)])
#define LWM(c, side, dir) ((c)->C.lwm[(side) + 2 * ((dir) - 1)])
#define ISB(c, side, dir) ((c)->is_bound[(side) + 2 * ((dir) - 1)])
#define IWM(c, side, dir) ((c)->index_wm[(side) + 2 * ((dir) - 1)])

#define HIPCHK(ctx, call)                                                                          \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess) {                                                                        \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                              \
      return 1;                                                                                    \
    }                                                                                              \
  } while (0)

// ---- every kernel launch is checked (the reference ignores its istat everywhere, src/solver_gpu.f90:80).
// LAUNCH = hipLaunchKernelGGL + hipGetLastError: an invalid configuration (block size, LDS size, too many registers for the block) is caught at the
// launch site. The FIRST failure is recorded in c->launch_err with the kernel's name and stays there: the context is failed, LAUNCHCHK at the end
// of every operator and cales_* entry returns non-zero with that message (cales_last_error), and every later call on the context refuses to run --
// never a silently wrong field. Void helpers and lambdas record through the same path (HIPSOFT for other HIP calls whose status must not be lost).
// Test hook: CALES_TEST_BAD_LAUNCH=<substring of a kernel name> gives the matching launches a block of 4096 threads (invalid on every device).
void launch_failed(cales_ctx *c, const char *what, hipError_t e);
#define LAUNCH(c, kernel, grid, block, lds, stream, ...)                                           \
  do {                                                                                             \
    dim3 blk_ = (block);                                                                           \
    if (!(c)->fl.test_bad_launch.empty() && strstr(#kernel, (c)->fl.test_bad_launch.c_str())) blk_ = dim3(4096, 1, 1); \
    hipLaunchKernelGGL(kernel, grid, blk_, lds, stream, __VA_ARGS__);                              \
    const hipError_t le_ = hipGetLastError();                                                      \
    if (le_ != hipSuccess) launch_failed(c, #kernel, le_);                                         \
  } while (0)
#define LAUNCHCHK(c)                                                                               \
  do {                                                                                             \
    if (!(c)->launch_err.empty()) { (c)->err = (c)->launch_err; return 1; }                        \
  } while (0)
#define HIPSOFT(c, call)                                                                           \
  do {                                                                                             \
    const hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) launch_failed(c, #call, e_);                                             \
  } while (0)

// profiling bracket: PROF_BEGIN(ctx,"name"); launch...; PROF_END(ctx)
int  prof_begin(cales_ctx *c, const char *name, hipStream_t s = nullptr);
void prof_end(cales_ctx *c, int slot, hipStream_t s = nullptr);
// `later` waits for everything queued on `earlier` so far
int  stream_after(cales_ctx *c, hipStream_t later, hipStream_t earlier);
void prof_flush(cales_ctx *c);
struct ProfScope {
  cales_ctx *c; int slot;
  hipStream_t s;
  ProfScope(cales_ctx *c_, const char *name, hipStream_t s_ = nullptr) : c(c_), slot(c_->prof ? prof_begin(c_, name, s_) : -1), s(s_) {}
  ~ProfScope() { if (slot >= 0) prof_end(c, slot, s); }
};

// ---- host-side set-up (host_setup.cpp)
void   hs_initgrid(int gtype, int n, real gr, real lz, real *dzc, real *dzf, real *zc, real *zf);
void   hs_initbc(cales_ctx *c, std::vector<real> hb[11][3]);
void   hs_eigenvalues(int n, const char *cbc2, char c_or_f, real *lambda);
void   hs_tridmatrix(const char *cbc2, int n, const real *dzci, const real *dzfi, char c_or_f, real *a, real *b, real *c);
int    hs_initflow(const cales_case *cs, const char *inivel, int is_wallturb, real *u, real *v, real *w, real *p, int rank, int nranks);
int    hs_check_case(const cales_case *cs, std::string &msg);
void   hs_bc_rhs(const char *cbc2, const real *bc, int na, int nb, const real *dlc, const real *dlf, char c_or_f, real *rhs);

// ---- device operators (k_*.hip); all asynchronous on c->stream
int op_bounduvw(cales_ctx *c, DBound &bu, DBound &bv, DBound &bw, int is_updt_wm, int is_correc, real *u, real *v, real *w);
int op_boundp(cales_ctx *c, real *p, int which);
int op_boundp_multi(cales_ctx *c, int nf, real **p, int which);
int halo_flush_deferred(cales_ctx *c, bool overlapped = true);
int halo_y_rows(cales_ctx *c, int nf, real **flds, int kind);      // kind 2 / 3: rows 2, n2-1 (3, n2-2) of the neighbours into the ghost rows of the fields' first (second) companions
int op_mom(cales_ctx *c);
int op_rk(cales_ctx *c, int irk, real dt);
int op_rk_par(cales_ctx *c, real rkpar1, real rkpar2, real dt);
int op_momrk(cales_ctx *c, real f1, real f2, real f12);
int op_bulk_forcing(cales_ctx *c);
int op_bulk_mean_dev(cales_ctx *c, const real *p, int c_or_f, real *d_out);   // result to device scalar
int op_fillps(cales_ctx *c, real dtrki);
int op_updt_rhs_b(cales_ctx *c);
int op_solver(cales_ctx *c);
int op_helmholtz_z(cales_ctx *c, int ivel, real alpha);
extern "C" void cales_comm_release_native(cales_ctx *c);
int op_helmholtz(cales_ctx *c, int ivel, real alpha);
int op_correc(cales_ctx *c, real dtrk);
int materialize_visct(cales_ctx *c);
int op_stats_chan(cales_ctx *c, real *buf);
int op_stats_chan_budget(cales_ctx *c, real *budget, real *leak);
int op_out1d(cales_ctx *c, int field, int idir, int use_dzc, real *buf);
int op_out1d_chan(cales_ctx *c, real *buf);
int op_out2d_duct(cales_ctx *c, real *buf);
bool solver_can_fuse_fillps(cales_ctx *c);
std::string solver_path_name(cales_ctx *c);      // which transform / tridiagonal kernels the pressure solve of this context takes (cales_describe_plan)
const char *sgs_path_name(const cales_ctx *c);      // likewise for cmpt_sgs
int op_force_from_partials(cales_ctx *c, int mask, const real *part, int nblk);
int op_correc_updatep(cales_ctx *c, real dtrk, real alpha, int upd);
int op_updatep(cales_ctx *c, real alpha);
int op_cmpt_sgs(cales_ctx *c);
bool dsmag_pairs(const cales_ctx *c);
int op_boundp_wide(cales_ctx *c, int nf, real **p2, int which);      // ghost cells of pair fields (y, z; x periodic and wrapped by the consumers)
int op_xwrap_zghost(cales_ctx *c, int nf, real **f);      // periodic copy of the x ghost columns on the planes k = 0 and n3+1
bool sgs_wraps_x(const cales_ctx *c);      // the SGS pass of this case reads wrapped interior columns instead of x ghost columns
int op_chkdt(cales_ctx *c, real *dtmax);
int op_chkdiv(cales_ctx *c, real *divtot, real *divmax);
int solver_setup(cales_ctx *c);
void solver_teardown(cales_ctx *c);

// XCD-aware block order for the one-plane-per-block stencil kernels (grid = x tiles, y tiles, z planes).
// Workgroups are dealt round-robin to the 8 XCDs, each with its private 4 MiB L2 (MI355X_MICROARCH.md, "Workgroup
// dispatch"). The remap lets every XCD walk a fixed (x tile, group of 8 y-adjacent tiles) column through k, so the k+-1
// planes and j+-1 rows a block needs were fetched moments earlier by blocks of the SAME XCD and are L2 hits instead
// of HBM re-reads. Only the order changes (a bijection on block ids); results are identical.
#ifdef __HIPCC__

// ---- global accesses as base pointer (kernel argument, scalar registers) + BYTE offset. With OFF = unsigned the compiler
// emits the saddr + 32-bit voffset form: one VGPR per access stream instead of a 64-bit address per field (tile kernels
// stream up to 17 fields). Hosts pick OFF = unsigned when a field is smaller than 4 GB, size_t otherwise.
// element k of a small read-only coefficient array (grid spacings, zc, ...) through the scalar cache: the constant address space makes the
// load an s_load (lgkmcnt) whatever stores the kernel makes, so waiting for it never drains the vector loads in flight (a global_load of the
// same value costs an s_waitcnt vmcnt(0) in the middle of a prefetch)
// (in the LDS-heavy last dsmag pass the same change measured 7 % slower -- scalar loads share lgkmcnt with the LDS reads and return
// out of order, so waiting for one drains the LDS queue -- and that kernel keeps its vector loads)
// ONLY for data no kernel writes while a kernel reads it through here: the scalar cache is not coherent with vector stores inside a launch (it is
// invalidated at kernel boundaries). The grid tables never change; the plane coefficients d_cs of the dynamic model are written by k_dsmag_coef and
// read through ldc by LATER launches only (k_momrk) -- a kernel that both wrote and read them would see stale values.
__device__ inline real ldc(const real *p, int k) { return ((const __attribute__((address_space(4))) real *)p)[k]; }
template <typename OFF> __device__ inline real ldb(const real *b, OFF o) { return *(const real *)((const char *)b + o); }
template <typename OFF> __device__ inline void stb(real *b, OFF o, real v) { *(real *)((char *)b + o) = v; }

// ---- cross-lane moves on the vector ALU (DPP) instead of ds_bpermute: no LDS-pipe traffic, short latency ----
template <int CTRL, int ROWMASK = 0xf>
__device__ inline real dpp_f64(real v) {       // lanes without a source receive 0 (bound_ctrl); rows masked out keep v itself
#ifdef CALES_SINGLE
  const int w = __float_as_int(v);
  return __int_as_float(__builtin_amdgcn_update_dpp(w, w, CTRL, ROWMASK, 0xf, true));
#else
  int lo = __double2loint(v), hi = __double2hiint(v);
  // old = the source register and bound_ctrl: no register has to be zeroed before every move (the old form cost one v_mov per dpp move)
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROWMASK, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROWMASK, 0xf, true);
  return __hiloint2double(hi, lo);
#endif
}
// the value of one lane (a compile-time lane number) in all lanes: v_readlane_b32 through a scalar register
template <int LANE>
__device__ inline real lane_bcast(real v) {
#ifdef CALES_SINGLE
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), LANE));
#else
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), LANE), __builtin_amdgcn_readlane(__double2loint(v), LANE));
#endif
}
__device__ inline real lane_prev(real v) { return dpp_f64<0x138>(v); }   // wave_shr:1, lane i <- lane i-1 (lane 0 <- 0)
__device__ inline real lane_next(real v) { return dpp_f64<0x130>(v); }   // wave_shl:1, lane i <- lane i+1 (lane 63 <- 0)
// sum over the 64 lanes, valid in lane 63 ONLY (row_shr 1,2,4,8 then row_bcast 15 and 31; the rows the two broadcasts mask out
// add their own value to themselves, so every other lane ends with garbage -- callers must consume the result in lane 63 and nowhere else)
__device__ inline real wave_sum_lane63(real s) {
  s += dpp_f64<0x111>(s); s += dpp_f64<0x112>(s); s += dpp_f64<0x114>(s); s += dpp_f64<0x118>(s);
  s += dpp_f64<0x142, 0xa>(s); s += dpp_f64<0x143, 0xc>(s);
  return s;
}

// Block -> tile map of the plane-marching tile kernels, launched as a 1-D grid: consecutive blocks go to the eight XCDs in turn, and each XCD
// takes bands of `sub` consecutive y tiles with all their x tiles and k chunks (x fastest, then y inside the band, then k), so that the halo
// rows and columns two neighbouring tiles both read meet in one XCD's L2 instead of being fetched over the fabric twice. With the plain 3-D
// grid every y neighbour sits on another XCD. The grid is padded to whole bands; blocks of the padding return at once (false).
struct BandMap { int gx, gy, gz, sub; };
static inline BandMap band_map(int gx, int gy, int gz) { BandMap m{gx, gy, gz, 1}; m.sub = gy >= 64 ? 8 : (gy + 7) / 8; if (m.sub < 1) m.sub = 1; return m; }
// The plain 3-D grid already gives every XCD fixed x-tile columns (y neighbours in one L2) when the number of x tiles divides 8 or is a multiple
// of it -- the 64-wide tiles of power-of-two grids: there the bands bring nothing (momentum pass 8.0 -> 8.6 ms/step at 512^3, measured) and
// are not used; the 62-wide tiles (9 per 512 cells: the XCD of a tile then runs along diagonals) and odd sizes take the bands.
static inline bool band_wanted(int gx) { return !(gx % 8 == 0 || 8 % gx == 0); }
static inline unsigned band_blocks(const BandMap &m) { return 8u * m.gx * m.sub * m.gz * ((m.gy + 8 * m.sub - 1) / (8 * m.sub)); }
__device__ inline bool band_block(const BandMap &m, int &bx, int &by, int &bz) {
  unsigned s = blockIdx.x >> 3; const unsigned xcd = blockIdx.x & 7u;
  bx = s % m.gx; s /= m.gx; const unsigned sb = s % m.sub; s /= m.sub; bz = s % m.gz; s /= m.gz;
  by = (s * 8 + xcd) * m.sub + sb;
  return by < m.gy;
}
__device__ inline void stencil_block(int &bx, int &by, int &bz) {
  const unsigned gx = gridDim.x, gy = gridDim.y, gz = gridDim.z, SUB = 8;
  if (gy % SUB != 0 || (gx * (gy / SUB)) % 8 != 0) { bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z; return; }
  const unsigned L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  const unsigned xcd = L & 7u, s = L >> 3;
  const unsigned sub = s % SUB, k = (s / SUB) % gz, T = (s / (SUB * gz)) * 8 + xcd;
  bx = (int)(T % gx); by = (int)((T / gx) * SUB + sub); bz = (int)k;
}
#endif

// tile kernels split k into chunks until at least this many blocks exist (several rounds per CU balance the chip)
// k extent of the chunks the marching tile kernels split the z range into. CALES_KCHUNK overrides (experiments).
static inline int tile_kchunk(const cales_ctx *c, long nxy_blocks, int n3) {
  const int forced = c->fl.kchunk;
  if (forced > 0) return forced < n3 ? forced : n3;
  return 0;
}
static inline long tile_min_blocks(const cales_ctx *c) { return c->fl.tile_min_blocks; }
// shortest chunk of a grid with fewer blocks than CUs: every block runs at once, a launch lasts (chunk + 3-plane prologue) planes -- 64^3 Taylor-Green:
// momentum pass 77 -> 54 us per step with chunks of 4 planes instead of 8 (192 blocks of 7 planes against 48 of 11), the step 0.223 -> 0.200 ms
constexpr int SMALL_KCH = 4;
// Few rounds of blocks (a y slab of a decomposed grid, a mid-size grid): the marching tile kernels hold ONE block per CU, a launch then runs in
// ceil(blocks / CUs) rounds of (chunk length + prologue) planes each, and half-empty last rounds show -- one rank of eight of the 512^3 channel:
// strain-rate pass 1.48 -> 1.32 ms/step with 9 chunks of 57 planes (504 blocks, 1.97 rounds) instead of 16 of 32 (896 blocks, 3.5 rounds), the
// dynamic model's last pass 1.39 -> 1.27 with 7 chunks instead of 16; the measured order of eight chunk lengths follows rounds x (planes + 3).
// With many rounds (the 512^3 grid on one GPU: 6.5) the passes are bound by bandwidth, idle CUs of the last round leave theirs to the others and
// the rule does not hold (measured: 13 full rounds 3 % SLOWER than 6.5) -- so: only below seven rounds, chunk lengths from 16 planes. (6.5 rounds, end of
// round 6: one rank of four -- 16 chunks of 32 planes against 12 of 43 -- strain-rate pass 2.94 -> 2.74, last pass 2.60 -> 2.56, momentum 2.03 -> 1.91 ms/step,
// every chunk length tried in the order of the product; the 512 x 256 x 256 duct momentum 2.52 -> 2.44; one rank of two -- 8 chunks of 64 against 6 of 86 -- equal.)
static inline int balanced_kchunk(const cales_ctx *c, long nxy_blocks, int n3, int kch0, int kmax = 1 << 30) {
  const long ncu = c->ncu > 0 ? c->ncu : 256;
  const int nch0 = (n3 + kch0 - 1) / kch0;
  if (nxy_blocks * nch0 >= 7 * ncu || c->fl.kchunk > 0 || c->fl.tile_min_blocks != 2048) return kch0;      // (switches that force chunking: the tests' business)
  long best = -1; int bk = kch0;
  for (int nch = 1; nch <= 4 * nch0 + 4; ++nch) {
    const int kch = (n3 + nch - 1) / nch;
    if (kch < 16 && kch < kch0) break;
    if (kch > kmax) continue;
    const long nb = nxy_blocks * ((n3 + kch - 1) / kch), cost = ((nb + ncu - 1) / ncu) * (kch + 3);
    if (best < 0 || cost < best) { best = cost; bk = kch; }
  }
  return bk;
}
// some wall-model face of the CASE (on whichever rank) has its sampling height inside the first cell: its interpolation reaches the ghost cell (index_wm = 1 / n,
// wmodel.f90:120-131; the conditions of host_setup.cpp's index search, from global quantities only -- every rank must answer alike: what hangs on the answer
// moves an all-reduce from one place of the substep to another)
static inline bool wm_samples_ghost(const cales_ctx *c) {
  const real h = c->C.hwm;
  for (int d = 1; d <= 3; ++d) for (int sd = 0; sd <= 1; ++sd) if (c->C.lwm[sd + 2 * (d - 1)] != 0) {
    const real first = d < 3 ? 0.5 * c->dl[d - 1] : (sd == 0 ? c->zc[1] : c->C.l[2] - c->zc[c->n[2]]);      // distance of the first cell centre from the wall
    if (!(first < h)) return true;
  }
  return false;
}
static inline int bc_skipped(const cales_ctx *c) { return c->bc_skip | (c->step_xskip ? 1 : 0); }
static inline dim3 grid3(int nx, int ny, int nz, dim3 b) { return dim3((nx + b.x - 1) / b.x, (ny + b.y - 1) / b.y, (nz + b.z - 1) / b.z); }
