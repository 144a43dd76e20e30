// FFT-based direct Poisson solver and the z tridiagonal sweeps (reference src/solver.f90:20-233,
// src/solver_gpu.f90:32-477, src/fft.f90, src/initsolver.f90).
//
// MI355X design (no hipFFT/rocFFT, no transposes on one GPU, no MFMA -- the pass is HBM-bound):
//   x pass : batched real-to-complex FFT of the contiguous rows. The n1 reals of a row become n1/2+1
//            complex modes that are written IN PLACE over the haloed row (n1+2 doubles = n1/2+1 real2).
//   y pass : batched complex FFT of the strided columns; a workgroup stages CB adjacent complex columns
//            (CB*16 B contiguous per row) of one z-plane in LDS, transforms them there and writes back.
//   z pass : Thomas algorithm, one thread per complex mode (re,im share the pivots), coalesced over x.
//   then y and x inverse passes, scaled by normfft.
// Transforms are radix-4/2/3/5 Stockham stages in LDS (ping-pong buffers), twiddles from a table.
// Spectral layout differs from FFTW's half-complex order; eigenvalues are laid out to match, so the
// solution p = solver(rhs) is the same discrete function (checked against the oracle).
#include "common.hpp"
#include <list>
#include <mutex>

// (aligned to its own size: LDS and global accesses of a complex value are ONE 128-bit operation -- ds_read_b128 at 256 B/clk instead of ds_read2_b64 at
//  128, MI355X_MICROARCH.md "LDS"; every cpx array here starts on a 16-byte boundary: hipMalloc'ed tables, the __align__(16) dynamic LDS block)
struct __attribute__((aligned(2 * sizeof(real)))) cpx { real x, y; };
__device__ inline cpx cadd(cpx a, cpx b) { return {a.x + b.x, a.y + b.y}; }
__device__ inline cpx csub(cpx a, cpx b) { return {a.x - b.x, a.y - b.y}; }
__device__ inline cpx cmul(cpx a, cpx b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ inline cpx cconj(cpx a) { return {a.x, -a.y}; }
template <int INV> __device__ inline cpx mul_mi(cpx a) { return INV ? cpx{-a.y, a.x} : cpx{a.y, -a.x}; }   // fwd: *(-i), inv: *(+i)

// one Stockham autosort stage of radix R on a length-N line: in -> out, t = lane within the line's T threads
template <int R, int INV>
__device__ inline void fft_stage(const cpx *__restrict__ in, cpx *__restrict__ out, int N, int Ns, int t, int T, const cpx *__restrict__ tw) {
  const int M = N / R, tstep = N / (Ns * R);
  for (int j = t; j < M; j += T) {
    const int k = j % Ns;
    cpx v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = in[j + r * M];
    if (Ns > 1) {
#pragma unroll
      for (int r = 1; r < R; ++r) { cpx w = tw[k * r * tstep]; if (INV) w.y = -w.y; v[r] = cmul(v[r], w); }
    }
    const int j0 = (j - k) * R + k;
    if (R == 2) {
      out[j0] = cadd(v[0], v[1]); out[j0 + Ns] = csub(v[0], v[1]);
    } else if (R == 4) {
      const cpx a0 = cadd(v[0], v[2]), a1 = csub(v[0], v[2]), a2 = cadd(v[1], v[3]), a3 = mul_mi<INV>(csub(v[1], v[3]));
      out[j0] = cadd(a0, a2); out[j0 + Ns] = cadd(a1, a3); out[j0 + 2 * Ns] = csub(a0, a2); out[j0 + 3 * Ns] = csub(a1, a3);
    } else if (R == 3) {
      const real s3 = INV ? 0.86602540378443864676 : -0.86602540378443864676;
      const cpx s = cadd(v[1], v[2]), d = csub(v[1], v[2]);
      const cpx m = {v[0].x - 0.5 * s.x, v[0].y - 0.5 * s.y}, rot = {-s3 * d.y, s3 * d.x};   // i*s3*d
      out[j0] = cadd(v[0], s); out[j0 + Ns] = cadd(m, rot); out[j0 + 2 * Ns] = csub(m, rot);
    } else if (R == 5) {
      const real c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;
      const real s1 = INV ? 0.95105651629515357212 : -0.95105651629515357212, s2 = INV ? 0.58778525229247312917 : -0.58778525229247312917;
      const cpx a = cadd(v[1], v[4]), b = csub(v[1], v[4]), c = cadd(v[2], v[3]), d = csub(v[2], v[3]);
      const cpx m1 = {v[0].x + c1 * a.x + c2 * c.x, v[0].y + c1 * a.y + c2 * c.y}, m2 = {v[0].x + c2 * a.x + c1 * c.x, v[0].y + c2 * a.y + c1 * c.y};
      const cpx r1 = {-(s1 * b.y + s2 * d.y), s1 * b.x + s2 * d.x}, r2 = {-(s2 * b.y - s1 * d.y), s2 * b.x - s1 * d.x};
      out[j0] = cadd(v[0], cadd(a, c)); out[j0 + Ns] = cadd(m1, r1); out[j0 + 4 * Ns] = csub(m1, r1);
      out[j0 + 2 * Ns] = cadd(m2, r2); out[j0 + 3 * Ns] = csub(m2, r2);
    } else {                       // other odd primes (7, 11, 13): direct DFT of the butterfly, W_R^m = tw[m N/R]
      const int wR = N / R;
#pragma unroll
      for (int q = 0; q < R; ++q) {
        cpx acc = v[0];
#pragma unroll
        for (int r = 1; r < R; ++r) { cpx w = tw[((q * r) % R) * wR]; if (INV) w.y = -w.y; acc = cadd(acc, cmul(v[r], w)); }
        out[j0 + q * Ns] = acc;
      }
    }
  }
}

// any other prime factor (17, 19, 23, ...): the same direct DFT with a run-time radix, operands re-read from LDS (R^2 complex
// multiply-adds per butterfly: meant for the odd sizes FFTW would also take, not for speed)
template <int INV>
__device__ inline void fft_stage_any(const cpx *__restrict__ in, cpx *__restrict__ out, int N, int Ns, int R, int t, int T, const cpx *__restrict__ tw) {
  const int M = N / R, tstep = N / (Ns * R), wR = N / R;
  for (int j = t; j < M; j += T) {
    const int k = j % Ns, j0 = (j - k) * R + k;
    for (int q = 0; q < R; ++q) {
      cpx acc = {0., 0.};
      for (int r = 0; r < R; ++r) {
        cpx v = in[j + r * M];
        if (Ns > 1 && r > 0) { cpx w = tw[k * r * tstep]; if (INV) w.y = -w.y; v = cmul(v, w); }
        cpx w2 = tw[((q * r) % R) * wR]; if (INV) w2.y = -w2.y;
        acc = cadd(acc, cmul(v, w2));
      }
      out[j0 + q * Ns] = acc;
    }
  }
}

struct FftPlan { int N, nst, radix[16]; };
static bool make_plan(int N, FftPlan &P) {
  P.N = N; P.nst = 0; int m = N;
  while (m % 4 == 0) { P.radix[P.nst++] = 4; m /= 4; }
  while (m % 2 == 0) { P.radix[P.nst++] = 2; m /= 2; }
  while (m % 3 == 0) { P.radix[P.nst++] = 3; m /= 3; }
  while (m % 5 == 0) { P.radix[P.nst++] = 5; m /= 5; }
  for (int pr : {7, 11, 13}) while (m % pr == 0 && P.nst < 16) { P.radix[P.nst++] = pr; m /= pr; }
  for (int pr = 17; pr <= 127 && m > 1; pr += 2) while (m % pr == 0 && P.nst < 16) { P.radix[P.nst++] = pr; m /= pr; }      // fft_stage_any
  return m == 1;
}
// runs all stages; returns the buffer holding the result (a or b). All threads of the block must call it.
template <int INV>
__device__ inline cpx *fft_line(const FftPlan &P, cpx *a, cpx *b, int t, int T, const cpx *tw) {
  int Ns = 1;
  for (int s = 0; s < P.nst; ++s) {
    const int R = P.radix[s];
    if (R == 4) fft_stage<4, INV>(a, b, P.N, Ns, t, T, tw);
    else if (R == 2) fft_stage<2, INV>(a, b, P.N, Ns, t, T, tw);
    else if (R == 3) fft_stage<3, INV>(a, b, P.N, Ns, t, T, tw);
    else if (R == 5) fft_stage<5, INV>(a, b, P.N, Ns, t, T, tw);
    else if (R == 7) fft_stage<7, INV>(a, b, P.N, Ns, t, T, tw);
    else if (R == 11) fft_stage<11, INV>(a, b, P.N, Ns, t, T, tw);
    else if (R == 13) fft_stage<13, INV>(a, b, P.N, Ns, t, T, tw);
    else fft_stage_any<INV>(a, b, P.N, Ns, R, t, T, tw);
    Ns *= R;
    __syncthreads();
    cpx *tmp = a; a = b; b = tmp;
  }
  return a;
}

// ------------------------------------------------------------------------------------------ power-of-two lines: radix-8 in registers
// N = 2^p >= 16, T = N/8 threads per line. Each stage: every thread reads 8 complex from LDS into registers, ALL threads
// of the block pass a barrier, then write their 8 results back (Stockham positions) -- one LDS buffer, no ping-pong, and
// log8(N) stages (512 = 8*8*8) instead of log4. A last radix-4 / radix-2 stage handles p mod 3 != 0.
template <int INV> __device__ inline cpx tw_mul(cpx v, cpx w) { if (INV) w.y = -w.y; return cmul(v, w); }
template <int INV>
__device__ inline void fft8_regs(cpx *v) {   // natural order in -> natural order out
  const real h = 0.70710678118654752440;
  cpx a0 = cadd(v[0], v[4]), a1 = cadd(v[1], v[5]), a2 = cadd(v[2], v[6]), a3 = cadd(v[3], v[7]);
  cpx b0 = csub(v[0], v[4]), b1 = csub(v[1], v[5]), b2 = csub(v[2], v[6]), b3 = csub(v[3], v[7]);
  // b_n *= w8^n  (forward w8 = e^{-i pi/4})
  b1 = INV ? cpx{(b1.x - b1.y) * h, (b1.x + b1.y) * h} : cpx{(b1.x + b1.y) * h, (b1.y - b1.x) * h};
  b2 = mul_mi<INV>(b2);
  b3 = INV ? cpx{(-b3.x - b3.y) * h, (b3.x - b3.y) * h} : cpx{(-b3.x + b3.y) * h, (-b3.x - b3.y) * h};
  cpx c0 = cadd(a0, a2), c1 = cadd(a1, a3), d0 = csub(a0, a2), d1 = mul_mi<INV>(csub(a1, a3));
  v[0] = cadd(c0, c1); v[4] = csub(c0, c1); v[2] = cadd(d0, d1); v[6] = csub(d0, d1);
  c0 = cadd(b0, b2); c1 = cadd(b1, b3); d0 = csub(b0, b2); d1 = mul_mi<INV>(csub(b1, b3));
  v[1] = cadd(c0, c1); v[5] = csub(c0, c1); v[3] = cadd(d0, d1); v[7] = csub(d0, d1);
}
// all threads of the block must call it (barriers); buf holds the line; result in buf
// LDS index skew: one extra slot every 8 complex, so the stride-8 scatter of the first stage (and the transposed
// loads/stores of the callers) spread over the banks
__device__ inline int lpad(int i) { return i + (i >> 3); }
template <int INV>
__device__ inline void fft_line8(int N, cpx *buf, int t, const cpx *__restrict__ tw) {
  const int T = N >> 3;
  int Ns = 1;
  while (Ns * 8 <= N) {                         // radix-8 stages, one butterfly per thread
    const int M = N >> 3, tstep = N / (Ns * 8), k = t % Ns, j0 = (t - k) * 8 + k;
    cpx v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = buf[lpad(t + r * M)];
    if (Ns > 1) {
#pragma unroll
      for (int r = 1; r < 8; ++r) v[r] = tw_mul<INV>(v[r], tw[k * r * tstep]);
    }
    fft8_regs<INV>(v);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) buf[lpad(j0 + r * Ns)] = v[r];
    __syncthreads();
    Ns *= 8;
  }
  if (Ns * 4 == N) {                             // last radix-4 stage: two butterflies per thread
    const int M = N >> 2, tstep = 1;
    cpx v[2][4]; int j0[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int j = t + b * T, k = j % Ns; j0[b] = (j - k) * 4 + k;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[b][r] = buf[lpad(j + r * M)];
#pragma unroll
      for (int r = 1; r < 4; ++r) v[b][r] = tw_mul<INV>(v[b][r], tw[k * r * tstep]);
      const cpx a0 = cadd(v[b][0], v[b][2]), a1 = csub(v[b][0], v[b][2]), a2 = cadd(v[b][1], v[b][3]), a3 = mul_mi<INV>(csub(v[b][1], v[b][3]));
      v[b][0] = cadd(a0, a2); v[b][1] = cadd(a1, a3); v[b][2] = csub(a0, a2); v[b][3] = csub(a1, a3);
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) buf[lpad(j0[b] + r * Ns)] = v[b][r];
    __syncthreads();
  } else if (Ns * 2 == N) {                      // last radix-2 stage: four butterflies per thread
    const int M = N >> 1;
    cpx v[4][2]; int j0[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int j = t + b * T, k = j % Ns; j0[b] = (j - k) * 2 + k;
      const cpx x0 = buf[lpad(j)], x1 = tw_mul<INV>(buf[lpad(j + M)], tw[k]);
      v[b][0] = cadd(x0, x1); v[b][1] = csub(x0, x1);
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 4; ++b) { buf[lpad(j0[b])] = v[b][0]; buf[lpad(j0[b] + Ns)] = v[b][1]; }
    __syncthreads();
  }
}

// The same transform with its ends in REGISTERS (decimation in frequency, the transposed flow graph of fft_line8): on entry v[e] = x[t + T e],
// e = 0..7, T = N/8 -- the elements a thread gets from coalesced global loads anyway --, on exit v[e] = X[t + T e], the elements it can store the
// same way. No staging copy in or out, one LDS round trip per stage boundary instead of two per stage, three barriers instead of eight for 512
// points. N = R0 8^a with R0 = 1, 2, 4: the odd radix comes FIRST (from registers, butterflies over x[j + r N/R0] = register slots of one thread),
// then radix-8 stages with Ns' = N/(8 R0), ..., 8, 1: stage (R, Ns') takes in[(j - k) R + k + r Ns'], k = j mod Ns', applies the butterfly, THEN the
// twiddle w^(k r N / (Ns' R)) to output r, and leaves out[j + r N/R]; the last stage (Ns' = 1) has no twiddles and stays in registers.
// All threads of the block call it.
// Index map of the register-ended transform: the identity. Its accesses are 128-bit: ds_write_b128 is served in groups of 8 CONSECUTIVE lanes on 32
// banks (8 slots of 16 B), ds_read_b128 in the four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, +32 on 64 banks (16 slots) -- MI355X_MICROARCH.md
// "LDS". In k_fft_y8r consecutive lanes are the CB columns of one t, so what spreads a group over the banks is the line pitch: odd (N + 9 at 512 points,
// N + 1 below, N + 2 for the four columns of 1024 points) makes every write conflict-free and the reads of the first stage 1.5-way -- against 4-way
// writes with the skewed map and a pitch of 4 (mod 16), whose aim was the reads only (a model of all stages: 4608 -> 1792 LDS-array cycles per tile and
// plane at 512 points). Measured: 1.46 -> 1.43 and 1.53 -> 1.52 ms per step for the two y passes -- the pass sits at the rate of its global access
// pattern, as round 3 found; the conflict counter is what changes.
__device__ inline int lmap(int i) { return i; }
template <int INV>
__device__ inline void fft_line8_dif(int N, cpx *buf, int t, const cpx *__restrict__ tw, cpx *v) {
  const int T = N >> 3;
  int R0 = N; while ((R0 & 7) == 0) R0 >>= 3;      // 1, 2 or 4
  int Nsp;
  if (R0 == 1) {                                 // radix 8 from registers: k = j = t, twiddle step 1
    fft8_regs<INV>(v);
    if (T == 1) return;
#pragma unroll
    for (int r = 1; r < 8; ++r) v[r] = tw_mul<INV>(v[r], tw[t * r]);
    Nsp = T >> 3;
  } else if (R0 == 4) {                          // two radix-4 butterflies: j = t + b T over the slots e = b, b + 2, b + 4, b + 6
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const cpx a0 = cadd(v[b], v[b + 4]), a1 = csub(v[b], v[b + 4]), a2 = cadd(v[b + 2], v[b + 6]), a3 = mul_mi<INV>(csub(v[b + 2], v[b + 6]));
      const int j = t + b * T;
      v[b] = cadd(a0, a2); v[b + 2] = tw_mul<INV>(cadd(a1, a3), tw[j]); v[b + 4] = tw_mul<INV>(csub(a0, a2), tw[2 * j]); v[b + 6] = tw_mul<INV>(csub(a1, a3), tw[3 * j]);
    }
    Nsp = T >> 2;                                // (N/4)/8
  } else {                                       // four radix-2 butterflies: j = t + b T over the slots e = b, b + 4
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const cpx x0 = v[b], x1 = v[b + 4];
      v[b] = cadd(x0, x1); v[b + 4] = tw_mul<INV>(csub(x0, x1), tw[t + b * T]);
    }
    Nsp = T >> 1;                                // (N/2)/8
  }
  while (true) {
#pragma unroll
    for (int e = 0; e < 8; ++e) buf[lmap(t + e * T)] = v[e];
    __syncthreads();
    const int k = t % Nsp, base = (t - k) * 8 + k, tstep = N / (Nsp * 8);
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = buf[lmap(base + r * Nsp)];
    fft8_regs<INV>(v);
    if (Nsp == 1) break;
#pragma unroll
    for (int r = 1; r < 8; ++r) v[r] = tw_mul<INV>(v[r], tw[k * r * tstep]);
    __syncthreads();                              // everybody has read: the line may be overwritten
    Nsp >>= 3;
  }
}

__device__ inline int dct_src(int e, int n) { return e < n / 2 ? 2 * e : 2 * (n - 1 - e) + 1; }   // Makhoul: v[e] = x[dct_src(e)]
// x pass for nh = n1/2 = 2^p. Persistent: a block owns `iters` consecutive groups of R rows, R = blockDim.x / (nh/8);
// the next group's rows are prefetched into registers while the current one is transformed; twiddles live in LDS.
// FILL = 1 (forward only, cales_step): the rows are not read from pp but formed from the velocities, pp = div(u*)/dtrk
// (fillps.f90:36-47, same expression as k_fillps) -- the separate fillps pass and its write + re-read of pp disappear.
// mean_mask != 0: the pass also sums comp*grid_vol_ratio(k) of the forced velocity components it reads anyway (bulk_mean,
// utils.f90:35-44), one partial per block and component -> the separate reduction pass over u disappears.
struct FillArgs { const real *u, *v, *w, *dzfi; real dti, dtidxi, dtidyi; int mean_mask; const real *gvr_f, *gvr_c; real *part; int pstride = 0, pofs = 0; int xwrap = 0; };      // xwrap = n1 with periodic x: u(0) is read as u(n1) (the ghost column may be stale inside cales_step), 0 otherwise      // pstride: partial sums per component over all launches of a chunked pass (0: gridDim.x)
template <int INV, int KIND, int FILL = 0>
// (the Neumann forward pass with fillps held 256 VGPRs + 13 AGPRs, i.e. ONE wave per SIMD; two blocks per CU cap it at 256 in all: 18 spilled
//  registers, 10.3 -> 7.3 ms at 1024^3)
__global__ __launch_bounds__(256, (KIND == 1 && FILL == 1) ? 2 : 1) void k_fft_x8(Geom g, int nh, int iters, const cpx *__restrict__ twg, const cpx *__restrict__ twpg,
                                                 const cpx *__restrict__ twd, real *__restrict__ p, real scale, Spec S, real2 *__restrict__ spec,
                                                 FillArgs F = FillArgs{}, long rbeg = 0, long rend = -1) {      // rows [rbeg, rend) of the (j,k) row list: a k-chunk of the pipelined solve
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int kind = KIND;                                                 // 0 periodic (R2HC/HC2R), 1 Neumann-Neumann (DCT-II/III)
  const int T = nh >> 3, R = blockDim.x / T, row = threadIdx.x / T, t = threadIdx.x % T, ld = lpad(nh) + 2;
  cpx *tw = reinterpret_cast<cpx *>(smem), *twp = tw + nh;                   // nh + (nh+1) twiddles
  cpx *twl = twp + nh + 1;                                                    // DCT weights (kind 1): nh+1 more
  cpx *A = twl + (kind ? nh + 1 : 0) + (size_t)row * ld;
  for (int q = threadIdx.x; q < nh; q += blockDim.x) tw[q] = twg[q];
  for (int q = threadIdx.x; q <= nh; q += blockDim.x) twp[q] = twpg[q];
  if (kind) for (int q = threadIdx.x; q <= nh; q += blockDim.x) twl[q] = twd[q];
  const long nrows = rend < 0 ? (long)g.n2 * g.n3 : rend;
  const int NE = 8;                                                          // elements per thread and row: nh / T
  cpx nxt[NE + 1];
  real macc[3] = {0., 0., 0.};
  auto rowptr = [&](long r, int &j, int &k) { j = (int)(r % g.n2) + 1; k = (int)(r / g.n2) + 1; };
  auto fetch = [&](long r) {
    if (r >= nrows) return;
    int j, k; rowptr(r, j, k);
    if (!INV && FILL) {
      const size_t c0 = g.ix(0, j, k);
      const real dz = F.dzfi[k];
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const size_t c = c0 + 1 + 2 * (t + e * T);      // cell i = 1 + 2q and its right neighbour: aligned pairs
        const real2 uu = *reinterpret_cast<const real2 *>(F.u + c); const real um = (F.xwrap && t + e * T == 0) ? F.u[c0 + F.xwrap] : F.u[c - 1];
        const real2 vv = *reinterpret_cast<const real2 *>(F.v + c), vm = *reinterpret_cast<const real2 *>(F.v + c - g.s1);
        const real2 ww = *reinterpret_cast<const real2 *>(F.w + c), wm = *reinterpret_cast<const real2 *>(F.w + c - g.s12);
        nxt[e] = cpx{((ww.x - wm.x) * F.dti * dz + (vv.x - vm.x) * F.dtidyi + (uu.x - um) * F.dtidxi),
                     ((ww.y - wm.y) * F.dti * dz + (vv.y - vm.y) * F.dtidyi + (uu.y - uu.x) * F.dtidxi)};
        if (F.mean_mask & 1) macc[0] += (uu.x + uu.y) * F.gvr_f[k];
        if (F.mean_mask & 2) macc[1] += (vv.x + vv.y) * F.gvr_f[k];
        if (F.mean_mask & 4) macc[2] += (ww.x + ww.y) * F.gvr_c[k];
      }
    } else if (!INV) {
      const real *rowp = p + g.ix(0, j, k);
#pragma unroll
      for (int e = 0; e < NE; ++e) { const int q = t + e * T; nxt[e] = cpx{rowp[1 + 2 * q], rowp[2 + 2 * q]}; }      // Makhoul's order is applied in LDS
    } else if (!kind) {
#pragma unroll
      for (int e = 0; e < NE; ++e) { const real2 v = spec[S.at_slab(g, t + e * T, j, k)]; nxt[e] = cpx{v.x, v.y}; }
      if (t == 0) {
        if (S.nyq) { nxt[NE] = cpx{nxt[0].y, 0.}; nxt[0].y = 0.; }      // slot 0 = (mode 0, mode nh), both real
        else { const real2 v = spec[S.at_slab(g, nh, j, k)]; nxt[NE] = cpx{v.x, v.y}; }
      }
    } else {            // DCT-III: the n real coefficients of the row as nh coalesced pairs; combined in LDS below
#pragma unroll
      for (int e = 0; e < NE; ++e) { const real2 v = spec[S.at_slab(g, t + e * T, j, k)]; nxt[e] = cpx{v.x, v.y}; }
    }
  };
  long r = rbeg + ((long)blockIdx.x * iters) * R + row;
  fetch(r);
  for (int it = 0; it < iters; ++it, r += R) {
    const bool live = r < nrows;
    int j = 1, k = 1; if (live) rowptr(r, j, k);
    if (kind && !INV) {          // x[2q] -> v[q], x[2q+1] -> v[n-1-q] (v = the real sequence the r2c transform sees, two reals per complex slot)
      real *Ad = reinterpret_cast<real *>(A);
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int q = t + e * T, r = 2 * nh - 1 - q;
        Ad[2 * lpad(q >> 1) + (q & 1)] = nxt[e].x; Ad[2 * lpad(r >> 1) + (r & 1)] = nxt[e].y;
      }
    } else {
#pragma unroll
      for (int e = 0; e < NE; ++e) A[lpad(t + e * T)] = nxt[e];
    }
    if (INV && !kind && t == 0) A[lpad(nh)] = nxt[NE];
    __syncthreads();
    if (it + 1 < iters) fetch(r + R);                                        // in flight during the transform
    if (INV && kind) {           // X_k = conj(w_k) (Y_k - i Y_{n-k}), Y_n := 0, k = 0..nh, from the coefficients staged in A
      const real *Ad = reinterpret_cast<const real *>(A); const int n = 2 * nh;
      auto coef = [&](int kk) {
        const real yk = Ad[2 * lpad(kk >> 1) + (kk & 1)];
        const int r2 = n - kk; const real ym = kk == 0 ? 0. : Ad[2 * lpad(r2 >> 1) + (r2 & 1)];
        return cmul(cconj(twl[kk]), cpx{yk, -ym});
      };
      cpx xk[NE + 1];
#pragma unroll
      for (int e = 0; e < NE; ++e) xk[e] = coef(t + e * T);
      if (t == 0) xk[NE] = coef(nh);
      __syncthreads();
#pragma unroll
      for (int e = 0; e < NE; ++e) A[lpad(t + e * T)] = xk[e];
      if (t == 0) A[lpad(nh)] = xk[NE];
      __syncthreads();
    }
    real *rowp = p + g.ix(0, j, k);
    if (!INV) {
      fft_line8<0>(nh, A, t, tw);
      if (live) {
        for (int kk = t; kk <= nh / 2; kk += T) {
          const cpx zk = A[lpad(kk)], zm = cconj(A[lpad((nh - kk) % nh)]);
          const cpx E = {(real)(0.5 * (zk.x + zm.x)), (real)(0.5 * (zk.y + zm.y))};
          const cpx D = csub(zk, zm), O = {(real)(0.5 * D.y), (real)(-0.5 * D.x)};     // -i/2 * (zk - conj(zm))
          const cpx wO = cmul(twp[kk], O);
          const cpx xk = cadd(E, wO), xm = cconj(csub(E, wO));
          if (!kind) {
            // (S.nyq: the modes 0 and nh of a real row are real -- they share slot 0, and no slot nh exists)
            if (S.nyq && kk == 0) spec[S.at_slab(g, 0, j, k)] = make_real2(xk.x, xm.x);
            else {
            spec[S.at_slab(g, kk, j, k)] = make_real2(xk.x, xk.y);
            spec[S.at_slab(g, nh - kk, j, k)] = make_real2(xm.x, xm.y);
            }
          } else {      // DCT-II coefficients Y_k = 2 Re(w_k V_k), Y_{n-k} = -2 Im(w_k V_k) at the real slots of the row
            real *sd = reinterpret_cast<real *>(spec);
            const int n = 2 * nh, k2 = nh - kk;
            const cpx a = cmul(twl[kk], xk), b2 = cmul(twl[k2], xm);
            auto put = [&](int rr, real val) { if (rr < n) sd[2 * S.at_slab(g, rr >> 1, j, k) + (rr & 1)] = val; };
            put(kk, 2. * a.x); if (kk) put(n - kk, -2. * a.y);
            put(k2, 2. * b2.x); if (k2 && k2 != n - k2) put(n - k2, -2. * b2.y);
          }
        }
      }
    } else {
      // Z'_k = (X_k + conj X_{nh-k}) + i conj(w^k) (X_k - conj X_{nh-k}), pairs (k, nh-k) in place
      for (int kk = t; kk <= nh / 2; kk += T) {
        const cpx xa = A[lpad(kk)], xb = A[lpad(nh - kk)];
        const cpx Sa = cadd(xa, cconj(xb)), Da = csub(xa, cconj(xb)), wa = cmul(cconj(twp[kk]), Da);
        const cpx Sb = cadd(xb, cconj(xa)), Db = csub(xb, cconj(xa)), wb = cmul(cconj(twp[nh - kk]), Db);
        const cpx za = cpx{Sa.x - wa.y, Sa.y + wa.x}, zb = cpx{Sb.x - wb.y, Sb.y + wb.x};
        A[lpad(kk)] = za;
        if (kk != 0 && 2 * kk != nh) A[lpad(nh - kk)] = zb;
      }
      __syncthreads();
      fft_line8<1>(nh, A, t, tw);
      if (live) {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const int q = t + e * T;
          if (kind) {            // x[2q] = v[q], x[2q+1] = v[n-1-q], read back through the same LDS mapping
            const real *Ad = reinterpret_cast<const real *>(A); const int r = 2 * nh - 1 - q;
            rowp[1 + 2 * q] = Ad[2 * lpad(q >> 1) + (q & 1)] * scale; rowp[2 + 2 * q] = Ad[2 * lpad(r >> 1) + (r & 1)] * scale;
          } else { const cpx z = A[lpad(q)]; rowp[1 + 2 * q] = z.x * scale; rowp[2 + 2 * q] = z.y * scale; }
        }
      }
    }
    __syncthreads();
  }
  if (FILL && F.mean_mask) {      // smem is free again: wave sums, then one partial per block and component
    real *red = reinterpret_cast<real *>(smem);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      if (!(F.mean_mask >> q & 1)) continue;
      real v = macc[q];
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
      __syncthreads();
      if (threadIdx.x == 0) { real a = 0.; for (unsigned w = 0; w < blockDim.x / 64; ++w) a += red[w]; F.part[(size_t)q * (F.pstride ? F.pstride : gridDim.x) + F.pofs + blockIdx.x] = a; }
      __syncthreads();
    }
  }
}

// Face-centred transform kinds of a velocity component along its OWN direction in the 3-D implicit step (find_fft with c_or_f = 'f',
// fft.f90:221-243), each through ONE complex FFT of the symmetric extension of the line (one line per block; these are the rare paths):
//   kind 5  'DD'  RODFT00 both ways, n-1 unknowns between the wall faces 0 and n: odd extension to 2n points, Y_k = i Z_k
//   kind 6  'NN'  REDFT00 both ways, n points: even extension to 2(n-1) points, Y_k = Z_k
//   kind 7  'DN'  RODFT01 forward: Y_k = (-1)^k x_{n-1} + 2 sum_{j<n-1} x_j sin(pi (j+1)(k+1/2)/n) = (i/2) Z_{2k+1} of the extension w of 4n points,
//                 odd about 0 and 2n and even about n (w_j' = x_{j'-1}, j' = 1..n); RODFT10 backward: Y_k = 2 sum x_j sin(pi (j+1/2)(k+1)/n) = i Z_{k+1}
//                 of z_{2j+1} = x_j, z_{4n-2j-1} = -x_j (4n points)
// ('ND' face-centred is REDFT10/01, the cell-centred Neumann pair: kind 1 with its own eigenvalues.) All are linear with real coefficients, so a
// complex column (y direction: real and imaginary part of an x mode) goes through as it is.
// DIR 0 / 2: rows along x, forward (field row -> real x modes, slab side of the spectrum) / inverse (modes -> field row, times `scale`),
// one row per block, coefficient k at the place of x_k; DIR 1: columns along y of the complex spectrum, one column per block (bwd: inverse).
template <int DIR>
__global__ __launch_bounds__(256) void k_dst1(Geom g, FftPlan P, int ncols, const cpx *__restrict__ tw, real *__restrict__ p, real scale,
                                              Spec S, real2 *__restrict__ pc, int kind = 5, int bwd = 0) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int N2 = P.N, n = kind == 5 ? N2 / 2 : kind == 6 ? N2 / 2 + 1 : N2 / 4, ld = N2 + 1, t = threadIdx.x, T = blockDim.x;
  cpx *A = reinterpret_cast<cpx *>(smem), *B = A + ld;
  int j = 0, k = 0, m = 0;
  if (DIR != 1) { const long r = blockIdx.x; j = (int)(r % g.n2) + 1; k = (int)(r / g.n2) + 1; }
  else { m = blockIdx.x; k = blockIdx.y + 1; }
  real *specd = reinterpret_cast<real *>(pc);
  auto slot = [&](int e) -> real & { return specd[2 * S.at_slab(g, e >> 1, j, k) + (e & 1)]; };      // real x mode e of row (j,k)
  real *rowp = p + g.ix(0, j, k);                                                                  // field row: x_e at rowp[e]
  auto rd = [&](int pos) -> cpx {      // value number pos (1-based) of the line
    if (DIR == 0) return cpx{rowp[pos], 0.};
    if (DIR == 2) return cpx{slot(pos - 1), 0.};
    const real2 v = pc[S.at_mode(g, m, pos, k)]; return cpx{v.x, v.y};
  };
  auto wr = [&](int pos, cpx y) {
    if (DIR == 0) slot(pos - 1) = y.x * scale; else if (DIR == 2) rowp[pos] = y.x * scale; else pc[S.at_mode(g, m, pos, k)] = make_real2(y.x * scale, y.y * scale);
  };
  const bool inv = DIR == 2 || (DIR == 1 && bwd);
  for (int q = t; q < N2; q += T) {
    cpx z = {0., 0.};
    if (kind == 5) {
      const int jj = q < n ? q : N2 - q;                    // |index| of the odd extension; 0 and n are the wall faces
      if (q != 0 && q != n) { z = rd(jj); if (q > n) { z.x = -z.x; z.y = -z.y; } }
    } else if (kind == 6) {
      z = rd((q < n ? q : N2 - q) + 1);
    } else if (!inv) {
      const int r = q <= 2 * n ? q : N2 - q;
      if (r != 0 && r != 2 * n) { z = rd(r <= n ? r : 2 * n - r); if (q > 2 * n) { z.x = -z.x; z.y = -z.y; } }
    } else if (q & 1) {
      if (q < 2 * n) z = rd((q - 1) / 2 + 1); else { z = rd((N2 - q - 1) / 2 + 1); z.x = -z.x; z.y = -z.y; }
    }
    A[q] = z;
  }
  __syncthreads();
  cpx *Z = fft_line<0>(P, A, B, t, T, tw);
  if (kind == 5) { for (int kk = t + 1; kk < n; kk += T) wr(kk, cpx{-Z[kk].y, Z[kk].x}); }                  // Y_k = i Z_k
  else if (kind == 6) { for (int kk = t; kk < n; kk += T) wr(kk + 1, Z[kk]); }
  else if (!inv) { for (int kk = t; kk < n; kk += T) wr(kk + 1, cpx{(real)(-0.5 * Z[2 * kk + 1].y), (real)(0.5 * Z[2 * kk + 1].x)}); }      // every x_j sits twice in (0, 2n)
  else { for (int kk = t; kk < n; kk += T) wr(kk + 1, cpx{-Z[kk + 1].y, Z[kk + 1].x}); }
  (void)ncols;
}

// DCT-IV / DST-IV along y (pressure Neumann on one y face and Dirichlet on the other: REDFT11 / RODFT11, fft.f90:192-245), the y twin of
// k_fft_x4: the real and the imaginary part of a column are two real sequences, each transformed by an N/2-point complex FFT of
// (x_{2q} + i x_{N-1-2q}) e^{-i pi (4q+1)/(4N)} with the post-twiddle e^{-i pi k/N}. Both kinds are their own inverse up to 2N
// (left to the inverse x pass), so the same kernel serves both directions. CB columns = 2 CB lines per block.
template <int DST>
__global__ __launch_bounds__(256) void k_fft_y4(Geom g, FftPlan P, int CB, int ncols, const cpx *__restrict__ tw, const cpx *__restrict__ tw4,
                                                Spec S, real2 *__restrict__ pc) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int nh = P.N, N = 2 * nh, ld = nh + 1, T = blockDim.x / (2 * CB);
  const int m0 = blockIdx.x * CB, k = blockIdx.y + 1;
  cpx *base = reinterpret_cast<cpx *>(smem);                                  // line l = 2 col + part: A at base + l*2*ld, B behind it
  for (int q = threadIdx.x; q < CB * nh; q += blockDim.x) {
    const int col = q % CB, qq = q / CB;
    if (m0 + col < ncols) {
      real2 a = pc[S.at_mode(g, m0 + col, 2 * qq + 1, k)], b = pc[S.at_mode(g, m0 + col, N - 1 - 2 * qq + 1, k)];
      if (DST) { const real2 tmp = a; a = b; b = tmp; }                     // reversed input
      base[(size_t)(2 * col) * 2 * ld + qq] = cmul(cpx{a.x, b.x}, tw4[qq]);
      base[(size_t)(2 * col + 1) * 2 * ld + qq] = cmul(cpx{a.y, b.y}, tw4[qq]);
    }
  }
  __syncthreads();
  const int line = threadIdx.x / T, t = threadIdx.x % T;
  cpx *A = base + (size_t)line * 2 * ld, *B = A + ld;
  cpx *Z = fft_line<0>(P, A, B, t, T, tw);
  const bool swapped = (Z != A);
  __syncthreads();
  for (int q = threadIdx.x; q < CB * nh; q += blockDim.x) {
    const int col = q % CB, kk = q / CB;
    if (m0 + col < ncols) {
      const cpx *Z0 = base + (size_t)(2 * col) * 2 * ld + (swapped ? ld : 0), *Z1 = Z0 + 2 * ld;
      const cpx c0 = cmul(Z0[kk], tw4[nh + kk]), c1 = cmul(Z1[kk], tw4[nh + kk]);
      const real so = DST ? 2. : -2.;                                        // DST: (-1)^k on the odd slots
      pc[S.at_mode(g, m0 + col, 2 * kk + 1, k)] = make_real2(2. * c0.x, 2. * c1.x);
      pc[S.at_mode(g, m0 + col, N - 1 - 2 * kk + 1, k)] = make_real2(so * c0.y, so * c1.y);
    }
  }
}

// y pass for N = n2 = 2^p: CB = blockDim.x / (N/8) adjacent complex columns; persistent over `kchunk` planes with
// register prefetch of the next plane; twiddles in LDS.
template <int INV, int KIND>
__global__ __launch_bounds__(512) void k_fft_y8(Geom g, int N, int ncols, int kchunk, const cpx *__restrict__ twg,
                                                 const cpx *__restrict__ twd, Spec S, real2 *__restrict__ pc, int k0 = 0, int k1 = -1) {      // planes k0+1..k1 (k1 < 0: all)
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int kind = KIND;
  const int T = N >> 3, CB = blockDim.x / T, ld = lpad(N) + 1;
  // 1024-point lines: four columns per block = HALF a 128-B line per row. Blocks are dealt to the eight XCDs in turn, so the two tiles of a line go
  // to blocks b and b + 8: same XCD, dispatched together -- the second half of every line is then a hit in that XCD's L2 instead of a second fetch
  int tile = blockIdx.x;
  if (CB == 4 && tile < (int)(gridDim.x & ~15u)) { const int r = tile & 15; tile = (tile & ~15) + ((r & 7) << 1) + (r >> 3); }
  const int m0 = tile * CB, kbeg = k0 + blockIdx.y * kchunk + 1, kend = min(kbeg + kchunk - 1, k1 < 0 ? g.n3 : k1);
  cpx *tw = reinterpret_cast<cpx *>(smem), *base = tw + N;
  for (int q = threadIdx.x; q < N; q += blockDim.x) tw[q] = twg[q];
  const int NE = 8;                                                          // CB*N / blockDim.x
  cpx nxt[NE];
  // (every global access unconditional -- columns beyond the last one repeat it: their lanes transform a copy of the last column and store ITS values to
  //  ITS places once more. A branch around a load or a store makes every wait of the plane loop an s_waitcnt vmcnt(0): the wait for the prefetched
  //  plane then also waits for the write acknowledgements of the plane just stored; tools/memseq.py shows counted waits where it showed `w0`.
  //  512 x 256 x 256 duct: y passes 0.390 / 0.417 -> 0.362 / 0.395 ms per step)
  auto fetch = [&](int k) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int q = threadIdx.x + e * blockDim.x, col = q % CB, j = q / CB;
      const real2 v = pc[S.at_mode(g, min(m0 + col, ncols - 1), j + 1, k)]; nxt[e] = cpx{v.x, v.y};
    }
  };
  // kind 1 (Neumann-Neumann, DCT-II/III on the real and the imaginary part alike): Makhoul order in, weights out (forward);
  // weights in, Makhoul order out (inverse) -- see k_fft_y
  auto makhoul = [&](int j) { return (j & 1) ? N - 1 - (j >> 1) : (j >> 1); };
  fetch(kbeg);
  for (int k = kbeg; k <= kend; ++k) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int q = threadIdx.x + e * blockDim.x, col = q % CB, j = q / CB;
      base[(size_t)col * ld + lpad((kind && !INV) ? makhoul(j) : j)] = nxt[e];
    }
    __syncthreads();
    fetch(min(k + 1, kend));                                                  // in flight during the transform (the last plane again: unused)
    if (kind && INV) {            // Z_k = conj(w_k) (C_k - i C_{N-k}), C_N := 0
      cpx tmp[NE];
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int q = threadIdx.x + e * blockDim.x, col = q % CB, kk = q / CB;
        const cpx ck = base[(size_t)col * ld + lpad(kk)], cm = kk == 0 ? cpx{0., 0.} : base[(size_t)col * ld + lpad(N - kk)];
        tmp[e] = cmul(cconj(twd[kk]), cpx{ck.x + cm.y, ck.y - cm.x});
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < NE; ++e) { const int q = threadIdx.x + e * blockDim.x, col = q % CB, kk = q / CB; base[(size_t)col * ld + lpad(kk)] = tmp[e]; }
      __syncthreads();
    }
    fft_line8<INV>(N, base + (size_t)(threadIdx.x / T) * ld, threadIdx.x % T, tw);
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int q = threadIdx.x + e * blockDim.x, c2 = q % CB, j = q / CB;
      {
        const cpx *Zc = base + (size_t)c2 * ld;
        cpx v;
        if (!kind) v = Zc[lpad(j)];
        else if (!INV) { const cpx w = twd[j]; v = cadd(cmul(w, Zc[lpad(j)]), cmul(cconj(w), Zc[lpad((N - j) % N)])); }   // C_j = w_j V_j + conj(w_j) V_{N-j}
        else v = Zc[lpad(makhoul(j))];
        pc[S.at_mode(g, min(m0 + c2, ncols - 1), j + 1, k)] = make_real2(v.x, v.y);
      }
    }
    __syncthreads();
  }
}

// Periodic y transform with the line's ends in registers (fft_line8_dif): thread (c, t) = (threadIdx.x % CB, threadIdx.x / CB) owns the elements
// j = t + T e of column m0 + c -- consecutive lanes = adjacent columns = whole 128-B segments of a row, on the way in and on the way out --, so the
// transposed staging copies of k_fft_y8 (and their LDS bank conflicts) disappear. Line pitch and index map: see lmap above.
template <int INV>
__global__ __launch_bounds__(512) void k_fft_y8r(Geom g, int N, int ncols, int kchunk, const cpx *__restrict__ twg, Spec S, real2 *__restrict__ pc, int k0 = 0, int k1 = -1) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int T = N >> 3, CB = blockDim.x / T, ld = N + (CB >= 8 ? (N == 512 ? 9 : 1) : 2);      // (see lmap; host: SolverPlans::shy8r is sized for the larger old pitch)
  const int c = threadIdx.x % CB, t = threadIdx.x / CB;
  int tile = blockIdx.x;      // (half-line tiles of 1024-point lines: both halves of a line on one XCD, see k_fft_y8)
  if (CB == 4 && tile < (int)(gridDim.x & ~15u)) { const int r = tile & 15; tile = (tile & ~15) + ((r & 7) << 1) + (r >> 3); }
  const int m0 = tile * CB, kbeg = k0 + blockIdx.y * kchunk + 1, kend = min(kbeg + kchunk - 1, k1 < 0 ? g.n3 : k1);
  cpx *tw = reinterpret_cast<cpx *>(smem), *line = tw + N + (size_t)c * ld;
  for (int q = threadIdx.x; q < N; q += blockDim.x) tw[q] = twg[q];
  const bool colok = m0 + c < ncols;
  const int mc = colok ? m0 + c : ncols - 1;      // (columns beyond the last one repeat it: every load and store unconditional)
  cpx nxt[8], v[8];
  auto fetch = [&](int k) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { const real2 q = pc[S.at_mode(g, mc, t + e * T + 1, k)]; nxt[e] = cpx{q.x, q.y}; }
  };
  fetch(kbeg);
  __syncthreads();
  for (int k = kbeg; k <= kend; ++k) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = nxt[e];
    fetch(min(k + 1, kend));                                                   // in flight during the transform (the last plane again: unused)
    fft_line8_dif<INV>(N, line, t, tw, v);
    // (unconditional: lanes of columns beyond the last one hold a copy of the last column and store ITS values to ITS places once more -- a branch
    //  around the stores makes the wait for the next plane's loads a wait for these stores as well, s_waitcnt vmcnt(0) instead of vmcnt(8))
#pragma unroll
    for (int e = 0; e < 8; ++e) pc[S.at_mode(g, mc, t + e * T + 1, k)] = make_real2(v[e].x, v[e].y);
    __syncthreads();                                                           // the last stage's reads are done before the next plane writes
  }
}

// ------------------------------------------------------------------------------------------ 1024-point lines in y: 16 elements per thread
// radix-16 butterfly in registers, natural order in and out, as 4 x 4: y[n1][k2] = DFT4 over n2 of v[n1 + 4 n2], times W16^(n1 k2),
// X[k2 + 4 k1] = DFT4 over n1
template <int INV>
__device__ inline void fft16_regs(cpx *v) {
  const real c1 = 0.92387953251128675613, s1 = 0.38268343236508977173, h = 0.70710678118654752440;      // cos(pi/8), sin(pi/8), sqrt(1/2)
  auto dft4 = [](cpx &a0, cpx &a1, cpx &a2, cpx &a3) {
    const cpx b0 = cadd(a0, a2), b1 = csub(a0, a2), b2 = cadd(a1, a3), b3 = mul_mi<INV>(csub(a1, a3));
    a0 = cadd(b0, b2); a1 = cadd(b1, b3); a2 = csub(b0, b2); a3 = csub(b1, b3);
  };
  // W16^m = exp(-+ 2 pi i m / 16) (forward -, inverse +), as (re, im of the FORWARD value)
  auto w16 = [](cpx a, real wr, real wi) { if (INV) wi = -wi; return cpx{a.x * wr - a.y * wi, a.x * wi + a.y * wr}; };
#pragma unroll
  for (int n1 = 0; n1 < 4; ++n1) dft4(v[n1], v[n1 + 4], v[n1 + 8], v[n1 + 12]);      // v[n1 + 4 k2] = y[n1][k2]
  v[1 + 4] = w16(v[1 + 4], c1, -s1); v[1 + 8] = w16(v[1 + 8], h, -h);   v[1 + 12] = w16(v[1 + 12], s1, -c1);      // m = 1, 2, 3
  v[2 + 4] = w16(v[2 + 4], h, -h);   v[2 + 8] = mul_mi<INV>(v[2 + 8]);   v[2 + 12] = w16(v[2 + 12], -h, -h);       // m = 2, 4, 6
  v[3 + 4] = w16(v[3 + 4], s1, -c1); v[3 + 8] = w16(v[3 + 8], -h, -h);  v[3 + 12] = w16(v[3 + 12], -c1, s1);      // m = 3, 6, 9
  // DFT4 over n1 for every k2: results X[k2 + 4 k1] -- in place that is a 4 x 4 transposition of the register names
  cpx x[16];
#pragma unroll
  for (int k2 = 0; k2 < 4; ++k2) {
    cpx a0 = v[0 + 4 * k2], a1 = v[1 + 4 * k2], a2 = v[2 + 4 * k2], a3 = v[3 + 4 * k2];
    dft4(a0, a1, a2, a3);
    x[k2] = a0; x[k2 + 4] = a1; x[k2 + 8] = a2; x[k2 + 12] = a3;
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) v[q] = x[q];
}
// small radices in registers, natural order in and out (the last stage of k_fft_y16)
template <int R, int INV>
__device__ inline void fftR_regs(cpx *u) {
  if (R == 8) fft8_regs<INV>(u);
  else if (R == 4) {
    const cpx b0 = cadd(u[0], u[2]), b1 = csub(u[0], u[2]), b2 = cadd(u[1], u[3]), b3 = mul_mi<INV>(csub(u[1], u[3]));
    u[0] = cadd(b0, b2); u[1] = cadd(b1, b3); u[2] = csub(b0, b2); u[3] = csub(b1, b3);
  } else { const cpx a = u[0], b = u[1]; u[0] = cadd(a, b); u[1] = csub(a, b); }
}
// y pass for N = 256, 512, 1024 with SIXTEEN elements per thread (reference src/fft.f90:323-493 for the Neumann kinds): EIGHT adjacent
// complex columns per block -- whole 128-B segments of every row at any of these lengths (the eight-elements-per-thread kernels above hold four columns
// of a 1024-point line, 64-B segments: tools/micro/segcopy copies that pattern at 3.4-3.9 TB/s against 4.7) -- with 8 T threads, T = N/16, thread
// (c, t) = (threadIdx.x % 8, threadIdx.x / 8) owning the elements i = t + T e, e = 0..15, of column m0 + c. Register-ended decimation in frequency like
// fft_line8_dif: radix 16 from registers (twiddles w^(t r)), radix 8 with Ns' = N/128 (two butterflies per thread, twiddles w^(16 k r)), a last stage of
// radix Ns' = 8, 4, 2 (two, four, eight butterflies per thread) back into registers as X[t + T e] -- two LDS round trips for the transform; eight lines
// + the twiddle table + the row offsets: 150 / 76 / 38 KB of the 160 KB of a CU, i.e. one block of 512 / two of 256 / four of 128 threads per CU at the
// two waves per SIMD its ~210 registers allow. KIND = 1 (Neumann-Neumann, DCT-II / DCT-III of the real and the imaginary part alike, Makhoul's
// re-ordering): forward the rows are LOADED in Makhoul order (v[i] = x[dct_src(i)]: any row order is coalesced, the lanes of a segment are columns) and
// the weights C_k = w_k V_k + conj(w_k) V_{N-k} need one more LDS exchange for the partner; inverse Z_k = conj(w_k) (C_k - i C_{N-k}) takes the exchange
// first and the results are STORED in Makhoul order. w_k = twd[t] twd[T e]: one register and sixteen block-uniform values instead of sixteen loads per
// plane. All global accesses unconditional (columns beyond the last one repeat it), the next plane prefetched into registers.
// 1024^3 cavity: y passes 15.2 / 16.4 -> 10.3 / 10.3 ms per step (0.40 -> 0.63 of the HBM peak).
template <int N, int INV, int KIND>
__global__ __launch_bounds__(N / 2, 2) void k_fft_y16(Geom g, int ncols, int kchunk, const cpx *__restrict__ twg, const cpx *__restrict__ twd, Spec S,
                                                      real2 *__restrict__ pc, int k0 = 0, int k1 = -1) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int T = N / 16, ld = N + 1;      // (odd pitch: the eight lanes of a write group are the eight columns -- eight different 16-B slots)
  constexpr int NSA = N / 128, RB = NSA, NBB = 16 / RB;      // stage A: radix 8, Ns' = NSA; stage B: radix RB = NSA, Ns' = 1, NBB butterflies per thread
  const int c = threadIdx.x & 7, t = threadIdx.x >> 3;
  const int m0 = blockIdx.x * 8, kbeg = k0 + blockIdx.y * kchunk + 1, kend = min(kbeg + kchunk - 1, k1 < 0 ? g.n3 : k1);
  cpx *tw = reinterpret_cast<cpx *>(smem), *line = tw + N + (size_t)c * ld;
  unsigned *ro = reinterpret_cast<unsigned *>(tw + N + 8 * (size_t)ld);      // element offset of row q + 1 from row 1 (the same for every column and plane)
  const size_t r0 = S.at_mode(g, 0, 1, 1);
  for (int q = threadIdx.x; q < N; q += 8 * T) { tw[q] = twg[q]; ro[q] = (unsigned)(S.at_mode(g, 0, q + 1, 1) - r0); }
  const int mc = min(m0 + c, ncols - 1);
  // planes are pstride elements apart; a0: the column's row 1 in plane 1
  const size_t a0 = S.at_mode(g, mc, 1, 1), pstride = S.blocked ? (size_t)S.cw * S.n2l : (size_t)(g.s12 >> 1);
  // rows this thread loads / stores for element i = t + T e: natural (i), or Makhoul's dct_src(i) = 2 i (e < 8), 2 (N - 1 - i) + 1 (e >= 8)
  // (`tv` = t behind an opaque move made anew in every plane: the row offsets are loop invariants, and hoisted out of the plane loop their thirty-two
  //  registers are spilled to scratch memory -- whose reloads count in vmcnt like the prefetch they then wait for)
  int tv = t;
  auto mrow = [&](int e) { return e < 8 ? 2 * (tv + T * e) : 2 * (N - 1 - (tv + T * e)) + 1; };
  cpx wt = {1., 0.};
  if (KIND) wt = twd[t];
  const __attribute__((address_space(4))) real *twu = (const __attribute__((address_space(4))) real *)twd;      // twd[T e]: block-uniform, scalar loads
  cpx nxt[16], v[16];
  auto fetch = [&](int k) {
    const real2 *pl = pc + a0 + (size_t)(k - 1) * pstride;
#pragma unroll
    for (int e = 0; e < 16; ++e) { const real2 q = pl[ro[(KIND && !INV) ? mrow(e) : tv + T * e]]; nxt[e] = cpx{q.x, q.y}; }
  };
  __syncthreads();      // (the tables)
  fetch(kbeg);
  for (int k = kbeg; k <= kend; ++k) {
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = nxt[e];
    asm volatile("" : "+v"(tv));
    fetch(min(k + 1, kend));                                                   // in flight during the transform (the last plane again: unused)
    if (KIND && INV) {      // Z_k = conj(w_k) (C_k - i C_{N-k}), C_N := 0, k = t + T e
#pragma unroll
      for (int e = 0; e < 16; ++e) line[t + T * e] = v[e];
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int kk = t + T * e;
        const cpx cm = kk == 0 ? cpx{0., 0.} : line[N - kk];
        const cpx w = e == 0 ? wt : cmul(wt, cpx{twu[2 * T * e], twu[2 * T * e + 1]});
        v[e] = cmul(cconj(w), cpx{v[e].x + cm.y, v[e].y - cm.x});
        if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // (four partners at a time, see the twiddles below)
      }
      __syncthreads();
    }
    // radix 16 from registers: in[t + T r] -> butterfly -> twiddle w^(t r) -> out[t + T r]
    fft16_regs<INV>(v);
    line[t] = v[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) {      // (four at a time: all fifteen twiddles fetched at once cost the registers the prefetched plane needs)
      line[t + T * r] = tw_mul<INV>(v[r], tw[t * r]);
      if ((r & 3) == 0) __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    // stage A, radix 8, Ns' = NSA: butterflies j = t and t + T: in[(j - k) 8 + k + NSA r], k = j mod NSA -> twiddle w^(16 k r) -> out[j + (N/8) r]
    const int kq = t & (NSA - 1);      // (T is a multiple of NSA: k is the same for both butterflies)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int j = t + T * b, base = (j - kq) * 8 + kq;
#pragma unroll
      for (int r = 0; r < 8; ++r) v[8 * b + r] = line[base + NSA * r];
      fft8_regs<INV>(v + 8 * b);
#pragma unroll
      for (int r = 1; r < 8; ++r) v[8 * b + r] = tw_mul<INV>(v[8 * b + r], tw[16 * kq * r]);
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 8; ++r) line[t + T * b + (N / 8) * r] = v[8 * b + r];
    __syncthreads();
    // stage B, radix RB, Ns' = 1: butterflies j = t + T b: in[RB j + r] -> X[j + (N/RB) r] = X[t + T (b + NBB r)], in registers
    cpx xo[16];
#pragma unroll
    for (int b = 0; b < NBB; ++b) {
      const int j = t + T * b;
      cpx u[RB];
#pragma unroll
      for (int r = 0; r < RB; ++r) u[r] = line[RB * j + r];
      fftR_regs<RB, INV>(u);
#pragma unroll
      for (int r = 0; r < RB; ++r) xo[b + NBB * r] = u[r];
    }
    real2 *pl = pc + a0 + (size_t)(k - 1) * pstride;
    if (KIND && !INV) {      // C_k = w_k V_k + conj(w_k) V_{N-k}, k = t + T e: the partner through LDS
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 16; ++e) line[t + T * e] = xo[e];
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int kk = t + T * e;
        const cpx pm = line[(N - kk) & (N - 1)];
        const cpx w = e == 0 ? wt : cmul(wt, cpx{twu[2 * T * e], twu[2 * T * e + 1]});
        const cpx o = cadd(cmul(w, xo[e]), cmul(cconj(w), pm));
        pl[ro[tv + T * e]] = make_real2(o.x, o.y);
        if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) pl[ro[(KIND && INV) ? mrow(e) : tv + T * e]] = make_real2(xo[e].x, xo[e].y);
    }
    __syncthreads();                                                           // this plane's last LDS reads are done before the next plane writes
  }
}

// ------------------------------------------------------------------------------------------ x pass
// rows are (j,k), j=1..n2, k=1..n3. R rows per block, T = blockDim.x / R threads per row.
// kind 0: periodic (FFTW R2HC/HC2R); kind 1: Neumann-Neumann cell-centred (REDFT10/REDFT01 = DCT-II/III) by Makhoul's
// reordering v[i] = x[2i], v[n-1-i] = x[2i+1] around the same real FFT: Y_k = 2 Re(w_k V_k), Y_{n-k} = -2 Im(w_k V_k),
// w_k = e^{-i pi k/(2n)} (table twd); the n real coefficients Y_0..Y_{n-1} are stored at the real slots 0..n-1 of the row.
template <int INV>
__global__ __launch_bounds__(256) void k_fft_x(Geom g, FftPlan P, int R, int kind, const cpx *__restrict__ tw, const cpx *__restrict__ twp,
                                                const cpx *__restrict__ twd, real *__restrict__ p, real scale, Spec S, real2 *__restrict__ spec) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int nh = P.N, n = 2 * nh, T = blockDim.x / R, row = threadIdx.x / T, t = threadIdx.x % T;
  const int ld = nh + 1;
  cpx *A = reinterpret_cast<cpx *>(smem) + (size_t)row * 2 * ld, *B = A + ld;
  const long r = (long)blockIdx.x * R + row, nrows = (long)g.n2 * g.n3;
  const bool live = r < nrows;
  const int j = live ? (int)(r % g.n2) + 1 : 1, k = live ? (int)(r / g.n2) + 1 : 1;
  real *rowp = p + g.ix(0, j, k);
  real *specd = reinterpret_cast<real *>(spec);
  if (!INV) {
    if (live) for (int q = t; q < nh; q += T) {
      if (!kind) A[q] = cpx{rowp[1 + 2 * q], rowp[2 + 2 * q]};
      else {       // kind 2 (Dirichlet-Dirichlet, RODFT10 = DST-II): DST-II(x)_k = DCT-II((-1)^j x_j)_{n-1-k}
        const int s0 = dct_src(2 * q, n), s1 = dct_src(2 * q + 1, n);
        const real g0 = (kind == 2 && (s0 & 1)) ? -1. : 1., g1 = (kind == 2 && (s1 & 1)) ? -1. : 1.;
        A[q] = cpx{g0 * rowp[1 + s0], g1 * rowp[1 + s1]};
      }
    }
    __syncthreads();
    cpx *Z = fft_line<0>(P, A, B, t, T, tw);
    if (live) {
      for (int kk = t; kk <= nh / 2; kk += T) {
        const cpx zk = Z[kk], zm = cconj(Z[(nh - kk) % nh]);
        const cpx E = {(real)(0.5 * (zk.x + zm.x)), (real)(0.5 * (zk.y + zm.y))};
        const cpx D = csub(zk, zm), O = {(real)(0.5 * D.y), (real)(-0.5 * D.x)};     // -i/2 * (zk - conj(zm))
        const cpx wO = cmul(twp[kk], O);
        const cpx xk = cadd(E, wO), xm = cconj(csub(E, wO));          // V_kk and V_{nh-kk}
        if (!kind) {
          spec[S.at_slab(g, kk, j, k)] = make_real2(xk.x, xk.y);
          spec[S.at_slab(g, nh - kk, j, k)] = make_real2(xm.x, xm.y);
        } else {
          const int k2 = nh - kk;
          const cpx a = cmul(twd[kk], xk), b2 = cmul(twd[k2], xm);
          // real slot r lives in pair r/2, component r%2
          auto put = [&](int rr, real val) { if (rr < n) { const int sl = kind == 2 ? n - 1 - rr : rr; specd[2 * S.at_slab(g, sl >> 1, j, k) + (sl & 1)] = val; } };
          put(kk, 2. * a.x); if (kk) put(n - kk, -2. * a.y);
          put(k2, 2. * b2.x); if (k2 && k2 != n - k2) put(n - k2, -2. * b2.y);
        }
      }
    }
  } else {
    if (live) {
      if (!kind) { for (int kk = t; kk <= nh; kk += T) { const real2 v = spec[S.at_slab(g, kk, j, k)]; B[kk] = cpx{v.x, v.y}; } }
      else for (int kk = t; kk <= nh; kk += T) {                      // X_k = conj(w_k) (Y_k - i Y_{n-k}), Y_n := 0
        // kind 2 (RODFT01 = DST-III): DST-III(y)_j = (-1)^j DCT-III(y reversed)_j
        const int sk_ = kind == 2 ? n - 1 - kk : kk, r2 = n - kk, sr = kind == 2 ? n - 1 - r2 : r2;
        const real yk = (sk_ >= 0 && sk_ < n) ? specd[2 * S.at_slab(g, sk_ >> 1, j, k) + (sk_ & 1)] : 0.;
        const real ym = kk == 0 ? 0. : specd[2 * S.at_slab(g, sr >> 1, j, k) + (sr & 1)];
        B[kk] = cmul(cconj(twd[kk]), cpx{yk, -ym});
      }
    }
    __syncthreads();
    for (int kk = t; kk < nh; kk += T) {
      const cpx xk = B[kk], xm = cconj(B[nh - kk]);
      const cpx S2 = cadd(xk, xm), D = csub(xk, xm);
      const cpx wD = cmul(cconj(twp[kk]), D);
      A[kk] = cpx{S2.x - wD.y, S2.y + wD.x};                         // S + i*conj(w^k)*D
    }
    __syncthreads();
    cpx *z = fft_line<1>(P, A, B, t, T, tw);
    if (live) for (int q = t; q < nh; q += T) {
      if (!kind) { rowp[1 + 2 * q] = z[q].x * scale; rowp[2 + 2 * q] = z[q].y * scale; }
      else {
        const int s0 = dct_src(2 * q, n), s1 = dct_src(2 * q + 1, n);
        rowp[1 + s0] = ((kind == 2 && (s0 & 1)) ? -scale : scale) * z[q].x; rowp[1 + s1] = ((kind == 2 && (s1 & 1)) ? -scale : scale) * z[q].y;
      }
    }
  }
}

// kinds 3 (Neumann-Dirichlet, REDFT11 = DCT-IV) and 4 (Dirichlet-Neumann, RODFT11 = DST-IV) in x; both are their own inverse up
// to the factor 2n that normfft carries. DCT-IV of n reals through one nh = n/2 point complex transform:
//   v_q = (x_{2q} + i x_{n-1-2q}) e^{-i pi (4q+1)/(4n)},  V = FFT_nh(v),  c_k = V_k e^{-i pi k/n},  y_{2k} = 2 Re c_k,  y_{n-1-2k} = -2 Im c_k;
// DST-IV(x)_k = (-1)^k DCT-IV(x reversed)_k. tw4 = [e^{-i pi (4q+1)/(4n)}, q < nh | e^{-i pi k/n}, k < nh].
// INV = 0 reads the physical row and writes the coefficient slots (as the NN transform does), INV = 1 the other way with `scale`.
template <int INV, int DST>
__global__ __launch_bounds__(256) void k_fft_x4(Geom g, FftPlan P, int R, const cpx *__restrict__ tw, const cpx *__restrict__ tw4,
                                                 real *__restrict__ p, real scale, Spec S, real2 *__restrict__ spec) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int nh = P.N, n = 2 * nh, T = blockDim.x / R, row = threadIdx.x / T, t = threadIdx.x % T;
  const int ld = nh + 1;
  cpx *A = reinterpret_cast<cpx *>(smem) + (size_t)row * 2 * ld, *B = A + ld;
  const long r = (long)blockIdx.x * R + row, nrows = (long)g.n2 * g.n3;
  const bool live = r < nrows;
  const int j = live ? (int)(r % g.n2) + 1 : 1, k = live ? (int)(r / g.n2) + 1 : 1;
  real *rowp = p + g.ix(0, j, k);
  real *specd = reinterpret_cast<real *>(spec);
  auto phys = [&](int e) -> real & { return rowp[1 + e]; };
  auto coef = [&](int e) -> real & { return specd[2 * S.at_slab(g, e >> 1, j, k) + (e & 1)]; };
  if (live) for (int q = t; q < nh; q += T) {
    real a = INV ? coef(2 * q) : phys(2 * q), b = INV ? coef(n - 1 - 2 * q) : phys(n - 1 - 2 * q);
    if (DST) { const real tmp = a; a = b; b = tmp; }                // reversed input
    A[q] = cmul(cpx{a, b}, tw4[q]);
  }
  __syncthreads();
  cpx *Z = fft_line<0>(P, A, B, t, T, tw);
  if (live) for (int kk = t; kk < nh; kk += T) {
    const cpx c = cmul(Z[kk], tw4[nh + kk]);
    const real ye = 2. * c.x * (INV ? scale : 1.), yo = (DST ? 2. : -2.) * c.y * (INV ? scale : 1.);     // DST: (-1)^k on the odd slots
    if (INV) { phys(2 * kk) = ye; phys(n - 1 - 2 * kk) = yo; } else { coef(2 * kk) = ye; coef(n - 1 - 2 * kk) = yo; }
  }
}

// ------------------------------------------------------------------------------------------ y pass
// block = CB adjacent complex columns (m0..m0+CB-1) of plane k; T = blockDim.x / CB threads per column.
// kind 1 (Neumann-Neumann): the DCT acts on the real and imaginary parts separately; with V = FFT(reordered column),
// C_k = w_k V_k + conj(w_k) V_{N-k} (forward) and Z_k = conj(w_k) (C_k - i C_{N-k}), C_N := 0 (inverse).
template <int INV>
__global__ __launch_bounds__(256) void k_fft_y(Geom g, FftPlan P, int CB, int ncols, int kind, const cpx *__restrict__ tw,
                                                const cpx *__restrict__ twd, Spec S, real2 *__restrict__ pc) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int N = P.N, ld = N + 1, T = blockDim.x / CB;
  const int m0 = blockIdx.x * CB, k = blockIdx.y + 1;
  cpx *base = reinterpret_cast<cpx *>(smem);
  // load: forward NN reorders rows into Makhoul order; inverse NN stages the raw coefficients in the second buffer
  for (int q = threadIdx.x; q < CB * N; q += blockDim.x) {
    const int col = q % CB, j = q / CB;
    if (m0 + col < ncols) {
      real2 v = pc[S.at_mode(g, m0 + col, j + 1, k)];
      // kind 2 (Dirichlet-Dirichlet, RODFT10/01): DST-II(x)_k = DCT-II((-1)^j x_j)_{N-1-k} -- the rows keep the reversed order (the y
      // eigenvalues are stored reversed, solver_setup) and only the signs of the odd rows change, on the way in and on the way out
      if (kind == 2 && !INV && (j & 1)) { v.x = -v.x; v.y = -v.y; }
      int dst = j;
      if (kind && !INV) dst = (j & 1) ? N - 1 - (j >> 1) : (j >> 1);
      base[(size_t)col * 2 * ld + ((kind && INV) ? ld : 0) + dst] = cpx{v.x, v.y};
    }
  }
  __syncthreads();
  if (kind && INV) {
    for (int q = threadIdx.x; q < CB * N; q += blockDim.x) {
      const int col = q / N, kk = q % N;
      const cpx *Bc = base + (size_t)col * 2 * ld + ld;
      const cpx ck = Bc[kk], cm = kk == 0 ? cpx{0., 0.} : Bc[N - kk];
      base[(size_t)col * 2 * ld + kk] = cmul(cconj(twd[kk]), cpx{ck.x + cm.y, ck.y - cm.x});     // C_k - i C_{N-k}
    }
    __syncthreads();
  }
  const int col = threadIdx.x / T, t = threadIdx.x % T;
  cpx *A = base + (size_t)col * 2 * ld, *B = A + ld;
  cpx *Z = fft_line<INV>(P, A, B, t, T, tw);
  const bool swapped = (Z != A);   // same for every column
  for (int q = threadIdx.x; q < CB * N; q += blockDim.x) {
    const int c2 = q % CB, j = q / CB;
    if (m0 + c2 < ncols) {
      const cpx *Zc = base + (size_t)c2 * 2 * ld + (swapped ? ld : 0);
      cpx v;
      if (!kind) v = Zc[j];
      else if (!INV) { const cpx w = twd[j]; v = cadd(cmul(w, Zc[j]), cmul(cconj(w), Zc[(N - j) % N])); }
      else v = Zc[(j & 1) ? N - 1 - (j >> 1) : (j >> 1)];
      if (kind == 2 && INV && (j & 1)) { v.x = -v.x; v.y = -v.y; }
      pc[S.at_mode(g, m0 + c2, j + 1, k)] = make_real2(v.x, v.y);
    }
  }
}

// ------------------------------------------------------------------------------------------ z pass (solver.f90:82-179)
// VT = real2: spectral pairs (re,im) of mode m in row j; VT = real: real field columns i=1..n1.
template <typename VT> __device__ inline VT vmul(VT a, real s);
template <> __device__ inline real2 vmul<real2>(real2 a, real s) { return make_real2(a.x * s, a.y * s); }
template <> __device__ inline real vmul<real>(real a, real s) { return a * s; }
template <typename VT> __device__ inline VT vfms(VT a, real s, VT b);   // a - s*b
template <> __device__ inline real2 vfms<real2>(real2 a, real s, real2 b) { return make_real2(a.x - s * b.x, a.y - s * b.y); }
template <> __device__ inline real vfms<real>(real a, real s, real b) { return a - s * b; }
template <typename VT> __device__ inline VT vfma(VT a, real s, VT b);   // a + s*b
template <> __device__ inline real2 vfma<real2>(real2 a, real s, real2 b) { return make_real2(a.x + s * b.x, a.y + s * b.y); }
template <> __device__ inline real vfma<real>(real a, real s, real b) { return a + s * b; }

// ncol x nrow columns; spectral solve: S maps (m, j) (mode side), mofs = global index of local mode 0, nmode = number of
// real modes (padding columns beyond it are skipped). Real fields (VT = real): in-place haloed array, i0 = 1.
// NN in x: the two reals of a pair are different modes -> two scalar recurrences (pivots d1,d2 kept as a real2)
__global__ __launch_bounds__(256) void k_gaussel_split(Geom g, int nz, int ncol, int nrow, int mofs, int nmode, Spec S, real lscale,
                                                       const real *__restrict__ a, const real *__restrict__ b, const real *__restrict__ c,
                                                       const real *__restrict__ lamx, const real *__restrict__ lamy,
                                                       real2 *__restrict__ p, real2 *__restrict__ dscr, int fixnull) {
  const int m = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y + 1;
  if (m >= ncol || j > nrow || m + mofs >= nmode) return;
  const size_t e0 = S.at_mode(g, m, j, 1), st = S.blocked ? (size_t)S.cw * S.n2l : (size_t)g.s12 / 2;
  const size_t s0 = (size_t)m + (size_t)ncol * (size_t)(j - 1), sst = (size_t)ncol * nrow;
  const real l1 = (lamx[2 * (m + mofs)] + lamy[j - 1]) * lscale, l2 = (lamx[2 * (m + mofs) + 1] + lamy[j - 1]) * lscale;
  const bool null1 = fixnull && l1 == 0., null2 = fixnull && l2 == 0.;      // see k_gaussel_ri
  real z1 = 1. / (b[0] + l1 + CALES_EPS), z2 = 1. / (b[0] + l2 + CALES_EPS), d1 = c[0] * z1, d2 = c[0] * z2;
  real2 v = p[e0]; v.x *= z1; v.y *= z2; p[e0] = v; dscr[s0] = make_real2(d1, d2);
  for (int l = 1; l < nz; ++l) {
    z1 = 1. / ((b[l] + l1) - a[l] * d1 + CALES_EPS); z2 = 1. / ((b[l] + l2) - a[l] * d2 + CALES_EPS);
    d1 = c[l] * z1; d2 = c[l] * z2;
    const real2 q = p[e0 + l * st];
    v = make_real2((q.x - a[l] * v.x) * z1, (q.y - a[l] * v.y) * z2);
    if (l == nz - 1) { if (null1) v.x = 0.; if (null2) v.y = 0.; }
    p[e0 + l * st] = v; dscr[s0 + l * sst] = make_real2(d1, d2);
  }
  for (int l = nz - 2; l >= 0; --l) {
    const real2 q = p[e0 + l * st], d = dscr[s0 + l * sst];
    v = make_real2(q.x - d.x * v.x, q.y - d.y * v.y);
    p[e0 + l * st] = v;
  }
}

// Non-periodic z, complex modes: the real and the imaginary part of a mode are two independent systems with the same
// matrix, so they go to two neighbouring lanes (lane -> mode lane/2, part lane%2; a wave still covers 512 contiguous bytes
// per row). Twice the waves of k_gaussel<real2,0> for the same traffic: the sweeps are latency-bound chains.
__global__ __launch_bounds__(256) void k_gaussel_ri(Geom g, int nz, int ncol, int nrow, int mofs, int nmode, Spec S, real lscale,
                                                    const real *__restrict__ a, const real *__restrict__ b,
                                                    const real *__restrict__ c, const real *__restrict__ lamx,
                                                    const real *__restrict__ lamy, real *__restrict__ p, real *__restrict__ dscr, int fixnull, int xreal) {
  // threads run over (row, mode, part) linearly, so a block touches one contiguous piece of a plane per step
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  const int t = (int)(q % (2 * ncol)), m = t >> 1, part = t & 1, j = (int)(q / (2 * ncol)) + 1;
  if (j > nrow || m + mofs >= nmode) return;
  const size_t e0 = 2 * S.at_mode(g, m, j, 1) + part;
  const size_t st = 2 * (S.blocked ? (size_t)S.cw * S.n2l : (size_t)g.s12 / 2);
  // scratch [k][j][m]: one c' per complex mode (both parts store the same value), one per lane when the parts are different modes
  const size_t s0 = xreal ? (size_t)t + (size_t)2 * ncol * (size_t)(j - 1) : (size_t)m + (size_t)ncol * (size_t)(j - 1);
  const size_t sst = (xreal ? (size_t)2 : (size_t)1) * ncol * nrow;
  // xreal (Neumann-Neumann in x): the two reals of a pair are different modes with their own eigenvalue
  const real lam = ((xreal ? lamx[2 * (m + mofs) + part] : lamx[m + mofs]) + lamy[j - 1]) * lscale;     // lscale = alpha for the Helmholtz solves (main.f90:441), 1 for the pressure
  real z = 1. / (b[0] + lam + CALES_EPS), d = c[0] * z;
  real v = p[e0] * z;
  p[e0] = v; dscr[s0] = d;      // both lanes of a pair store the same c' (one merged write); each reads back what it wrote
  for (int l = 1; l < nz; ++l) {
    const real bb = b[l] + lam;
    z = 1. / (bb - a[l] * d + CALES_EPS);
    d = c[l] * z;
    v = (p[e0 + l * st] - a[l] * v) * z;
    // The mode with zero eigenvalue of an all-Neumann/periodic problem is singular: its last pivot is 0 + eps and the reference
    // (solver.f90:160-178) returns [round-off of the r.h.s.]/eps there -- an arbitrary, possibly huge constant added to the
    // pressure, which then costs digits in every pressure difference. The constant is free: take the member with p(n) = 0.
    if (l == nz - 1 && fixnull && lam == 0.) v = 0.;
    p[e0 + l * st] = v; dscr[s0 + l * sst] = d;
  }
  for (int l = nz - 2; l >= 0; --l) {
    v = p[e0 + l * st] - dscr[s0 + l * sst] * v;
    p[e0 + l * st] = v;
  }
}

// Non-periodic z, one rank: the whole z extent of 16 neighbouring columns (one 128-B line per plane) is brought into LDS, solved
// there and written back -- 2 words per cell instead of the 5.3 of the marching sweep, which has to store c' and d'.
// One wave per mode (NV = 2: real and imaginary part share the matrix; NV = 1: real x modes, one column each); lane = chunk of M
// consecutive planes (64 M >= nz). Substructuring: every lane eliminates the M-1 interior rows of its chunk against the two
// separator values beside them (`+eps` pivots as in solver.f90:160-178), the 64 separator rows (last row of every chunk) form a
// tridiagonal system solved across the wave by parallel cyclic reduction, and the interiors follow by back substitution. Same
// equations as the Thomas sweep, other association (differences at round-off level times the conditioning of the column).
// Rows beyond nz are identity rows. Reciprocals: v_rcp_f64 + two Newton steps (full precision for normal numbers).
// LDS layout [column][chunk][M+1] (+4 doubles per column): conflict-free for the lane-per-chunk accesses and the plane-wise copies.
__device__ inline real rcp_nr(real x) {
  real r = __builtin_amdgcn_rcp(x);
  r = fma(r, fma(-x, r, 1.), r);
  return fma(r, fma(-x, r, 1.), r);
}
// a,b,c of the 64 M rows (identity rows beyond nz, no coupling out of the first and the last row) as [which][r][chunk]: the
// lanes of a wave (= chunks) read consecutive doubles
__global__ void k_abc_chunked(int nz, int M, const real *__restrict__ a, const real *__restrict__ b, const real *__restrict__ c, real *__restrict__ t, int nch = 64) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= nch * M) return;
  const int o = (k % M) * nch + k / M;
  t[o] = (k > 0 && k < nz) ? a[k] : 0.; t[nch * M + o] = k < nz ? b[k] : 1.; t[2 * nch * M + o] = k < nz - 1 ? c[k] : 0.;
}
// Segments of columns (blockIdx.y): one rank -- row j of the in-place spectrum (ndbl doubles, plane stride s12); several ranks --
// the block of peer blockIdx.y in the layout [peer][k][jl][m], whose (jl, m) planes are contiguous runs of 2 cw n2l doubles.
struct TileMap { int blocked, cw, n2l, mofs, nmode; size_t kstride, segstride; int nyq = 0;      // nyq: mode 0's column holds the modes 0 and n1/2 (Spec::nyq): left to k_gaussel_nyq
                
                 // z-only Helmholtz sweeps of real fields (no eigenvalue shift): nolam; cales_step forms the r.h.s. of rk.f90:108-118 and
                 // main.f90:422-433 while loading, (u - hf12*dudtd) + f + rhs_b, and plane nz+1 (wall face of w) receives the first two terms
                 int nolam, nq, has_lo, has_hi; const real *dud, *force; RhsBz blo, bhi; real hf12; };
// PER = 1: periodic z (solver.f90:109-150, gaussel_periodic): the tile solves the (n-1)-row system for the right-hand side AND for the closure
// vector p2 = (-a(1), 0, ..., 0, -c(n-1)) -- one more right-hand side of the same matrix, kept in registers -- and the last row follows from
// p(n) = (p(n) - c(n) p1(1) - a(n) p1(n-1)) / (b(n) + lambda + c(n) p2(1) + a(n) p2(n-1) + eps), p(1:n-1) = p1 + p2 p(n); ra, rb_, rc: the raw a, b, c.
// (nz > 512 with real x modes -- M = 16, NV = 1, the 1024^3 cavity: 140 KB of LDS leave one block per CU, so traffic (3.5 ms) and solve of that pass
//  do not overlap, and a 1024-thread block is capped at 128 registers. Round 3: the 29 spilled registers were not the 2 x 15 elimination coefficients
//  but the sixteen global addresses and sixteen LDS places of the load phase, kept alive for the store phase -- as many bytes of scratch traffic as the
//  tile itself (FETCH_SIZE / WRITE_SIZE 16 + 17 GB for 8.6 + 8.6 compulsory); formed again behind an opaque offset: 2 spilled registers, 7.7 -> 6.2 ms.
//  Tried and dropped: half as many waves solving the tile in two rounds under the 256-register cap (8.2 ms); tiles of eight columns, two blocks per
//  CU, the two half-line tiles of a line on the same XCD (8.6 ms with the spills, 6.4 ms without them against 6.2 for the full-line tile); two waves per column with 8 planes per lane and a 2 x 2 interface system between
//  their two cyclic reductions (no spills, 8.2 ms: twice the reductions, six more barriers per tile).)
constexpr int gt_width(int, int) { return 16; }      // columns per tile: one 128-B line per plane
#ifndef GT_TL
#define GT_TL 1
#endif
// pitch of a tile column in LDS = 64 (M + 1) + GT_PAD doubles. The load / store phases touch eight columns x (two planes per 16-lane group of a ds_write_b64,
// four per 32-lane group of a ds_read_b64; banks: doubles mod 16 / mod 32, MI355X_MICROARCH.md "LDS"): with a pitch = 4 (mod 16) the eight columns fell on two
// bank offsets for the stores (4-way) and four for the loads (2-way); 9 (mod 16) puts two columns 2 (mod 16) / 18 (mod 32) doubles apart -- conflict-free stores,
// one 2-way pair among the loads. Measured (round 6, z sweep per step): 512^3 1.667 -> 1.634 ms, 1024^3 13.36 -> 12.70, 256 x 128 x 128 0.130 -> 0.124
// (pads 1 and 5 within noise of 9). The solve phase walks one column per wave with a lane pitch of M + 1 doubles (odd): conflict-free at any column pitch.
#ifndef GT_PAD
#define GT_PAD 9
#endif
template <int M, int NV, int PER, int DUD = 0>      // DUD = 1: the z-only Helmholtz sweeps (TileMap::dud), which form their right-hand side while loading -- an instantiation of their own, the pressure solve carries none of it
__global__ __launch_bounds__(64 * gt_width(M, NV) / NV, (PER ? 2 : M <= 8 ? 4 : NV == 1 ? 4 : 2)) void k_gaussel_tile(Geom g, int nz, int ndbl, real lscale, const real *__restrict__ abc,
                                                            const real *__restrict__ lamx, const real *__restrict__ lamy,
                                                            real *__restrict__ p, int fixnull, TileMap T,
                                                            const real *__restrict__ ra = nullptr, const real *__restrict__ rb_ = nullptr, const real *__restrict__ rc = nullptr) {
  extern __shared__ real shz[];
  const int nsys = PER ? nz - 1 : nz;      // rows of the tridiagonal system proper
  constexpr int W = gt_width(M, NV), CP = M + 1, P = 64 * CP + GT_PAD, NT = 64 * W / NV;
  const int t = threadIdx.x;
  const int tile = blockIdx.x;
  const size_t base = (T.blocked ? T.segstride * blockIdx.y : g.ix(0, blockIdx.y + 1, 1)) + (size_t)W * tile;      // doubles from p
  const size_t kst = T.blocked ? T.kstride : (size_t)g.s12;
  // 16-plane chunks: every wave reads the 3 x 16 table entries of its lanes -- 1.5 x the tile's own traffic through the vector cache. The sub- and
  // the superdiagonal go to the 16 KB of LDS the tile leaves free (GT_TL; the diagonal stays in the table)
  constexpr bool TL = GT_TL && M == 16;
  real *tabl = shz + W * P;
  if (TL) { for (int q = t; q < 64 * M; q += NT) { tabl[q] = abc[q]; tabl[64 * M + q] = abc[128 * M + q]; } }
  {
    // two neighbouring columns per thread, 16-byte accesses: eight lanes per 128-B line and half as many load / store instructions (the copy of this
    // pattern alone: 0.57 -> 0.46 ms at 512^3, 4.85 -> 3.62 ms at 1024^3, tools/micro/ztile.hip); ndbl is even (ng(1), 2 cw n2l)
    constexpr int W2 = W / 2, KP2 = NT / W2, NQ2 = 64 * M / KP2;
    const int x = 2 * (t % W2), kk = t / W2;
    const bool ok = W * tile + x < ndbl;
    real2 v[NQ2];
#pragma unroll
    for (int q = 0; q < NQ2; ++q) {      // (conditional loads: measured faster than unconditional loads from clamped places, 0.59 against 0.87 ms at 512^3)
      const int k = kk + KP2 * q;
      v[q] = (ok && k < nz) ? *reinterpret_cast<const real2 *>(p + base + x + (size_t)k * kst) : make_real2(0., 0.);
    }
    if (NV == 2 && DUD) {      // (the z-only Helmholtz sweeps come as pairs of real columns that share the matrix: NV = 2 only)
      const real f = T.force ? T.force[0] : 0.;
      const int bi = ok ? W * tile + x + 1 : 1, bj = (int)blockIdx.y + 1;      // column (i, j) of the pair's first member
      // every load of this part unconditional, from a place that exists (plane 1 of the pair) where the lane has none, the selection afterwards: inside the
      // divergent branches they used to sit in, each was waited for on the spot -- 4 + 2 + 2 serial round trips per block (tools/memseq.py: L w0 L w0 ...)
      const size_t osafe = base + (ok ? x : 0);
      real2 dd[NQ2], ex[NQ2];
#pragma unroll
      for (int q = 0; q < NQ2; ++q) {
        const int k = kk + KP2 * q;
        dd[q] = *reinterpret_cast<const real2 *>(T.dud + ((ok && k < T.nq) ? base + x + (size_t)k * kst : osafe));
      }
      if (T.nq > nz) {      // (the wall face of w: its plane lies outside the system and only receives the explicit terms)
#pragma unroll
        for (int q = 0; q < NQ2; ++q) {
          const int k = kk + KP2 * q;
          ex[q] = *reinterpret_cast<const real2 *>(p + ((ok && k >= nz && k < T.nq) ? base + x + (size_t)k * kst : osafe));
        }
      }
      real2 lo = make_real2(0., 0.), hi = lo;
      if (T.has_lo) lo = make_real2(T.blo.at(bi, bj, g.n1), T.blo.at(bi + 1, bj, g.n1));
      if (T.has_hi) hi = make_real2(T.bhi.at(bi, bj, g.n1), T.bhi.at(bi + 1, bj, g.n1));
#pragma unroll
      for (int q = 0; q < NQ2; ++q) {
        const int k = kk + KP2 * q;
        const bool in = ok && k < T.nq;
        const real2 own = k < nz ? v[q] : ex[q];
        real2 t = make_real2(own.x - T.hf12 * dd[q].x, own.y - T.hf12 * dd[q].y);
        if (T.force) { t.x = t.x + f; t.y = t.y + f; }
        if (k == 0) { t.x = t.x + lo.x; t.y = t.y + lo.y; }
        if (k == nz - 1) { t.x = t.x + hi.x; t.y = t.y + hi.y; }
        if (in && k < nz) v[q] = t;
        if (in && k >= nz) *reinterpret_cast<real2 *>(p + base + x + (size_t)k * kst) = t;
      }
    }
#pragma unroll
    for (int q = 0; q < NQ2; ++q) { const int k = kk + KP2 * q; shz[x * P + k + k / M] = v[q].x; shz[(x + 1) * P + k + k / M] = v[q].y; }
  }
  __syncthreads();
  const int x = (t >> 6) * NV, ch = t & 63, d = W * tile + x;
  // column d of the segment -> x mode and global row
  int mode = NV == 1 ? d : d >> 1, j = blockIdx.y + 1;
  bool colok = d < ndbl;
  if (T.blocked) {
    const int f = d >> 1, mm = f % T.cw, jl = f / T.cw;
    j = blockIdx.y * T.n2l + jl + 1;
    mode = NV == 1 ? 2 * (mm + T.mofs) + (d & 1) : mm + T.mofs;
    colok = colok && mm + T.mofs < T.nmode;      // padding modes of the last rank: skipped (their slots are never read)
  }
  if (T.nyq && mode == 0) colok = false;      // (the packed column of the modes 0 and n1/2: k_gaussel_nyq)
  if (colok) {
    const real lam = T.nolam ? 0. : (lamx[mode] + lamy[j - 1]) * lscale;
    const bool nullc = fixnull && lam == 0.;      // singular mode: the member with p(nz) = 0, see k_gaussel_ri
    real *col = shz + x * P + ch * CP;
    const int k0 = ch * M;
    real cp[M - 1], V[M - 1];      // the swept right-hand sides go back to their LDS slots (registers: no spills at 4 waves/SIMD)
    real cprev = 0., vprev = 0., rprev[NV] = {};
    // periodic z: the last plane's right-hand side (its row is an identity row of the tile), the closure vector's two entries
    real pnr[NV] = {}, E[PER ? M - 1 : 1] = {}, eprev = 0.;
    const real e_first = PER ? -ra[0] : 0., e_last = PER ? -rc[nsys - 1] : 0.;
    if (PER) {
#pragma unroll
      for (int q = 0; q < NV; ++q) pnr[q] = shz[(x + q) * P + (nz - 1) + (nz - 1) / M];
    }
#pragma unroll
    for (int r = 0; r < M - 1; ++r) {
      const int k = k0 + r;
      const bool pin = !PER && nullc && k == nz - 1, live = k < nsys && !pin;
      // (every load unconditional, the selection afterwards: a load inside a divergent branch is waited for on the spot -- sixteen serial
      //  round trips to the table per chunk)
      const real A0 = TL ? tabl[r * 64 + ch] : abc[r * 64 + ch], C0 = TL ? tabl[64 * M + r * 64 + ch] : abc[128 * M + r * 64 + ch], B0 = abc[64 * M + r * 64 + ch];
      const real A = pin ? 0. : A0, C = pin ? 0. : C0, B = pin ? 1. : B0 + (k < nsys ? lam : 0.);
      const real z = rcp_nr(B - A * cprev + CALES_EPS);
      cp[r] = C * z; V[r] = (r == 0 ? A : -A * vprev) * z;
#pragma unroll
      for (int q = 0; q < NV; ++q) { const real D0 = col[q * P + r], D = live ? D0 : 0.; rprev[q] = (D - A * rprev[q]) * z; col[q * P + r] = rprev[q]; }
      if (PER) { const real D2 = (k == 0 ? e_first : 0.) + (k == nsys - 1 ? e_last : 0.); eprev = (D2 - A * eprev) * z; E[r] = eprev; }
      cprev = cp[r]; vprev = V[r];
    }
    // first interior row of the chunk as a function of the separators beside it
    real Vb = V[M - 2], Wb = cp[M - 2], Rb[NV], Eb = eprev;
#pragma unroll
    for (int q = 0; q < NV; ++q) Rb[q] = rprev[q];
#pragma unroll
    for (int r = M - 3; r >= 0; --r) {
      Vb = V[r] - cp[r] * Vb; Wb = -cp[r] * Wb;
#pragma unroll
      for (int q = 0; q < NV; ++q) Rb[q] = col[q * P + r] - cp[r] * Rb[q];
      if (PER) Eb = E[r] - cp[r] * Eb;
    }
    const bool last = ch == 63;
    // (cross-lane reads are issued by all lanes and masked afterwards: a lane switched off by a branch would be read as zero)
    real Vn = __shfl_down(Vb, 1, 64), Wn = __shfl_down(Wb, 1, 64);
    if (last) { Vn = 0.; Wn = 0.; }
    // separator row
    real al, be, ga, de[NV], de2 = 0.;
    {
      const int k = k0 + M - 1, r = M - 1;
      const bool pin = !PER && nullc && k == nz - 1, live = k < nsys && !pin;
      const real A0 = TL ? tabl[r * 64 + ch] : abc[r * 64 + ch], C0 = TL ? tabl[64 * M + r * 64 + ch] : abc[128 * M + r * 64 + ch], B0 = abc[64 * M + r * 64 + ch];
      const real A = pin ? 0. : A0, C = pin ? 0. : C0, B = pin ? 1. : B0 + (k < nsys ? lam : 0.);
      al = -A * V[M - 2]; be = B - A * cp[M - 2] - C * Vn; ga = -C * Wn;
      if (PER) {
        real En = __shfl_down(Eb, 1, 64);
        if (last) En = 0.;
        const real D2 = (k == 0 ? e_first : 0.) + (k == nsys - 1 ? e_last : 0.);
        de2 = D2 - A * eprev - C * En;
      }
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        real Rn = __shfl_down(Rb[q], 1, 64);
        if (last) Rn = 0.;
        const real D0 = col[q * P + M - 1], D = live ? D0 : 0.;
        de[q] = D - A * rprev[q] - C * Rn;
      }
    }
    // parallel cyclic reduction over the 64 separators
#pragma unroll
    for (int h = 1; h < 64; h <<= 1) {
      const bool lo = ch >= h, hi = ch + h < 64;
      const real rb = rcp_nr(be);
      const real rbm = __shfl_up(rb, h, 64), rbp = __shfl_down(rb, h, 64);
      const real k1 = lo ? al * rbm : 0., k2 = hi ? ga * rbp : 0.;
      const real alm = __shfl_up(al, h, 64), gam = __shfl_up(ga, h, 64), alp = __shfl_down(al, h, 64), gap = __shfl_down(ga, h, 64);
      be = be - gam * k1 - alp * k2;
#pragma unroll
      for (int q = 0; q < NV; ++q) de[q] = de[q] - __shfl_up(de[q], h, 64) * k1 - __shfl_down(de[q], h, 64) * k2;
      if (PER) de2 = de2 - __shfl_up(de2, h, 64) * k1 - __shfl_down(de2, h, 64) * k2;
      al = -alm * k1; ga = -gap * k2;
    }
    const real rb = rcp_nr(be);
    // periodic z: row n-1 of the system sits in chunk cL at place rL; rows 1 and n-1 of p1 and p2 are fetched from their lanes
    const int cL = PER ? (nsys - 1) / M : 0, rL = PER ? (nsys - 1) % M : 0;
    real Es = 0., e1 = 0., eL = 0.;
    if (PER) {
      Es = de2 * rb;
      real sp2 = __shfl_up(Es, 1, 64);
      if (ch == 0) sp2 = 0.;
      real xv2 = eprev - V[M - 2] * sp2 - cp[M - 2] * Es;
      E[M - 2] = xv2;
#pragma unroll
      for (int r = M - 3; r >= 0; --r) { xv2 = E[r] - V[r] * sp2 - cp[r] * xv2; E[r] = xv2; }
      real mine = Es;
#pragma unroll
      for (int r = 0; r < M - 1; ++r) if (r == rL) mine = E[r];
      e1 = __shfl(E[0], 0, 64); eL = __shfl(mine, cL, 64);
    }
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const real s = de[q] * rb;
      real sp = __shfl_up(s, 1, 64);
      if (ch == 0) sp = 0.;
      real xv = rprev[q] - V[M - 2] * sp - cp[M - 2] * s;
      real mine = (rL == M - 2) ? xv : s;
      col[q * P + M - 2] = xv;
#pragma unroll
      for (int r = M - 3; r >= 0; --r) { xv = col[q * P + r] - V[r] * sp - cp[r] * xv; col[q * P + r] = xv; if (PER && r == rL) mine = xv; }
      col[q * P + M - 1] = s;
      if (PER) {
        const real p11 = __shfl(xv, 0, 64), p1n = __shfl(mine, cL, 64);      // xv: row 1 of the chunk (lane 0: row 1 of the system)
        const real an = ra[nz - 1], bn = rb_[nz - 1], cn = rc[nz - 1];
        const real den = (bn + lam) + cn * e1 + an * eL + CALES_EPS;
        const real pn = nullc ? 0. : ((pnr[q] - cn * p11) - an * p1n) * (1. / den);      // null mode of the triply periodic problem: p(n) = 0
#pragma unroll
        for (int r = 0; r < M - 1; ++r) if (k0 + r < nsys) col[q * P + r] = col[q * P + r] + E[r] * pn;
        if (k0 + M - 1 < nsys) col[q * P + M - 1] = s + Es * pn;
        if (ch == nsys / M) col[q * P + nsys % M] = pn;
      }
    }
  }
  __syncthreads();
  {
    // (addresses formed again from an offset the compiler cannot recognise: it would otherwise keep those of the load phase alive through the
    //  solve, in scratch memory where registers are short -- 19 of the 29 spilled registers of the 16-plane instantiation)
    constexpr int W2 = W / 2, KP2 = NT / W2, NQ2 = 64 * M / KP2;
    const int x = 2 * (t % W2), kk = t / W2;
    const bool ok = W * tile + x < ndbl;
    size_t o = base + x + (size_t)kk * kst;
    int kq = kk;      // (the LDS places likewise)
    asm volatile("" : "+v"(o), "+v"(kq));
    const size_t step = (size_t)KP2 * kst;
#pragma unroll
    for (int q = 0; q < NQ2; ++q) {
      const int k = kq + KP2 * q;
      if (ok && k < nz) *reinterpret_cast<real2 *>(p + o) = make_real2(shz[x * P + k + k / M], shz[(x + 1) * P + k + k / M]);
      o += step;
    }
  }
}
// Nyquist packing (Spec::nyq): with periodic x the modes 0 and nh = n1/2 of a real row are REAL, so the x pass stores them as ONE complex value in slot 0 and
// a row of the spectrum is nh complex values -- whole 128-B lines for the y transforms, the z tile and the mode blocks of several ranks (257 columns were
// 32 tiles of eight and a 33rd for one column; 33 columns per rank on eight ranks were rows of 528 bytes that no line is aligned to). The y transform of
// that column is Z = A + i B with A, B the Hermitian spectra of the two real sequences; they have DIFFERENT x eigenvalues, so the z solve separates them by
// the rows ky and N - ky -- A = (Z(ky) + conj Z(N-ky)) / 2, B = (Z(ky) - conj Z(N-ky)) / (2 i) --, solves Re A, Im A with lambda_x(0) and Re B, Im B with
// lambda_x(nh) (four real columns of a tile: the solve is the text of k_gaussel_tile<M, 1, 0>), and puts Z'(ky) = A' + i B', Z'(N-ky) = conj A' + i conj B'
// back. One block per pair of rows, ky = 0 .. N/2 (ky = 0 and N/2 pair with themselves: A, B real there). 1 / (nh + 1) of the solve's work; the rank that
// holds mode 0 runs it. Reference: the same equations as src/solver.f90:82-179 for the modes 0 and n1/2 (FFTW's half-complex r0 and r_{n/2}).
// Neumann y (HERM = 0; ducts): the y transform is real, so Re Z is the transform of mode 0 and Im Z that of mode n1/2, row by row -- no pairing, only the two eigenvalues.
template <int M, int HERM>
__global__ __launch_bounds__(256) void k_gaussel_nyq(Geom g, int nz, int N, int nh, real lscale, const real *__restrict__ abc, const real *__restrict__ lamx,
                                                     const real *__restrict__ lamy, real2 *__restrict__ p, int fixnull, Spec S,
                                                     const real *__restrict__ ra = nullptr, const real *__restrict__ rb_ = nullptr, const real *__restrict__ rc = nullptr) {
  extern __shared__ real shz[];
  constexpr int NV = 1, PER = 0, CP = M + 1, P = 64 * CP + GT_PAD, NT = 256;
  constexpr bool TL = false;
  const real *tabl = nullptr;
  const int nsys = nz;
  // HERM = 0 (Neumann y: a real transform, the real and the imaginary part of a column transform independently): Re = mode 0, Im = mode n1/2 of the same
  // row, no pairing -- the block takes the rows 2 b and 2 b + 1, columns (Re, Im) of the one and of the other
  const int t = threadIdx.x, ky = HERM ? blockIdx.x : 2 * blockIdx.x, kp = HERM ? (N - ky) % N : ky + 1;
  for (int k = t; k < 64 * M; k += NT) {
    real2 z1 = make_real2(0., 0.), z2 = z1;
    if (k < nz) { z1 = p[S.at_mode(g, 0, ky + 1, k + 1)]; z2 = p[S.at_mode(g, 0, kp + 1, k + 1)]; }
    const int o = k + k / M;
    if (HERM) {
      shz[0 * P + o] = 0.5 * (z1.x + z2.x); shz[1 * P + o] = 0.5 * (z1.y - z2.y);      // A = (Z(ky) + conj Z(N-ky)) / 2
      shz[2 * P + o] = 0.5 * (z1.y + z2.y); shz[3 * P + o] = -0.5 * (z1.x - z2.x);     // B = (Z(ky) - conj Z(N-ky)) / (2 i)
    } else { shz[0 * P + o] = z1.x; shz[1 * P + o] = z1.y; shz[2 * P + o] = z2.x; shz[3 * P + o] = z2.y; }
  }
  __syncthreads();
  const int x = t >> 6, ch = t & 63;
  const bool colok = true;
  if (colok) {
    const real lam = HERM ? (lamx[x < 2 ? 0 : nh] + lamy[ky]) * lscale : (lamx[(x & 1) ? nh : 0] + lamy[x < 2 ? ky : kp]) * lscale;
    const bool nullc = fixnull && lam == 0.;      // singular mode: the member with p(nz) = 0, see k_gaussel_ri
    real *col = shz + x * P + ch * CP;
    const int k0 = ch * M;
    real cp[M - 1], V[M - 1];      // the swept right-hand sides go back to their LDS slots (registers: no spills at 4 waves/SIMD)
    real cprev = 0., vprev = 0., rprev[NV] = {};
    // periodic z: the last plane's right-hand side (its row is an identity row of the tile), the closure vector's two entries
    real pnr[NV] = {}, E[PER ? M - 1 : 1] = {}, eprev = 0.;
    const real e_first = PER ? -ra[0] : 0., e_last = PER ? -rc[nsys - 1] : 0.;
    if (PER) {
#pragma unroll
      for (int q = 0; q < NV; ++q) pnr[q] = shz[(x + q) * P + (nz - 1) + (nz - 1) / M];
    }
#pragma unroll
    for (int r = 0; r < M - 1; ++r) {
      const int k = k0 + r;
      const bool pin = !PER && nullc && k == nz - 1, live = k < nsys && !pin;
      // (every load unconditional, the selection afterwards: a load inside a divergent branch is waited for on the spot -- sixteen serial
      //  round trips to the table per chunk)
      const real A0 = TL ? tabl[r * 64 + ch] : abc[r * 64 + ch], C0 = TL ? tabl[64 * M + r * 64 + ch] : abc[128 * M + r * 64 + ch], B0 = abc[64 * M + r * 64 + ch];
      const real A = pin ? 0. : A0, C = pin ? 0. : C0, B = pin ? 1. : B0 + (k < nsys ? lam : 0.);
      const real z = rcp_nr(B - A * cprev + CALES_EPS);
      cp[r] = C * z; V[r] = (r == 0 ? A : -A * vprev) * z;
#pragma unroll
      for (int q = 0; q < NV; ++q) { const real D0 = col[q * P + r], D = live ? D0 : 0.; rprev[q] = (D - A * rprev[q]) * z; col[q * P + r] = rprev[q]; }
      if (PER) { const real D2 = (k == 0 ? e_first : 0.) + (k == nsys - 1 ? e_last : 0.); eprev = (D2 - A * eprev) * z; E[r] = eprev; }
      cprev = cp[r]; vprev = V[r];
    }
    // first interior row of the chunk as a function of the separators beside it
    real Vb = V[M - 2], Wb = cp[M - 2], Rb[NV], Eb = eprev;
#pragma unroll
    for (int q = 0; q < NV; ++q) Rb[q] = rprev[q];
#pragma unroll
    for (int r = M - 3; r >= 0; --r) {
      Vb = V[r] - cp[r] * Vb; Wb = -cp[r] * Wb;
#pragma unroll
      for (int q = 0; q < NV; ++q) Rb[q] = col[q * P + r] - cp[r] * Rb[q];
      if (PER) Eb = E[r] - cp[r] * Eb;
    }
    const bool last = ch == 63;
    // (cross-lane reads are issued by all lanes and masked afterwards: a lane switched off by a branch would be read as zero)
    real Vn = __shfl_down(Vb, 1, 64), Wn = __shfl_down(Wb, 1, 64);
    if (last) { Vn = 0.; Wn = 0.; }
    // separator row
    real al, be, ga, de[NV], de2 = 0.;
    {
      const int k = k0 + M - 1, r = M - 1;
      const bool pin = !PER && nullc && k == nz - 1, live = k < nsys && !pin;
      const real A0 = TL ? tabl[r * 64 + ch] : abc[r * 64 + ch], C0 = TL ? tabl[64 * M + r * 64 + ch] : abc[128 * M + r * 64 + ch], B0 = abc[64 * M + r * 64 + ch];
      const real A = pin ? 0. : A0, C = pin ? 0. : C0, B = pin ? 1. : B0 + (k < nsys ? lam : 0.);
      al = -A * V[M - 2]; be = B - A * cp[M - 2] - C * Vn; ga = -C * Wn;
      if (PER) {
        real En = __shfl_down(Eb, 1, 64);
        if (last) En = 0.;
        const real D2 = (k == 0 ? e_first : 0.) + (k == nsys - 1 ? e_last : 0.);
        de2 = D2 - A * eprev - C * En;
      }
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        real Rn = __shfl_down(Rb[q], 1, 64);
        if (last) Rn = 0.;
        const real D0 = col[q * P + M - 1], D = live ? D0 : 0.;
        de[q] = D - A * rprev[q] - C * Rn;
      }
    }
    // parallel cyclic reduction over the 64 separators
#pragma unroll
    for (int h = 1; h < 64; h <<= 1) {
      const bool lo = ch >= h, hi = ch + h < 64;
      const real rb = rcp_nr(be);
      const real rbm = __shfl_up(rb, h, 64), rbp = __shfl_down(rb, h, 64);
      const real k1 = lo ? al * rbm : 0., k2 = hi ? ga * rbp : 0.;
      const real alm = __shfl_up(al, h, 64), gam = __shfl_up(ga, h, 64), alp = __shfl_down(al, h, 64), gap = __shfl_down(ga, h, 64);
      be = be - gam * k1 - alp * k2;
#pragma unroll
      for (int q = 0; q < NV; ++q) de[q] = de[q] - __shfl_up(de[q], h, 64) * k1 - __shfl_down(de[q], h, 64) * k2;
      if (PER) de2 = de2 - __shfl_up(de2, h, 64) * k1 - __shfl_down(de2, h, 64) * k2;
      al = -alm * k1; ga = -gap * k2;
    }
    const real rb = rcp_nr(be);
    // periodic z: row n-1 of the system sits in chunk cL at place rL; rows 1 and n-1 of p1 and p2 are fetched from their lanes
    const int cL = PER ? (nsys - 1) / M : 0, rL = PER ? (nsys - 1) % M : 0;
    real Es = 0., e1 = 0., eL = 0.;
    if (PER) {
      Es = de2 * rb;
      real sp2 = __shfl_up(Es, 1, 64);
      if (ch == 0) sp2 = 0.;
      real xv2 = eprev - V[M - 2] * sp2 - cp[M - 2] * Es;
      E[M - 2] = xv2;
#pragma unroll
      for (int r = M - 3; r >= 0; --r) { xv2 = E[r] - V[r] * sp2 - cp[r] * xv2; E[r] = xv2; }
      real mine = Es;
#pragma unroll
      for (int r = 0; r < M - 1; ++r) if (r == rL) mine = E[r];
      e1 = __shfl(E[0], 0, 64); eL = __shfl(mine, cL, 64);
    }
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const real s = de[q] * rb;
      real sp = __shfl_up(s, 1, 64);
      if (ch == 0) sp = 0.;
      real xv = rprev[q] - V[M - 2] * sp - cp[M - 2] * s;
      real mine = (rL == M - 2) ? xv : s;
      col[q * P + M - 2] = xv;
#pragma unroll
      for (int r = M - 3; r >= 0; --r) { xv = col[q * P + r] - V[r] * sp - cp[r] * xv; col[q * P + r] = xv; if (PER && r == rL) mine = xv; }
      col[q * P + M - 1] = s;
      if (PER) {
        const real p11 = __shfl(xv, 0, 64), p1n = __shfl(mine, cL, 64);      // xv: row 1 of the chunk (lane 0: row 1 of the system)
        const real an = ra[nz - 1], bn = rb_[nz - 1], cn = rc[nz - 1];
        const real den = (bn + lam) + cn * e1 + an * eL + CALES_EPS;
        const real pn = nullc ? 0. : ((pnr[q] - cn * p11) - an * p1n) * (1. / den);      // null mode of the triply periodic problem: p(n) = 0
#pragma unroll
        for (int r = 0; r < M - 1; ++r) if (k0 + r < nsys) col[q * P + r] = col[q * P + r] + E[r] * pn;
        if (k0 + M - 1 < nsys) col[q * P + M - 1] = s + Es * pn;
        if (ch == nsys / M) col[q * P + nsys % M] = pn;
      }
    }
  }
  __syncthreads();
  for (int k = t; k < nz; k += NT) {
    const int o = k + k / M;
    const real ar = shz[0 * P + o], ai = shz[1 * P + o], br = shz[2 * P + o], bi = shz[3 * P + o];
    if (HERM) {
      p[S.at_mode(g, 0, ky + 1, k + 1)] = make_real2(ar - bi, ai + br);      // A' + i B'
      p[S.at_mode(g, 0, kp + 1, k + 1)] = make_real2(ar + bi, br - ai);      // conj A' + i conj B'
    } else { p[S.at_mode(g, 0, ky + 1, k + 1)] = make_real2(ar, ai); p[S.at_mode(g, 0, kp + 1, k + 1)] = make_real2(br, bi); }
  }
}
template <int M>
static void launch_gaussel_nyq(cales_ctx *c, int nz, int N, int nh, real lscale, const real *da, const real *db, const real *dc, real2 *p, int fixnull, const Spec &S, const real *tab) {
  constexpr int lds = 4 * (64 * (M + 1) + GT_PAD) * 8;
  if (c->ykind == 0) LAUNCH(c, (k_gaussel_nyq<M, 1>), dim3(N / 2 + 1), dim3(256), lds, c->stream, c->g, nz, N, nh, lscale, tab, c->d_lamx, c->d_lamy, p, fixnull, S, da, db, dc);
  else LAUNCH(c, (k_gaussel_nyq<M, 0>), dim3(N / 2), dim3(256), lds, c->stream, c->g, nz, N, nh, lscale, tab, c->d_lamx, c->d_lamy, p, fixnull, S, da, db, dc);
}
// The same tile PERSISTENT over `tpb` neighbouring tiles of a segment, for nz = 1024 planes in chunks of sixteen (the 1024^3 cavity): the
// classic form holds one block of 1024 threads and 140 KB of LDS per CU, so the load of a tile, its solve and its store run one after the other (5.6 ms
// at 1024^3, 0.38 of the HBM peak). Here a block has 512 threads -- at two waves per SIMD the 256 registers a thread may use hold the NEXT tile (sixteen
// 16-byte loads issued before the solve, in flight while it runs) --, every wave solves TWO columns one after the other (NV = 1: real x modes, one
// eigenvalue each), and the stores of a tile drain while the next one is filled and solved: every global access is unconditional (full tiles of full
// chunks only: ndbl a multiple of 16, nz = 1024; everything else keeps the classic form), so the waits are counted. Same arithmetic: the solve below is
// the text of k_gaussel_tile<16, NV, 0>. (The same form for 512 planes of complex modes -- eight planes per lane, 256 threads, two blocks per CU, all three
// diagonals in registers, the two columns left over by the whole tiles in a classic launch -- was built and measured at 512^3: 1.65 against 1.68 ms per
// step. Two classic blocks per CU already overlap each other's phases as well; not kept.)
template <int NV>
__global__ __launch_bounds__(512, 2) void k_gaussel_tile_p(Geom g, int nz, int ndbl, real lscale, const real *__restrict__ abc,
                                                           const real *__restrict__ lamx, const real *__restrict__ lamy,
                                                           real *__restrict__ p, int fixnull, TileMap T, int tpb,
                                                           const real *__restrict__ ra = nullptr, const real *__restrict__ rb_ = nullptr, const real *__restrict__ rc = nullptr) {
  extern __shared__ real shz[];
  constexpr int M = 16, PER = 0, CPW = NV == 1 ? 2 : 1;
  const int nsys = nz;
  constexpr int W = gt_width(M, NV), CP = M + 1, P = 64 * CP + GT_PAD, NT = 512;
  static_assert(64 * W / (NV * CPW) == NT, "512 threads: sixteen columns, CPW column sets per wave");
  const int t = threadIdx.x;
  const size_t seg = T.blocked ? T.segstride * blockIdx.y : g.ix(0, blockIdx.y + 1, 1);      // doubles from p
  const size_t kst = T.blocked ? T.kstride : (size_t)g.s12;
  constexpr bool TL = GT_TL && M == 16;
  real *tabl = shz + W * P;
  if (TL) { for (int q = t; q < 64 * M; q += NT) { tabl[q] = abc[q]; tabl[64 * M + q] = abc[128 * M + q]; } }
  const int tile0 = blockIdx.x * tpb, tend = min(tile0 + tpb, ndbl / W);
  constexpr int W2 = W / 2, KP2 = NT / W2, NQ2 = 64 * M / KP2;
  const int xl = 2 * (t % W2), kk = t / W2;
  real2 vn[NQ2];
  auto fetch = [&](int tile) {
    const real *src = p + seg + (size_t)W * tile + xl + (size_t)kk * kst;
#pragma unroll
    for (int q = 0; q < NQ2; ++q) vn[q] = *reinterpret_cast<const real2 *>(src + (size_t)(KP2 * q) * kst);
  };
  // the diagonal of this lane's chunk (the same for every column and tile; the sub- and the superdiagonal sit in LDS): sixteen values loaded ONCE -- a
  // global load inside the solve would be waited for behind the prefetched tile (memory operations return in order)
  real B0r[M];
#pragma unroll
  for (int r = 0; r < M; ++r) B0r[r] = abc[64 * M + r * 64 + (t & 63)];
  if (tile0 < tend) fetch(tile0);
  // (one tile; the first is peeled off the loop below, which is then entered with the operations in flight that its back edge carries -- a tile's loads
  //  with the previous tile's stores behind them -- and the compiler's counted waits hold: merged with the prologue's state every wait would be for everything)
  auto one_tile = [&](const int tile) {
#pragma unroll
    for (int q = 0; q < NQ2; ++q) { const int k = kk + KP2 * q; shz[xl * P + k + k / M] = vn[q].x; shz[(xl + 1) * P + k + k / M] = vn[q].y; }
    // the eigenvalues of this wave's columns BEFORE the next tile's loads go out: a global load inside the solve would be waited for behind them (memory
    // operations return in order). Padding modes of the last rank of a blocked layout read the table's first entry: their columns are skipped below.
    // (Loading them with the tile they belong to, one iteration ahead, measured 3 % slower: 13.9 against 13.5 ms per step at 1024^3.)
    real lamv[CPW]; bool cok[CPW];
#pragma unroll
    for (int cc = 0; cc < CPW; ++cc) {
      const int x = ((t >> 6) * CPW + cc) * NV, d = W * tile + x;
      int mode = NV == 1 ? d : d >> 1, j = blockIdx.y + 1;
      cok[cc] = d < ndbl;
      if (T.blocked) {
        const int f = d >> 1, mm = f % T.cw, jl = f / T.cw;
        j = blockIdx.y * T.n2l + jl + 1;
        mode = NV == 1 ? 2 * (mm + T.mofs) + (d & 1) : mm + T.mofs;
        cok[cc] = cok[cc] && mm + T.mofs < T.nmode;
      }
      lamv[cc] = (lamx[cok[cc] ? mode : 0] + lamy[j - 1]) * lscale;      // (never the z-only sweeps without eigenvalues, T.nolam: the host keeps those on the classic form)
    }
    __syncthreads();
    fetch(min(tile + 1, tend - 1));      // in flight during the solve (the last tile again: unused)
#pragma unroll
    for (int cc = 0; cc < CPW; ++cc) {
      const int x = ((t >> 6) * CPW + cc) * NV, ch = t & 63;
      const bool colok = cok[cc];
      if (colok) {
        const real lam = lamv[cc];
        const bool nullc = fixnull && lam == 0.;      // singular mode: the member with p(nz) = 0, see k_gaussel_ri
        real *col = shz + x * P + ch * CP;
        const int k0 = ch * M;
        real cp[M - 1], V[M - 1];      // the swept right-hand sides go back to their LDS slots (registers: no spills at 4 waves/SIMD)
        real cprev = 0., vprev = 0., rprev[NV] = {};
        // periodic z: the last plane's right-hand side (its row is an identity row of the tile), the closure vector's two entries
        real pnr[NV] = {}, E[PER ? M - 1 : 1] = {}, eprev = 0.;
        const real e_first = PER ? -ra[0] : 0., e_last = PER ? -rc[nsys - 1] : 0.;
        if (PER) {
    #pragma unroll
          for (int q = 0; q < NV; ++q) pnr[q] = shz[(x + q) * P + (nz - 1) + (nz - 1) / M];
        }
    #pragma unroll
        for (int r = 0; r < M - 1; ++r) {
          const int k = k0 + r;
          const bool pin = !PER && nullc && k == nz - 1, live = k < nsys && !pin;
          // (every load unconditional, the selection afterwards: a load inside a divergent branch is waited for on the spot -- sixteen serial
          //  round trips to the table per chunk)
          const real A0 = TL ? tabl[r * 64 + ch] : abc[r * 64 + ch], C0 = TL ? tabl[64 * M + r * 64 + ch] : abc[128 * M + r * 64 + ch], B0 = B0r[r];
          const real A = pin ? 0. : A0, C = pin ? 0. : C0, B = pin ? 1. : B0 + (k < nsys ? lam : 0.);
          const real z = rcp_nr(B - A * cprev + CALES_EPS);
          cp[r] = C * z; V[r] = (r == 0 ? A : -A * vprev) * z;
    #pragma unroll
          for (int q = 0; q < NV; ++q) { const real D0 = col[q * P + r], D = live ? D0 : 0.; rprev[q] = (D - A * rprev[q]) * z; col[q * P + r] = rprev[q]; }
          if (PER) { const real D2 = (k == 0 ? e_first : 0.) + (k == nsys - 1 ? e_last : 0.); eprev = (D2 - A * eprev) * z; E[r] = eprev; }
          cprev = cp[r]; vprev = V[r];
        }
        // first interior row of the chunk as a function of the separators beside it
        real Vb = V[M - 2], Wb = cp[M - 2], Rb[NV], Eb = eprev;
    #pragma unroll
        for (int q = 0; q < NV; ++q) Rb[q] = rprev[q];
    #pragma unroll
        for (int r = M - 3; r >= 0; --r) {
          Vb = V[r] - cp[r] * Vb; Wb = -cp[r] * Wb;
    #pragma unroll
          for (int q = 0; q < NV; ++q) Rb[q] = col[q * P + r] - cp[r] * Rb[q];
          if (PER) Eb = E[r] - cp[r] * Eb;
        }
        const bool last = ch == 63;
        // (cross-lane reads are issued by all lanes and masked afterwards: a lane switched off by a branch would be read as zero)
        real Vn = __shfl_down(Vb, 1, 64), Wn = __shfl_down(Wb, 1, 64);
        if (last) { Vn = 0.; Wn = 0.; }
        // separator row
        real al, be, ga, de[NV], de2 = 0.;
        {
          const int k = k0 + M - 1, r = M - 1;
          const bool pin = !PER && nullc && k == nz - 1, live = k < nsys && !pin;
          const real A0 = TL ? tabl[r * 64 + ch] : abc[r * 64 + ch], C0 = TL ? tabl[64 * M + r * 64 + ch] : abc[128 * M + r * 64 + ch], B0 = B0r[r];
          const real A = pin ? 0. : A0, C = pin ? 0. : C0, B = pin ? 1. : B0 + (k < nsys ? lam : 0.);
          al = -A * V[M - 2]; be = B - A * cp[M - 2] - C * Vn; ga = -C * Wn;
          if (PER) {
            real En = __shfl_down(Eb, 1, 64);
            if (last) En = 0.;
            const real D2 = (k == 0 ? e_first : 0.) + (k == nsys - 1 ? e_last : 0.);
            de2 = D2 - A * eprev - C * En;
          }
    #pragma unroll
          for (int q = 0; q < NV; ++q) {
            real Rn = __shfl_down(Rb[q], 1, 64);
            if (last) Rn = 0.;
            const real D0 = col[q * P + M - 1], D = live ? D0 : 0.;
            de[q] = D - A * rprev[q] - C * Rn;
          }
        }
        // parallel cyclic reduction over the 64 separators
    #pragma unroll
        for (int h = 1; h < 64; h <<= 1) {
          const bool lo = ch >= h, hi = ch + h < 64;
          const real rb = rcp_nr(be);
          const real rbm = __shfl_up(rb, h, 64), rbp = __shfl_down(rb, h, 64);
          const real k1 = lo ? al * rbm : 0., k2 = hi ? ga * rbp : 0.;
          const real alm = __shfl_up(al, h, 64), gam = __shfl_up(ga, h, 64), alp = __shfl_down(al, h, 64), gap = __shfl_down(ga, h, 64);
          be = be - gam * k1 - alp * k2;
    #pragma unroll
          for (int q = 0; q < NV; ++q) de[q] = de[q] - __shfl_up(de[q], h, 64) * k1 - __shfl_down(de[q], h, 64) * k2;
          if (PER) de2 = de2 - __shfl_up(de2, h, 64) * k1 - __shfl_down(de2, h, 64) * k2;
          al = -alm * k1; ga = -gap * k2;
        }
        const real rb = rcp_nr(be);
        // periodic z: row n-1 of the system sits in chunk cL at place rL; rows 1 and n-1 of p1 and p2 are fetched from their lanes
        const int cL = PER ? (nsys - 1) / M : 0, rL = PER ? (nsys - 1) % M : 0;
        real Es = 0., e1 = 0., eL = 0.;
        if (PER) {
          Es = de2 * rb;
          real sp2 = __shfl_up(Es, 1, 64);
          if (ch == 0) sp2 = 0.;
          real xv2 = eprev - V[M - 2] * sp2 - cp[M - 2] * Es;
          E[M - 2] = xv2;
    #pragma unroll
          for (int r = M - 3; r >= 0; --r) { xv2 = E[r] - V[r] * sp2 - cp[r] * xv2; E[r] = xv2; }
          real mine = Es;
    #pragma unroll
          for (int r = 0; r < M - 1; ++r) if (r == rL) mine = E[r];
          e1 = __shfl(E[0], 0, 64); eL = __shfl(mine, cL, 64);
        }
    #pragma unroll
        for (int q = 0; q < NV; ++q) {
          const real s = de[q] * rb;
          real sp = __shfl_up(s, 1, 64);
          if (ch == 0) sp = 0.;
          real xv = rprev[q] - V[M - 2] * sp - cp[M - 2] * s;
          real mine = (rL == M - 2) ? xv : s;
          col[q * P + M - 2] = xv;
    #pragma unroll
          for (int r = M - 3; r >= 0; --r) { xv = col[q * P + r] - V[r] * sp - cp[r] * xv; col[q * P + r] = xv; if (PER && r == rL) mine = xv; }
          col[q * P + M - 1] = s;
          if (PER) {
            const real p11 = __shfl(xv, 0, 64), p1n = __shfl(mine, cL, 64);      // xv: row 1 of the chunk (lane 0: row 1 of the system)
            const real an = ra[nz - 1], bn = rb_[nz - 1], cn = rc[nz - 1];
            const real den = (bn + lam) + cn * e1 + an * eL + CALES_EPS;
            const real pn = nullc ? 0. : ((pnr[q] - cn * p11) - an * p1n) * (1. / den);      // null mode of the triply periodic problem: p(n) = 0
    #pragma unroll
            for (int r = 0; r < M - 1; ++r) if (k0 + r < nsys) col[q * P + r] = col[q * P + r] + E[r] * pn;
            if (k0 + M - 1 < nsys) col[q * P + M - 1] = s + Es * pn;
            if (ch == nsys / M) col[q * P + nsys % M] = pn;
          }
        }
      }
    }
    __syncthreads();
    {
      size_t o = seg + (size_t)W * tile + xl + (size_t)kk * kst;
      int kq = kk;      // (formed again behind an opaque move, as in the classic form: the places of the load phase would otherwise stay alive through the solve)
      asm volatile("" : "+v"(o), "+v"(kq));
      const size_t step = (size_t)KP2 * kst;
#pragma unroll
      for (int q = 0; q < NQ2; ++q) {
        const int k = kq + KP2 * q;
        *reinterpret_cast<real2 *>(p + o) = make_real2(shz[xl * P + k + k / M], shz[(xl + 1) * P + k + k / M]);
        o += step;
      }
    }
    __syncthreads();      // the tile has left the LDS before the next one is written
  };
  if (tile0 < tend) one_tile(tile0);
  for (int tile = tile0 + 1; tile < tend; ++tile) one_tile(tile);
}
template <int M, int NV, int PER = 0>
static void launch_gaussel_tile(cales_ctx *c, int nz, int ndbl, int nrow, real lscale, const real *da, const real *db, const real *dc,
                                real *p, int fixnull, const TileMap &T, real *tab_of_caller = nullptr, bool tab_ready = false) {
  constexpr int W = gt_width(M, NV), lds = W * (64 * (M + 1) + GT_PAD) * 8 + (GT_TL && M == 16 ? 2 * 64 * M * 8 : 0);
  static bool once = false;
  if (!once) {
    HIPSOFT(c, hipFuncSetAttribute((const void *)k_gaussel_tile<M, NV, PER>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    if constexpr (NV == 2 && PER == 0) HIPSOFT(c, hipFuncSetAttribute((const void *)k_gaussel_tile<M, NV, PER, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    once = true;
  }
  // (a failed allocation fails the context: launch_failed is sticky, op_solver's LAUNCHCHK returns it -- the z solve is never skipped silently)
  if (!c->d_abct) { const hipError_t e = hipMalloc(&c->d_abct, 2 * 3 * 64 * 16 * sizeof(real)); if (e != hipSuccess) { c->d_abct = nullptr; launch_failed(c, "hipMalloc(tridiagonal coefficient table)", e); return; } }
  // the pressure operands never change: their table is built once (Helmholtz operands are rescaled every substep -> second table)
  const bool pressure = da == c->d_a;
  real *tab = tab_of_caller ? tab_of_caller : c->d_abct + (pressure ? 0 : 3 * 64 * 16);
  // (periodic z: the table holds the n-1 rows of the system proper; identity rows from row n on)
  if (tab_of_caller ? !tab_ready : (!pressure || !c->abct_ready)) LAUNCH(c, k_abc_chunked, dim3((64 * M + 255) / 256), dim3(256), 0, c->stream, PER ? nz - 1 : nz, M, da, db, dc, tab);
  if (pressure && !tab_of_caller) c->abct_ready = true;
  if constexpr (NV == 2 && PER == 0) {
    if (T.dud) { LAUNCH(c, (k_gaussel_tile<M, NV, PER, 1>), dim3((ndbl + W - 1) / W, nrow), dim3(64 * W / NV), lds, c->stream, c->g, nz, ndbl, lscale, tab, c->d_lamx, c->d_lamy, p, fixnull, T, da, db, dc); return; }
  }
  if (T.dud) { launch_failed(c, "k_gaussel_tile: the Helmholtz form exists for pairs of real columns and non-periodic z only", hipErrorInvalidValue); return; }
  LAUNCH(c, (k_gaussel_tile<M, NV, PER>), dim3((ndbl + W - 1) / W, nrow), dim3(64 * W / NV), lds, c->stream, c->g, nz, ndbl, lscale, tab,
                     c->d_lamx, c->d_lamy, p, fixnull, T, da, db, dc);
}
// the persistent form (k_gaussel_tile_p): pressure operands only (their table is built once), tiles per block so that ~2048 blocks or more remain
template <int NV>
static void launch_gaussel_tile_p(cales_ctx *c, int nz, int ndbl, int nrow, real lscale, const real *da, const real *db, const real *dc, real *p, int fixnull, const TileMap &T) {
  constexpr int M = 16, W = gt_width(M, NV), lds = W * (64 * (M + 1) + GT_PAD) * 8 + (GT_TL ? 2 * 64 * M * 8 : 0);
  static bool once = false;
  if (!once) { HIPSOFT(c, hipFuncSetAttribute((const void *)k_gaussel_tile_p<NV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); once = true; }
  if (!c->d_abct) { const hipError_t e = hipMalloc(&c->d_abct, 2 * 3 * 64 * 16 * sizeof(real)); if (e != hipSuccess) { c->d_abct = nullptr; launch_failed(c, "hipMalloc(tridiagonal coefficient table)", e); return; } }
  const bool pressure = da == c->d_a;
  real *tab = c->d_abct + (pressure ? 0 : 3 * 64 * 16);
  if (!pressure || !c->abct_ready) LAUNCH(c, k_abc_chunked, dim3((64 * M + 255) / 256), dim3(256), 0, c->stream, nz, M, da, db, dc, tab);
  if (pressure) c->abct_ready = true;
  const int ntile = ndbl / W;
  int tpb = 1; while (tpb * 2 <= ntile && tpb < 32 && (long)nrow * ((ntile + 2 * tpb - 1) / (2 * tpb)) >= 2048) tpb *= 2;
  LAUNCH(c, (k_gaussel_tile_p<NV>), dim3((ntile + tpb - 1) / tpb, nrow), dim3(512), lds, c->stream, c->g, nz, ndbl, lscale, tab, c->d_lamx, c->d_lamy, p, fixnull, T, tpb, da, db, dc);
}
// one rank: ndbl doubles of each of the nrow rows; several ranks (T.blocked): nrow = peers, ndbl = 2 cw n2l doubles per plane of a peer block
template <int NV, int PER = 0>
static bool gaussel_tile(cales_ctx *c, int nz, int ndbl, int nrow, real lscale, const real *da, const real *db, const real *dc,
                         real *p, int fixnull, const TileMap &T, real *tab = nullptr, bool tab_ready = false) {
  if (nz < (PER ? 4 : 2) || nz > 1024 || c->fl.gaussel_march) return false;
  if (nz <= 128) launch_gaussel_tile<2, NV, PER>(c, nz, ndbl, nrow, lscale, da, db, dc, p, fixnull, T, tab, tab_ready);
  else if (nz <= 256) launch_gaussel_tile<4, NV, PER>(c, nz, ndbl, nrow, lscale, da, db, dc, p, fixnull, T, tab, tab_ready);
  else if (nz <= 512) launch_gaussel_tile<8, NV, PER>(c, nz, ndbl, nrow, lscale, da, db, dc, p, fixnull, T, tab, tab_ready);
  else if (!PER && NV == 1 && nz == 1024 && ndbl % 16 == 0 && !T.dud && !T.nolam && !tab) launch_gaussel_tile_p<1>(c, nz, ndbl, nrow, lscale, da, db, dc, p, fixnull, T);
  else launch_gaussel_tile<16, NV, PER>(c, nz, ndbl, nrow, lscale, da, db, dc, p, fixnull, T, tab, tab_ready);
  return true;
}

// Non-periodic x (real modes r = 2m, 2m+1 paired into one complex column) with PERIODIC y: the y transform of the pair gives
// Z_ky = A_ky + i B_ky, A and B the (Hermitian) spectra of the two real columns, which have DIFFERENT x eigenvalues. Rows ky and
// N-ky are separated, A = (Z_ky + conj Z_{N-ky})/2, B = (Z_ky - conj Z_{N-ky})/(2i), the four real systems (Re/Im of A and B) go to
// four neighbouring lanes, and the rows are rebuilt as Z_ky = a + i b, Z_{N-ky} = conj(a) + i conj(b). In place: every lane keeps
// its intermediate values in one of the four slots (Re/Im of the two rows) and finally overwrites that slot.
template <int CTRL> __device__ inline real quad_bcast(real v) {
#ifdef CALES_SINGLE
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
#else
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
#endif
}
__global__ __launch_bounds__(256) void k_gaussel_herm(Geom g, int nz, int ncol, int N, int mofs, int nmode, Spec S, real lscale,
                                                      const real *__restrict__ a, const real *__restrict__ b,
                                                      const real *__restrict__ c, const real *__restrict__ lamx,
                                                      const real *__restrict__ lamy, real *__restrict__ p, real *__restrict__ dscr, int fixnull) {
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  const int t = (int)(q % (4 * ncol)), m = t >> 2, role = t & 3, ky = (int)(q / (4 * ncol));     // role: Re a, Im a, Re b, Im b
  const bool live = ky <= N / 2 && m + mofs < nmode;
  const int kn = (N - ky) % N; const bool self = kn == ky;
  // real offsets of plane 1: components of rows ky+1 and kn+1 (at_mode rows are 1-based)
  const size_t ek = live ? 2 * S.at_mode(g, m, ky + 1, 1) : 0, en = live ? 2 * S.at_mode(g, m, kn + 1, 1) : 0;
  const size_t st = 2 * (S.blocked ? (size_t)S.cw * S.n2l : (size_t)g.s12 / 2);
  const size_t s0 = (size_t)t + (size_t)4 * ncol * (size_t)ky, sst = (size_t)4 * ncol * (N / 2 + 1);
  const real lam = live ? (lamx[2 * (m + mofs) + (role >> 1)] + lamy[ky]) * lscale : 1.;      // lscale: alpha of a Helmholtz solve (lambdaxy*alpha, main.f90:438)
  // my slot: role 0 -> Re row ky, 1 -> Im row ky, 2 -> Re row kn, 3 -> Im row kn; self-conjugate rows: role 2 -> Im row ky, roles 1,3 idle
  const bool active = live && !(self && (role & 1));
  const size_t slot = self ? (role == 0 ? ek : ek + 1) : (role == 0 ? ek : role == 1 ? ek + 1 : role == 2 ? en : en + 1);
  auto rhs = [&](int l) -> real {
    if (!live) return 0.;
    const real zrk = p[ek + l * st], zik = p[ek + 1 + l * st], zrn = p[en + l * st], zin = p[en + 1 + l * st];
    return role == 0 ? 0.5 * (zrk + zrn) : role == 1 ? 0.5 * (zik - zin) : role == 2 ? 0.5 * (zik + zin) : 0.5 * (zrn - zrk);
  };
  real z = 1. / (b[0] + lam + CALES_EPS), d = c[0] * z;
  real v = rhs(0) * z;
  if (active) { p[slot] = v; dscr[s0] = d; }
  for (int l = 1; l < nz; ++l) {
    z = 1. / ((b[l] + lam) - a[l] * d + CALES_EPS);
    d = c[l] * z;
    v = (rhs(l) - a[l] * v) * z;
    if (l == nz - 1 && fixnull && lam == 0.) v = 0.;       // null mode: see k_gaussel_ri
    if (active) { p[slot + l * st] = v; dscr[s0 + l * sst] = d; }
  }
  // back substitution; after each plane the quad holds (Re a, Im a, Re b, Im b) and rebuilds the two rows
  for (int l = nz - 1; l >= 0; --l) {
    if (l < nz - 1) v = active ? p[slot + l * st] - dscr[s0 + l * sst] * v : 0.;
    else if (!active) v = 0.;
    const real ra = quad_bcast<0x00>(v), ia = quad_bcast<0x55>(v), rb = quad_bcast<0xAA>(v), ib = quad_bcast<0xFF>(v);
    const real out = self ? (role == 0 ? ra : rb) : (role == 0 ? ra - ib : role == 1 ? ia + rb : role == 2 ? ra + ib : rb - ia);
    if (active) p[slot + l * st] = out;
  }
}

template <typename VT, int PERIODIC>
__global__ __launch_bounds__(256) void k_gaussel(Geom g, int nz, int ncol, int nrow, int i0, int mofs, int nmode, Spec S, real lscale,
                                                 const real *__restrict__ a, const real *__restrict__ b,
                                                 const real *__restrict__ c, const real *__restrict__ lamx,
                                                 const real *__restrict__ lamy, real *__restrict__ pd, real *__restrict__ dscr,
                                                 real *__restrict__ p2scr, int fixnull) {
  const int m = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y + 1;
  if (m >= ncol || j > nrow || m + mofs >= nmode) return;
  constexpr int W = sizeof(VT) / sizeof(real);
  VT *p = reinterpret_cast<VT *>(pd);
  const size_t e0 = W == 2 ? S.at_mode(g, m, j, 1) : g.ix(m + i0, j, 1);    // element index of k=1 in units of VT
  const size_t st = W == 2 ? (S.blocked ? (size_t)S.cw * S.n2l : (size_t)g.s12 / 2) : (size_t)g.s12;
  const size_t s0 = (size_t)m + (size_t)ncol * (size_t)(j - 1), sst = (size_t)ncol * nrow;   // scratch [k][j][m]
  const real lam = ((lamx ? lamx[m + mofs] : 0.) + (lamy ? lamy[j - 1] : 0.)) * lscale;
  const int n = PERIODIC ? nz - 1 : nz;
  // forward elimination
  real z = 1. / (b[0] + lam + CALES_EPS), d = c[0] * z;
  VT v = vmul(p[e0], z);
  p[e0] = v; dscr[s0] = d;
  real v2 = 0.;
  if (PERIODIC) { v2 = (n == 1 ? -c[0] : -a[0]) * z; p2scr[s0] = v2; }
  for (int l = 1; l < n; ++l) {
    const real bb = b[l] + lam;
    z = 1. / (bb - a[l] * d + CALES_EPS);
    d = c[l] * z;
    v = vmul(vfms(p[e0 + l * st], a[l], v), z);
    if (!PERIODIC && l == n - 1 && fixnull && lam == 0.) v = vmul(v, 0.);      // null mode: see k_gaussel_ri
    p[e0 + l * st] = v; dscr[s0 + l * sst] = d;
    if (PERIODIC) { const real r2 = (l == n - 1) ? -c[n - 1] : 0.; v2 = (r2 - a[l] * v2) * z; p2scr[s0 + l * sst] = v2; }
  }
  // back substitution
  for (int l = n - 2; l >= 0; --l) {
    v = vfms(p[e0 + l * st], dscr[s0 + l * sst], v);
    p[e0 + l * st] = v;
    if (PERIODIC) { v2 = p2scr[s0 + l * sst] - dscr[s0 + l * sst] * v2; p2scr[s0 + l * sst] = v2; }
  }
  if (PERIODIC) {   // solver.f90:124-133
    const VT p11 = p[e0], p1n = p[e0 + (size_t)(n - 1) * st];
    const real p21 = p2scr[s0], p2n = p2scr[s0 + (size_t)(n - 1) * sst];
    const real den = (b[nz - 1] + lam) + c[nz - 1] * p21 + a[nz - 1] * p2n + CALES_EPS;
    VT pn = vfms(vfms(p[e0 + (size_t)(nz - 1) * st], c[nz - 1], p11), a[nz - 1], p1n);
    pn = vmul(pn, (fixnull && lam == 0.) ? 0. : 1. / den);      // null mode of the triply periodic problem: p(n) = 0
    p[e0 + (size_t)(nz - 1) * st] = pn;
    for (int l = 0; l < n; ++l) p[e0 + l * st] = vfma(p[e0 + l * st], p2scr[s0 + l * sst], pn);
  }
}

// ------------------------------------------------------------------------------------------ host side
struct SolverPlans { FftPlan py4; int CBy4; size_t shy4; FftPlan px, py; int Rx, CBy; size_t shx, shy; bool x8, y8, y16 = false; int x8_threads, y8_threads; size_t shx8, shy8, shy8r, shy16 = 0; };
struct VelSet { bool ready = false; int xkind = 0, ykind = 0; real *lamx = nullptr, *lamy = nullptr; real normfft = 1.; FftPlan p1x, p1y; real *tw1x = nullptr, *tw1y = nullptr; };
struct PlanSlot { cales_ctx *ctx; SolverPlans sp; VelSet vs[3]; };
// one entry per context; a list (stable addresses) behind a mutex: contexts are created and destroyed from several host threads in the
// loopback tests while others hold pointers to their own entry
static std::list<PlanSlot> g_slots;
static std::mutex g_slots_mx;
static PlanSlot *find_slot(cales_ctx *c) { std::lock_guard<std::mutex> lk(g_slots_mx); for (auto &s : g_slots) if (s.ctx == c) return &s; return nullptr; }
static SolverPlans *find_plans(cales_ctx *c) { PlanSlot *s = find_slot(c); return s ? &s->sp : nullptr; }

bool solver_can_fuse_fillps(cales_ctx *c) { SolverPlans *sp = find_plans(c); return sp && sp->x8; }

int solver_setup(cales_ctx *c) {
  const int *n = c->n; const int n1 = c->C.ng[0], n2g = c->C.ng[1], n3 = n[2];
  const std::string bx = std::string(1, c->C.cbcpre[0]) + c->C.cbcpre[1], by = std::string(1, c->C.cbcpre[2]) + c->C.cbcpre[3];
  // transform kinds of the cell-centred pressure (fft.f90:192-245): 0 PP (R2HC/HC2R), 1 NN (REDFT10/01), 2 DD (RODFT10/01), 3 ND (REDFT11), 4 DN (RODFT11)
  auto kind_of = [](const std::string &b) { return b == "PP" ? 0 : b == "NN" ? 1 : b == "DD" ? 2 : b == "ND" ? 3 : b == "DN" ? 4 : -1; };
  c->xkind = kind_of(bx); c->ykind = kind_of(by);
  if (c->xkind < 0 || c->ykind < 0) { c->err = "solver: unknown pressure BC pair in x or y"; return 1; }
  if (c->ykind >= 3 && (n2g % 2)) { c->err = "solver: ND/DN in y need an even ng(2)"; return 1; }
  if (c->xkind && c->C.cbcpre[4] == 'P' && (!c->ykind || c->P > 1)) { c->err = "solver: a non-periodic x with periodic z needs a non-periodic y and one rank"; return 1; }
  SolverPlans sp;
  if (!make_plan(n1 / 2, sp.px) || !make_plan(n2g, sp.py)) { c->err = "solver: ng(1)/2 and ng(2) must factor into primes <= 127"; return 1; }
  // rows per block in x: aim at ~nh/4 threads per row, 256 threads per block
  { int T = std::max(1, std::min(256, (n1 / 2) / 4)); int p2 = 1; while (p2 * 2 <= T) p2 *= 2; T = p2; sp.Rx = 256 / T; }
  sp.shx = (size_t)sp.Rx * 2 * (n1 / 2 + 1) * sizeof(cpx);
  while (sp.shx > 60 * 1024 && sp.Rx > 1) { sp.Rx /= 2; sp.shx = (size_t)sp.Rx * 2 * (n1 / 2 + 1) * sizeof(cpx); }
  sp.CBy = 8;
  sp.shy = (size_t)sp.CBy * 2 * (n2g + 1) * sizeof(cpx);
  while (sp.shy > 60 * 1024 && sp.CBy > 1) { sp.CBy /= 2; sp.shy = (size_t)sp.CBy * 2 * (n2g + 1) * sizeof(cpx); }
  if (sp.shx > 64 * 1024 || sp.shy > 64 * 1024) { c->err = "solver: line too long for the LDS-resident transform"; return 1; }
  // power-of-two lines take the radix-8 register kernels
  auto pow2 = [](int v) { return v >= 16 && (v & (v - 1)) == 0; };
  sp.x8 = pow2(n1 / 2) && n1 / 2 <= 1024 && c->xkind <= 1; sp.y8 = pow2(n2g) && n2g <= 1024;
  if (sp.x8) { const int T = (n1 / 2) / 8; sp.x8_threads = T >= 256 ? T : (256 / T) * T;
               sp.shx8 = ((size_t)(sp.x8_threads / T) * (n1 / 2 + n1 / 16 + 2) + (n1 + 1) + (c->xkind ? n1 / 2 + 1 : 0)) * sizeof(cpx); }
  if (sp.y8) { const int T = n2g / 8; int CB = std::max(1, std::min(std::max(8, 256 / T), 512 / T));
               while (CB > 1 && ((size_t)CB * (n2g + n2g / 8 + 1) + n2g) * sizeof(cpx) > 150 * 1024) CB /= 2;
               sp.y8_threads = CB * T; sp.shy8 = ((size_t)CB * (n2g + n2g / 8 + 1) + n2g) * sizeof(cpx);
               sp.shy8r = ((size_t)CB * ((((n2g + n2g / 8) + 15) & ~15) + 4) + n2g) * sizeof(cpx);      // k_fft_y8r: line pitch = 4 (mod 16) slots
               if (sp.shy8r > 64 * 1024) { HIPSOFT(c, hipFuncSetAttribute((const void *)k_fft_y8r<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp.shy8r));
                                           HIPSOFT(c, hipFuncSetAttribute((const void *)k_fft_y8r<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp.shy8r)); }
               if (sp.shy8 > 64 * 1024) {      // n2 = 1024: 4 columns (64-B row segments) need 90 KB of LDS
                 HIPSOFT(c, hipFuncSetAttribute((const void *)k_fft_y8<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp.shy8));
                 HIPSOFT(c, hipFuncSetAttribute((const void *)k_fft_y8<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp.shy8)); } }
  // eight columns per block, sixteen elements per thread (k_fft_y16): 1024-point lines of either kind, 256- and 512-point Neumann lines (where the
  // alternative is the staged k_fft_y8: 512 x 256 x 256 duct 0.39 / 0.42 -> 0.30 / 0.31 ms per step). Periodic 512-point lines stay with k_fft_y8r:
  // measured 1.49 / 1.55 against 1.45 / 1.52 ms per step at 512^3 -- both sit at the copy rate of that access pattern (128-B segments at a pitch of 4224 B)
  sp.y16 = sp.y8 && c->ykind <= 1 && (n2g == 1024 || (c->ykind == 1 && (n2g == 512 || n2g == 256)));
  if (sp.y16) {
    sp.shy16 = ((size_t)n2g + 8 * ((size_t)n2g + 1)) * sizeof(cpx) + (size_t)n2g * sizeof(unsigned);
    if (sp.shy16 > 64 * 1024) {
#define Y16_ATTR(NN) do { HIPSOFT(c, hipFuncSetAttribute((const void *)k_fft_y16<NN, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp.shy16)); \
                          HIPSOFT(c, hipFuncSetAttribute((const void *)k_fft_y16<NN, 1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp.shy16)); \
                          HIPSOFT(c, hipFuncSetAttribute((const void *)k_fft_y16<NN, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp.shy16)); \
                          HIPSOFT(c, hipFuncSetAttribute((const void *)k_fft_y16<NN, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp.shy16)); } while (0)
      if (n2g == 1024) Y16_ATTR(1024); else Y16_ATTR(512);
#undef Y16_ATTR
    }
  }
  if (c->ykind >= 3) {      // DCT-IV / DST-IV in y: N/2-point lines, two per complex column
    if (!make_plan(n2g / 2, sp.py4)) { c->err = "solver: ng(2)/2 must factor into primes <= 127"; return 1; }
    sp.CBy4 = 4; sp.shy4 = (size_t)2 * sp.CBy4 * 2 * (n2g / 2 + 1) * sizeof(cpx);
    while (sp.shy4 > 60 * 1024 && sp.CBy4 > 1) { sp.CBy4 /= 2; sp.shy4 = (size_t)2 * sp.CBy4 * 2 * (n2g / 2 + 1) * sizeof(cpx); }
    if (sp.shy4 > 64 * 1024) { c->err = "solver: y line too long for the LDS-resident DCT-IV"; return 1; }
    sp.y8 = false;
    std::vector<real> t(4 * (size_t)(n2g / 2)); const real pi = std::acos(-1.0); const int nh2 = n2g / 2;
    for (int q = 0; q < nh2; ++q) { const real a1 = -pi * (4. * q + 1.) / (4. * n2g), a2 = -pi * q / (real)n2g;
                                    t[2 * q] = std::cos(a1); t[2 * q + 1] = std::sin(a1); t[2 * (nh2 + q)] = std::cos(a2); t[2 * (nh2 + q) + 1] = std::sin(a2); }
    HIPCHK(c, hipMalloc(&c->d_tw4y, t.size() * sizeof(real)));
    HIPCHK(c, hipMemcpy(c->d_tw4y, t.data(), t.size() * sizeof(real), hipMemcpyHostToDevice));
  }
  if (c->ykind == 2) sp.y8 = false;      // the sign changes of the Dirichlet-Dirichlet transform live in the generic y kernel only
  if (c->fl.fft_generic) sp.x8 = sp.y8 = false;
  if (!sp.y8) sp.y16 = false;
  // eigenvalues (initsolver.f90:66-98); x: modes 0..n1/2 (half-complex symmetry), y: modes 0..n2-1
  std::vector<real> lx(n1 + 2, 0.), ly(n2g);
  hs_eigenvalues(n1, bx.c_str(), 'c', lx.data()); hs_eigenvalues(n2g, by.c_str(), 'c', ly.data());
  for (auto &v : lx) v = v * (c->dli[0] * c->dli[0]);
  for (auto &v : ly) v = v * (c->dli[1] * c->dli[1]);
  if (c->ykind == 2) std::reverse(ly.begin(), ly.end());      // row r of the transformed field holds coefficient N-1-r (see k_fft_y)
  const int mh = n1 / 2 + 1;
  c->cw = (mh + c->P - 1) / c->P;                            // complex mode columns per rank (last block padded)
  // rows of the mode-block layout in whole 128-B lines (cw a multiple of 8) where the padding columns -- they travel in the all-to-all -- add 6 % or less: the y
  // transforms and the z tile then move whole lines instead of segments that straddle two. Measured on rank 0 of 2 of the 512^3 channel (cw 129 -> 136, +5 % to
  // send): y passes 2.75 -> 1.44 ms per step, z sweep 1.12 -> 0.86, rank compute 20.5 -> 18.8; of 8 (33 -> 40, +21 %): compute 5.38 -> 5.29 for 0.14 ms more
  // on the links -- not padded there.
  if (c->P > 1) { const int cw8 = (c->cw + 7) / 8 * 8; if (100 * (cw8 - c->cw) <= 6 * c->cw && (size_t)cw8 * n2g * n3 <= c->ntot) c->cw = cw8; }
  if (c->P > 1 && (size_t)c->cw * n2g * n3 > c->ntot) { c->err = "solver: scratch too small for the mode-block layout"; return 1; }
  // Nyquist packing of the pressure solve (k_gaussel_nyq): periodic x, periodic or Neumann y (the pair is separated by the Hermitian pairing of rows / by real and
  // imaginary part), radix-8 passes, the z solve in the LDS tile; n1/2 mode columns instead of n1/2 + 1
  { const bool pz = CBP(c, 0, 3) == 'P' && CBP(c, 1, 3) == 'P', hasd = CBP(c, 0, 3) == 'D' || CBP(c, 1, 3) == 'D';
    c->nyq_ok = c->xkind == 0 && c->ykind <= 1 && sp.x8 && sp.y8 && !pz && (hasd || !c->fl.keep_null_mode) && n3 >= 2 && n3 <= 1024 && !c->fl.gaussel_march && !c->fl.no_nyquist_packing;
    c->cw_nyq = (n1 / 2 + c->P - 1) / c->P; }
  HIPCHK(c, hipMalloc(&c->d_lamx, (n1 + 2) * sizeof(real))); HIPCHK(c, hipMalloc(&c->d_lamy, n2g * sizeof(real)));
  HIPCHK(c, hipMemcpy(c->d_lamx, lx.data(), (n1 + 2) * sizeof(real), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_lamy, ly.data(), n2g * sizeof(real), hipMemcpyHostToDevice));
  // tridiagonal (initsolver.f90:127-169), pressure: cell-centred
  std::vector<real> a(n3), b(n3), cc(n3);
  hs_tridmatrix(&c->C.cbcpre[4], n3, c->dzci.data(), c->dzfi.data(), 'c', a.data(), b.data(), cc.data());
  HIPCHK(c, hipMalloc(&c->d_a, n3 * sizeof(real))); HIPCHK(c, hipMalloc(&c->d_b, n3 * sizeof(real))); HIPCHK(c, hipMalloc(&c->d_c, n3 * sizeof(real)));
  HIPCHK(c, hipMemcpy(c->d_a, a.data(), n3 * sizeof(real), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_b, b.data(), n3 * sizeof(real), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_c, cc.data(), n3 * sizeof(real), hipMemcpyHostToDevice));
  c->normfft = 1. / ((c->xkind ? 2. : 1.) * (real)n1 * (c->ykind ? 2. : 1.) * (real)n2g);   // fft.f90:99,136,142; find_fft norm = [1,0] (PP) / [2,0] (NN)
  // twiddles: exp(-2 pi i q/N)
  auto mk = [&](int N, int cnt, real **dev) -> int {
    std::vector<real> t(2 * (size_t)cnt);
    const real pi = std::acos(-1.0);
    for (int q = 0; q < cnt; ++q) { const real ang = -2. * pi * q / N; t[2 * q] = std::cos(ang); t[2 * q + 1] = std::sin(ang); }
    HIPCHK(c, hipMalloc(dev, t.size() * sizeof(real)));
    HIPCHK(c, hipMemcpy(*dev, t.data(), t.size() * sizeof(real), hipMemcpyHostToDevice));
    return 0;
  };
  if (mk(n1 / 2, n1 / 2, &c->d_twx)) return 1;
  if (mk(n1, n1 / 2 + 1, &c->d_twx_post)) return 1;
  if (mk(n2g, n2g, &c->d_twy)) return 1;
  if (mk(4 * n1, n1 / 2 + 1, &c->d_twy_post)) return 1;      // DCT weights e^{-i pi k/(2 n1)}, k = 0..n1/2 (x)
  if (mk(4 * n2g, n2g, &c->scr_twyd)) return 1;              // e^{-i pi k/(2 n2)}, k = 0..n2-1 (y)
  if (c->ykind >= 3) { if (mk(n2g / 2, n2g / 2, &c->d_twy4)) return 1; }      // N/2-point lines of k_fft_y4
  if (c->xkind >= 3) {            // DCT-IV weights (k_fft_x4)
    const int nh = n1 / 2; std::vector<real> t(4 * (size_t)nh); const real pi = std::acos(-1.0);
    for (int q = 0; q < nh; ++q) { const real a1 = -pi * (4. * q + 1.) / (4. * n1), a2 = -pi * q / (real)n1;
                                   t[2 * q] = std::cos(a1); t[2 * q + 1] = std::sin(a1); t[2 * (nh + q)] = std::cos(a2); t[2 * (nh + q) + 1] = std::sin(a2); }
    HIPCHK(c, hipMalloc(&c->d_tw4x, t.size() * sizeof(real)));
    HIPCHK(c, hipMemcpy(c->d_tw4x, t.data(), t.size() * sizeof(real), hipMemcpyHostToDevice));
  }
  if (c->C.impdiff)
    for (int iv = 0; iv < 3; ++iv) {
      hs_tridmatrix(&c->cbcvel[6 * iv + 4], n3, c->dzci.data(), c->dzfi.data(), iv == 2 ? 'f' : 'c', a.data(), b.data(), cc.data());
      HIPCHK(c, hipMalloc(&c->d_av[iv], 3 * n3 * sizeof(real)));   // a | b | c (unscaled); scaled copies follow
      c->d_bv[iv] = c->d_av[iv] + n3; c->d_cv[iv] = c->d_av[iv] + 2 * n3;
      HIPCHK(c, hipMemcpy(c->d_av[iv], a.data(), n3 * sizeof(real), hipMemcpyHostToDevice));
      HIPCHK(c, hipMemcpy(c->d_bv[iv], b.data(), n3 * sizeof(real), hipMemcpyHostToDevice));
      HIPCHK(c, hipMemcpy(c->d_cv[iv], cc.data(), n3 * sizeof(real), hipMemcpyHostToDevice));
    }
  { PlanSlot ps_; ps_.ctx = c; ps_.sp = sp; std::lock_guard<std::mutex> lk(g_slots_mx); g_slots.push_back(ps_); }
  return 0;
}
void solver_teardown(cales_ctx *c) {
  { std::lock_guard<std::mutex> lk(g_slots_mx);
    for (auto it = g_slots.begin(); it != g_slots.end(); ++it) if (it->ctx == c) {
      for (auto &V : it->vs) { if (V.lamx) hipFree(V.lamx); if (V.lamy) hipFree(V.lamy); if (V.tw1x) hipFree(V.tw1x); if (V.tw1y) hipFree(V.tw1y); }
      g_slots.erase(it); break; } }
  hipFree(c->d_lamx); hipFree(c->d_lamy); hipFree(c->d_a); hipFree(c->d_b); hipFree(c->d_c);
  hipFree(c->d_twx); hipFree(c->d_twx_post); hipFree(c->d_twy); hipFree(c->d_twy_post); hipFree(c->scr_twyd); hipFree(c->d_tw4x); if (c->d_tw4y) hipFree(c->d_tw4y); if (c->d_twy4) hipFree(c->d_twy4);
  for (int iv = 0; iv < 3; ++iv) hipFree(c->d_av[iv]);
}

// CALES_KEEP_NULL_MODE: the zero-eigenvalue column of a singular pressure problem (no Dirichlet condition in z) solved in the REFERENCE's own
// sequential order -- dgtsv_homebrewed / gaussel_periodic with their +eps pivots, solver.f90:109-179, one operation at a time, no contraction.
// Its solution is a constant C times the null vector plus a regular part, with C = (a sum that vanishes for a compatible r.h.s.) / (a pivot
// that vanishes but for round-off and eps): both are defined by the order of the operations, so only the same order reproduces the reference's C
// (1e4..1e7 on triply periodic boxes whose default-real grid arithmetic, initgrid.f90:63, leaves dzf non-uniform at 1e-8: there the r.h.s. is
// incompatible at that level and the velocities inherit C times the round-off of the null vector). The substructured sweeps solve every other
// column (regular: any order gives the same result to round-off); this one is saved before them (phase 0) and overwritten after (phase 1).
// One thread; w holds 5 nz reals.
// (`#pragma clang fp contract(off)` in every body: HIP compiles with -ffp-contract=fast, and __dadd_rn / __dmul_rn are plain operators there --
// a fused multiply-add rounds once where the reference rounds twice, and the sign of a pivot of +-eps hangs on that bit)
__device__ inline real r_add(real a, real b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ inline real r_mul(real a, real b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ inline void ref_dgtsv(int n, const real *a, const real *b, const real *c, real *p, real *d) {      // solver.f90:153-179
#pragma clang fp contract(off)
  const real eps = CALES_EPS;
  real z = (real)1. / r_add(b[0], eps);
  d[0] = r_mul(c[0], z); p[0] = r_mul(p[0], z);
  for (int l = 1; l < n; ++l) {
    z = (real)1. / r_add(r_add(b[l], -r_mul(a[l], d[l - 1])), eps);
    d[l] = r_mul(c[l], z);
    p[l] = r_mul(r_add(p[l], -r_mul(a[l], p[l - 1])), z);
  }
  for (int l = n - 2; l >= 0; --l) p[l] = r_add(p[l], -r_mul(d[l], p[l + 1]));
}
__global__ void k_null_column(Geom g, Spec S, int nz, int periodic, const real *__restrict__ a, const real *__restrict__ b, const real *__restrict__ c,
                              const real *__restrict__ lamx, const real *__restrict__ lamy, real2 *__restrict__ p, real *__restrict__ w, int phase) {
#pragma clang fp contract(off)
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const real lam = lamx[0] + lamy[0];
  if (lam != 0.) return;      // x or y carries a Dirichlet condition: nothing is singular
  real *bb = w, *d = w + nz, *p1 = w + 2 * nz, *p2 = w + 3 * nz, *col = w + 4 * nz;
  if (phase == 0) { for (int l = 0; l < nz; ++l) col[l] = p[S.at_mode(g, 0, 1, l + 1)].x; return; }
  for (int l = 0; l < nz; ++l) bb[l] = r_add(b[l], lam);
  if (!periodic) ref_dgtsv(nz, a, bb, c, col, d);
  else {      // gaussel_periodic, solver.f90:109-150
    const int n = nz;
    for (int l = 0; l < n - 1; ++l) p1[l] = col[l];
    ref_dgtsv(n - 1, a, bb, c, p1, d);
    for (int l = 0; l < n; ++l) p2[l] = 0.;
    p2[0] = -a[0]; p2[n - 2] = -c[n - 2];
    ref_dgtsv(n - 1, a, bb, c, p2, d);
    const real num = r_add(r_add(col[n - 1], -r_mul(c[n - 1], p1[0])), -r_mul(a[n - 1], p1[n - 2]));
    const real den = r_add(r_add(r_add(bb[n - 1], r_mul(c[n - 1], p2[0])), r_mul(a[n - 1], p2[n - 2])), (real)CALES_EPS);
    const real pn = num / den;
    col[n - 1] = pn;
    for (int l = 0; l < n - 1; ++l) col[l] = r_add(p1[l], r_mul(p2[l], pn));
  }
  for (int l = 0; l < nz; ++l) p[S.at_mode(g, 0, 1, l + 1)].x = col[l];
}

// FFT x, FFT y, tridiagonal z, and back, in place on `pp`; (da,db,dc,nz,lscale) = (a,b,c,n3,1) for the pressure Poisson equation,
// (alpha a, alpha b + 1, alpha c, n3 - q, alpha) for the Helmholtz equation of a velocity component (main.f90:435-445)
static int solve_field(cales_ctx *c, real *pp, const real *da, const real *db, const real *dc, int nz, real lscale, bool periodic_z, bool poisson) {
  SolverPlans *sp = find_plans(c);
  if (!sp) { c->err = "solver not initialised"; return 1; }
  const int *n = c->n;
  const int mh = n[0] / 2 + 1, n2g = c->C.ng[1];
  const long nrows = (long)n[1] * n[2];
  const bool dist = c->P > 1;
  // the radix-8 kernels know the periodic and the Neumann-Neumann transform; a velocity component of the 3-D implicit step may need others
  const bool use8x = sp->x8 && c->xkind <= 1, use8y = sp->y8 && c->ykind <= 1;
  const VelSet *VS = static_cast<const VelSet *>(c->cur_velset);      // transform set of the velocity component being solved (nullptr: the pressure's)
  if (dist && !c->comm.on) { c->err = "solver: nranks > 1 but no communication hooks registered (cales_set_comm)"; return 1; }
  const bool nyq = poisson && !VS && c->nyq_ok && c->xkind == 0 && c->ykind <= 1;
  const int cw = nyq ? c->cw_nyq : c->cw, nmodes = nyq ? n[0] / 2 : mh;      // complex mode columns: per rank, in all
  Spec S; S.blocked = dist ? 1 : 0; S.cw = cw; S.n2l = n[1]; S.n3 = n[2]; S.nyq = nyq ? 1 : 0;
  real2 *slab_spec = dist ? reinterpret_cast<real2 *>(c->comm.A) : reinterpret_cast<real2 *>(pp + 1);   // in place: modes of row (j,k) from i = 1
  real2 *mode_spec = dist ? reinterpret_cast<real2 *>(c->comm.B) : reinterpret_cast<real2 *>(pp + 1);
  const int ncol = dist ? cw : nmodes, mofs = dist ? c->rank * cw : 0;
  const int64_t a2a_count = (int64_t)n[2] * n[1] * cw * 2;
  const int nh = c->C.ng[0] / 2;
  const bool use16y = use8y && sp->y16;
  const int Rx8 = use8x ? sp->x8_threads / (nh / 8) : 1, CB8 = use16y ? 8 : use8y ? sp->y8_threads / (n2g / 8) : 1;
  // real x modes (Neumann in x) fill the complex columns 0 .. n1/2 - 1: the slot of "mode n1/2" is never written by the x pass nor read back by it
  const int ncol_y = (!dist && c->xkind == 1 && use8y) ? nh : ncol;
  auto launch_y16 = [&](int inv, dim3 gy, int kc, int ka, int kb) {
#define Y16_GO(NN, IV, KD) LAUNCH(c, (k_fft_y16<NN, IV, KD>), gy, dim3(NN / 2), sp->shy16, c->stream, c->g, ncol_y, kc, (const cpx *)c->d_twy, (const cpx *)c->scr_twyd, S, mode_spec, ka, kb)
#define Y16_N(NN) do { if (inv) { if (c->ykind) Y16_GO(NN, 1, 1); else Y16_GO(NN, 1, 0); } else { if (c->ykind) Y16_GO(NN, 0, 1); else Y16_GO(NN, 0, 0); } } while (0)
    if (n2g == 1024) Y16_N(1024); else if (n2g == 512) Y16_N(512); else Y16_N(256);
#undef Y16_N
#undef Y16_GO
  };
  // persistent blocks: several row groups / planes per block so that the register prefetch overlaps the transforms
  const long xgroups = (nrows + Rx8 - 1) / Rx8;
  int xiters = 1; while (xiters < 8 && xgroups / (xiters * 2) >= 2048) xiters *= 2;
  // measured (round 6, one box, row groups per block 1 / 2 / 4 / 8): forward pass with fillps 512 x 256 x 256 0.672 / 0.690 / 0.723 / 0.748 ms per step but
  // 512^3 3.15 / 2.90 / 2.84 / 2.77 and 1024^3 25.3 / 23.2 / 22.9 / 21.2 -- persistent blocks pay from ~16 000 row groups; inverse periodic pass 512^3
  // 1.36 / 1.41 / 1.44 / 1.50 (one group per block), inverse Neumann pass 1024^3 12.6 / 12.0 / 11.4 / 10.8 (eight)
  if (xgroups <= 8192) xiters = 1;
  const unsigned xblocks = (unsigned)((xgroups + xiters - 1) / xiters);
  const int xiters_b = (c->xkind == 0 && nh <= 256) ? 1 : xiters;
  const unsigned xblocks_b = (unsigned)((xgroups + xiters_b - 1) / xiters_b);
  // planes per block of the eight-elements-per-thread y kernels: measured at 512^3 (round 6, in the step, planes per block 1 / 2 / 8): y passes 2.77 / 2.69 / 2.94 ms
  // per step -- many short blocks beat long persistent ones; from 8192 blocks on
  int ykchunk = 1; { const long cg = (ncol + CB8 - 1) / CB8; while (ykchunk < 8 && cg * (n[2] / (ykchunk * 2)) >= 8192 && n[2] % (ykchunk * 2) == 0) ykchunk *= 2; }
  // k_fft_y16 holds 1 / 2 / 4 blocks per CU (1024 / 512 / 256 points): a launch runs in ceil(blocks / slots) rounds of (planes per block + ~1.5 planes of
  // set-up: tables, the first plane's latency) each -- the chunk length that minimises that product (512^3: 7 planes, 5 rounds of 8.5 against 5 of 9.5
  // with 8; 512 x 256 x 256: 3 planes; 1024^3: 32)
  auto y16_chunk = [&](int planes) {
    const long slots = (long)(c->ncu > 0 ? c->ncu : 256) * (n2g >= 1024 ? 1 : n2g >= 512 ? 2 : 4), tiles = (ncol_y + 7) / 8;
    double best = 1e300; int bk = 1;
    for (int kc = 1; kc <= 32 && kc <= planes; ++kc) {
      const long nb = tiles * ((planes + kc - 1) / kc);
      const double cost = (double)((nb + slots - 1) / slots) * (kc + 1.5);
      if (cost < best) { best = cost; bk = kc; }
    }
    return bk;
  };
  if (use16y) ykchunk = y16_chunk(n[2]);
  const int ychunks = (n[2] + ykchunk - 1) / ykchunk;
  // ---- pipelined exchange (cales_set_comm_overlap): the spectrum travels in k-chunks on the second stream while the x transforms of
  // the next chunk and the y transforms of the previous one run on the context's stream (the reference's cuDecomp pipelined
  // transposes, src/initmpi.f90:94-139). Radix-8 periodic / Neumann-Neumann transforms only; everything else keeps the single exchange.
  int NCH = 1;
  if (dist && c->comm.a2a_part && c->comm_stream && use8x && use8y) { if (n[2] % 4 == 0 && n[2] >= 32) NCH = 4; else if (n[2] % 2 == 0 && n[2] >= 8) NCH = 2; }
  const bool pipe = NCH > 1;
  const int kpc = n[2] / NCH;                                   // planes per chunk
  const int64_t a2a_stride = a2a_count, a2a_chunk = (int64_t)kpc * n[1] * cw * 2;
  const long rows_c = (long)n[1] * kpc, xgroups_c = (rows_c + Rx8 - 1) / Rx8;
  int xiters_c = 1; while (xiters_c < 8 && xgroups_c / (xiters_c * 2) >= 2048 / NCH) xiters_c *= 2;
  const unsigned xblocks_c = (unsigned)((xgroups_c + xiters_c - 1) / xiters_c);
  int ykchunk_c = 1; { const long cg = (ncol + CB8 - 1) / CB8; while (ykchunk_c < 8 && cg * (kpc / (ykchunk_c * 2)) >= 2048 / NCH && kpc % (ykchunk_c * 2) == 0) ykchunk_c *= 2; }
  if (use16y) ykchunk_c = y16_chunk(kpc);
  hipEvent_t ev_arrived[4];
  // the bulk means summed by the fused fillps pass: several ranks all-reduce them on the context's stream, and RCCL orders the operations of
  // one communicator across streams -- issued between the chunk exchanges it would hold the y transforms back, so it follows the z sweep
  int pend_mean_mask = 0, pend_mean_nblk = 0; const real *pend_mean_part = nullptr;
  auto mark = [&](hipStream_t st, hipEvent_t &e) -> int {
    if (c->sync_ev.size() < 64) { hipEvent_t ne; HIPCHK(c, hipEventCreateWithFlags(&ne, hipEventDisableTiming)); c->sync_ev.push_back(ne); c->sync_next = c->sync_ev.size() - 1; }
    e = c->sync_ev[c->sync_next]; c->sync_next = (c->sync_next + 1) % c->sync_ev.size();
    HIPCHK(c, hipEventRecord(e, st)); return 0; };
  auto exchange_chunk = [&](int dir, int ch) -> int {          // comm stream: after everything queued on the context's stream so far
    if (int e = stream_after(c, c->comm_stream, c->stream)) return e;
    { ProfScope ps(c, "alltoall", c->comm_stream);
      if (c->comm.a2a_part(c->comm.user, dir, a2a_stride, (int64_t)ch * a2a_chunk, a2a_chunk, (void *)c->comm_stream)) { c->err = "alltoall_part callback failed"; return 1; } }
    return mark(c->comm_stream, ev_arrived[ch]); };
  if (pipe) {
    const bool fill = poisson && c->fuse_fillps_dti != 0. && sp->x8;
    FillArgs F{};
    if (fill) {
      const real dti = c->fuse_fillps_dti;
      F = FillArgs{c->f[CALES_U], c->f[CALES_V], c->f[CALES_W], c->d_dzfi, dti, dti * c->dli[0], dti * c->dli[1], c->fuse_mean_mask, c->d_gvr_f, c->d_gvr_c, nullptr};
      F.xwrap = c->step_xskip ? n[0] : 0;
      if (F.mean_mask) {
        const size_t need = 3 * (size_t)xblocks_c * NCH;
        if (c->n_mpart < need) { if (c->d_mpart) hipFree(c->d_mpart); HIPCHK(c, hipMalloc(&c->d_mpart, need * sizeof(real))); c->n_mpart = need; }
        F.part = c->d_mpart; F.pstride = (int)(xblocks_c * NCH);
      }
    }
    for (int ch = 0; ch < NCH; ++ch) {
      const long rb = (long)n[1] * kpc * ch, re = rb + rows_c;
      { ProfScope ps(c, fill ? "fillps_fft_x_fwd" : "fft_x_fwd");
        F.pofs = (int)(xblocks_c * ch);
        if (fill && c->xkind) LAUNCH(c, (k_fft_x8<0, 1, 1>), dim3(xblocks_c), dim3(sp->x8_threads), sp->shx8, c->stream, c->g, nh, xiters_c, (const cpx *)c->d_twx, (const cpx *)c->d_twx_post, (const cpx *)c->d_twy_post, pp, 1., S, slab_spec, F, rb, re);
        else if (fill) LAUNCH(c, (k_fft_x8<0, 0, 1>), dim3(xblocks_c), dim3(sp->x8_threads), sp->shx8, c->stream, c->g, nh, xiters_c, (const cpx *)c->d_twx, (const cpx *)c->d_twx_post, (const cpx *)c->d_twy_post, pp, 1., S, slab_spec, F, rb, re);
        else if (c->xkind) LAUNCH(c, (k_fft_x8<0, 1>), dim3(xblocks_c), dim3(sp->x8_threads), sp->shx8, c->stream, c->g, nh, xiters_c, (const cpx *)c->d_twx, (const cpx *)c->d_twx_post, (const cpx *)c->d_twy_post, pp, 1., S, slab_spec, FillArgs{}, rb, re);
        else LAUNCH(c, (k_fft_x8<0, 0>), dim3(xblocks_c), dim3(sp->x8_threads), sp->shx8, c->stream, c->g, nh, xiters_c, (const cpx *)c->d_twx, (const cpx *)c->d_twx_post, (const cpx *)c->d_twy_post, pp, 1., S, slab_spec, FillArgs{}, rb, re); }
      if (int e = exchange_chunk(0, ch)) return e;
    }
    if (fill && F.mean_mask) { pend_mean_mask = F.mean_mask; pend_mean_part = F.part; pend_mean_nblk = (int)(xblocks_c * NCH); }
    for (int ch = 0; ch < NCH; ++ch) {
      HIPCHK(c, hipStreamWaitEvent(c->stream, ev_arrived[ch], 0));
      ProfScope ps(c, "fft_y_fwd");
      const dim3 gy((ncol_y + CB8 - 1) / CB8, (kpc + ykchunk_c - 1) / ykchunk_c);
      if (use16y) launch_y16(0, gy, ykchunk_c, kpc * ch, kpc * (ch + 1));
      else if (c->ykind) LAUNCH(c, (k_fft_y8<0, 1>), gy, dim3(sp->y8_threads), sp->shy8, c->stream, c->g, n2g, ncol_y, ykchunk_c, (const cpx *)c->d_twy, (const cpx *)c->scr_twyd, S, mode_spec, kpc * ch, kpc * (ch + 1));
      else LAUNCH(c, (k_fft_y8r<0>), gy, dim3(sp->y8_threads), sp->shy8r, c->stream, c->g, n2g, ncol, ykchunk_c, (const cpx *)c->d_twy, S, mode_spec, kpc * ch, kpc * (ch + 1));
    }
  } else {
  if (poisson && c->fuse_fillps_dti != 0. && sp->x8) {
    ProfScope ps(c, "fillps_fft_x_fwd");
    const real dti = c->fuse_fillps_dti;
    FillArgs F{c->f[CALES_U], c->f[CALES_V], c->f[CALES_W], c->d_dzfi, dti, dti * c->dli[0], dti * c->dli[1], c->fuse_mean_mask, c->d_gvr_f, c->d_gvr_c, nullptr};
    F.xwrap = c->step_xskip ? n[0] : 0;
    if (F.mean_mask) {
      if (c->n_mpart < 3 * (size_t)xblocks) { if (c->d_mpart) hipFree(c->d_mpart); HIPCHK(c, hipMalloc(&c->d_mpart, 3 * (size_t)xblocks * sizeof(real))); c->n_mpart = 3 * (size_t)xblocks; }
      F.part = c->d_mpart;
    }
    if (c->xkind) LAUNCH(c, (k_fft_x8<0, 1, 1>), dim3(xblocks), dim3(sp->x8_threads), sp->shx8, c->stream, c->g, nh, xiters,
                                     (const cpx *)c->d_twx, (const cpx *)c->d_twx_post, (const cpx *)c->d_twy_post, pp, 1., S, slab_spec, F);
    else LAUNCH(c, (k_fft_x8<0, 0, 1>), dim3(xblocks), dim3(sp->x8_threads), sp->shx8, c->stream, c->g, nh, xiters,
                            (const cpx *)c->d_twx, (const cpx *)c->d_twx_post, (const cpx *)c->d_twy_post, pp, 1., S, slab_spec, F);
    if (F.mean_mask) if (int e = op_force_from_partials(c, F.mean_mask, F.part, (int)xblocks)) return e;
  } else { ProfScope ps(c, "fft_x_fwd");
    if (use8x && c->xkind) LAUNCH(c, (k_fft_x8<0, 1>), dim3(xblocks), dim3(sp->x8_threads), sp->shx8, c->stream, c->g, nh, xiters,
                                   (const cpx *)c->d_twx, (const cpx *)c->d_twx_post, (const cpx *)c->d_twy_post, pp, 1., S, slab_spec);
    else if (use8x) LAUNCH(c, (k_fft_x8<0, 0>), dim3(xblocks), dim3(sp->x8_threads), sp->shx8, c->stream, c->g, nh, xiters,
                                   (const cpx *)c->d_twx, (const cpx *)c->d_twx_post, (const cpx *)c->d_twy_post, pp, 1., S, slab_spec);
    else if (c->xkind >= 5) LAUNCH(c, k_dst1<0>, dim3((unsigned)nrows), dim3(256), (size_t)2 * (VS->p1x.N + 1) * sizeof(cpx), c->stream, c->g, VS->p1x, 0, (const cpx *)VS->tw1x, pp, 1., S, slab_spec, c->xkind, 0);
    else if (c->xkind == 3) LAUNCH(c, (k_fft_x4<0, 0>), dim3((unsigned)((nrows + sp->Rx - 1) / sp->Rx)), dim3(256), sp->shx, c->stream, c->g, sp->px, sp->Rx,
                       (const cpx *)c->d_twx, (const cpx *)c->d_tw4x, pp, 1., S, slab_spec);
    else if (c->xkind == 4) LAUNCH(c, (k_fft_x4<0, 1>), dim3((unsigned)((nrows + sp->Rx - 1) / sp->Rx)), dim3(256), sp->shx, c->stream, c->g, sp->px, sp->Rx,
                       (const cpx *)c->d_twx, (const cpx *)c->d_tw4x, pp, 1., S, slab_spec);
    else LAUNCH(c, k_fft_x<0>, dim3((unsigned)((nrows + sp->Rx - 1) / sp->Rx)), dim3(256), sp->shx, c->stream, c->g, sp->px, sp->Rx, c->xkind,
                       (const cpx *)c->d_twx, (const cpx *)c->d_twx_post, (const cpx *)c->d_twy_post, pp, 1., S, slab_spec); }
  if (dist) { ProfScope ps(c, "alltoall"); if (c->comm.a2a(c->comm.user, 0, a2a_count)) { c->err = "alltoall callback failed"; return 1; } }
  { ProfScope ps(c, "fft_y_fwd");
    if (c->ykind >= 5) LAUNCH(c, k_dst1<1>, dim3(ncol, n[2]), dim3(256), (size_t)2 * (VS->p1y.N + 1) * sizeof(cpx), c->stream, c->g, VS->p1y, ncol, (const cpx *)VS->tw1y, pp, 1., S, mode_spec, c->ykind, 0);
    else if (c->ykind == 3) LAUNCH(c, k_fft_y4<0>, dim3((ncol + sp->CBy4 - 1) / sp->CBy4, n[2]), dim3(256), sp->shy4, c->stream, c->g, sp->py4, sp->CBy4, ncol, (const cpx *)c->d_twy4, (const cpx *)c->d_tw4y, S, mode_spec);
    else if (c->ykind == 4) LAUNCH(c, k_fft_y4<1>, dim3((ncol + sp->CBy4 - 1) / sp->CBy4, n[2]), dim3(256), sp->shy4, c->stream, c->g, sp->py4, sp->CBy4, ncol, (const cpx *)c->d_twy4, (const cpx *)c->d_tw4y, S, mode_spec);
    else if (use16y) launch_y16(0, dim3((ncol_y + 7) / 8, ychunks), ykchunk, 0, -1);
    else if (use8y && c->ykind) LAUNCH(c, (k_fft_y8<0, 1>), dim3((ncol_y + CB8 - 1) / CB8, ychunks), dim3(sp->y8_threads), sp->shy8, c->stream, c->g, n2g, ncol_y, ykchunk, (const cpx *)c->d_twy, (const cpx *)c->scr_twyd, S, mode_spec);
    else if (use8y) LAUNCH(c, (k_fft_y8r<0>), dim3((ncol + CB8 - 1) / CB8, ychunks), dim3(sp->y8_threads), sp->shy8r, c->stream, c->g, n2g, ncol, ykchunk, (const cpx *)c->d_twy, S, mode_spec);
    else LAUNCH(c, k_fft_y<0>, dim3((ncol + sp->CBy - 1) / sp->CBy, n[2]), dim3(256), sp->shy, c->stream, c->g, sp->py, sp->CBy, ncol, c->ykind, (const cpx *)c->d_twy, (const cpx *)c->scr_twyd, S, mode_spec); }
  }
  // pressure equation without a Dirichlet condition in z: the zero-eigenvalue mode (if x and y have one) is singular
  const int fixnull = (poisson && CBP(c, 0, 3) != 'D' && CBP(c, 1, 3) != 'D' && !c->fl.keep_null_mode) ? 1 : 0;
  // CALES_KEEP_NULL_MODE: the singular column in the reference's sequential order (k_null_column); the rank that holds mode 0 does it
  const bool refnull = poisson && lscale == (real)1. && CBP(c, 0, 3) != 'D' && CBP(c, 1, 3) != 'D' && c->fl.keep_null_mode && mofs == 0;
  if (refnull) {
    if (!c->d_nullw) HIPCHK(c, hipMalloc(&c->d_nullw, (size_t)5 * (c->C.ng[2] + 2) * sizeof(real)));
    LAUNCH(c, k_null_column, dim3(1), dim3(64), 0, c->stream, c->g, S, nz, periodic_z ? 1 : 0, da, db, dc, c->d_lamx, c->d_lamy, mode_spec, c->d_nullw, 0);
  }
  { ProfScope ps(c, "gaussel_z");
    dim3 b(64, 4), gr((ncol + 63) / 64, (n2g + 3) / 4);
    if (c->xkind && !c->ykind)       // real x modes paired into complex columns + periodic y: Hermitian separation of rows ky and N-ky
      LAUNCH(c, k_gaussel_herm, dim3((unsigned)(((long)4 * ncol * (n2g / 2 + 1) + 255) / 256)), dim3(256), 0, c->stream, c->g, nz, ncol, n2g, mofs, c->C.ng[0] / 2, S, lscale,
                         da, db, dc, c->d_lamx, c->d_lamy, (real *)mode_spec, c->scr1, fixnull);
    else if ((periodic_z || fixnull || !poisson || CBP(c, 0, 3) == 'D' || CBP(c, 1, 3) == 'D') &&
             [&]() {
               TileMap T{}; T.blocked = dist ? 1 : 0; T.cw = cw; T.n2l = n[1]; T.mofs = mofs; T.nmode = c->xkind ? c->C.ng[0] / 2 : nmodes; T.nyq = nyq ? 1 : 0;
               T.kstride = (size_t)2 * cw * n[1]; T.segstride = T.kstride * n[2];
               const int ndbl = dist ? 2 * cw * n[1] : (c->xkind ? 2 * (c->C.ng[0] / 2) : 2 * nmodes), nseg = dist ? c->P : n2g;
               if (periodic_z)      // the cyclic closure inside the tile (gaussel_periodic, solver.f90:109-150)
                 return c->xkind ? gaussel_tile<1, 1>(c, nz, ndbl, nseg, lscale, da, db, dc, (real *)mode_spec, fixnull, T)
                                 : gaussel_tile<2, 1>(c, nz, ndbl, nseg, lscale, da, db, dc, (real *)mode_spec, fixnull, T);
               return c->xkind ? gaussel_tile<1>(c, nz, ndbl, nseg, lscale, da, db, dc, (real *)mode_spec, fixnull, T)
                               : gaussel_tile<2>(c, nz, ndbl, nseg, lscale, da, db, dc, (real *)mode_spec, fixnull, T); }()) {}
    else if (c->xkind && !periodic_z)
      LAUNCH(c, k_gaussel_ri, dim3((unsigned)(((long)2 * ncol * n2g + 255) / 256)), dim3(256), 0, c->stream, c->g, nz, ncol, n2g, mofs, c->C.ng[0] / 2, S, lscale, da, db, dc,
                         c->d_lamx, c->d_lamy, (real *)mode_spec, c->scr1, fixnull, 1);
    else if (periodic_z && c->xkind)      // real x modes (one eigenvalue each) with the periodic-z closure: scalar columns of the in-place spectrum, one rank
      LAUNCH(c, (k_gaussel<real, 1>), dim3((2 * (c->C.ng[0] / 2) + 63) / 64, (n2g + 3) / 4), b, 0, c->stream, c->g, nz, 2 * (c->C.ng[0] / 2), n2g, 1, 0, 2 * (c->C.ng[0] / 2), S, lscale,
                         da, db, dc, c->d_lamx, c->d_lamy, pp, c->scr1, c->scr2, fixnull);
    else if (c->xkind) LAUNCH(c, k_gaussel_split, gr, b, 0, c->stream, c->g, nz, ncol, n2g, mofs, c->C.ng[0] / 2, S, lscale, da, db, dc, c->d_lamx, c->d_lamy,
                                     (real2 *)mode_spec, (real2 *)c->scr1, fixnull);
    else if (periodic_z) LAUNCH(c, (k_gaussel<real2, 1>), gr, b, 0, c->stream, c->g, nz, ncol, n2g, 0, mofs, mh, S, lscale, da, db, dc, c->d_lamx, c->d_lamy, (real *)mode_spec, c->scr1, c->scr2, fixnull);
    else LAUNCH(c, k_gaussel_ri, dim3((unsigned)(((long)2 * ncol * n2g + 255) / 256)), dim3(256), 0, c->stream, c->g, nz, ncol, n2g, mofs, mh, S, lscale, da, db, dc, c->d_lamx, c->d_lamy, (real *)mode_spec, c->scr1, fixnull, 0); }
  if (nyq && mofs == 0) {      // the packed column of the modes 0 and n1/2 (the tile left it out); same coefficient table as the tile of this nz
    ProfScope ps(c, "gaussel_z");
    if (nz <= 128) launch_gaussel_nyq<2>(c, nz, n2g, nh, lscale, da, db, dc, mode_spec, fixnull, S, c->d_abct);
    else if (nz <= 256) launch_gaussel_nyq<4>(c, nz, n2g, nh, lscale, da, db, dc, mode_spec, fixnull, S, c->d_abct);
    else if (nz <= 512) launch_gaussel_nyq<8>(c, nz, n2g, nh, lscale, da, db, dc, mode_spec, fixnull, S, c->d_abct);
    else launch_gaussel_nyq<16>(c, nz, n2g, nh, lscale, da, db, dc, mode_spec, fixnull, S, c->d_abct);
  }
  if (refnull) LAUNCH(c, k_null_column, dim3(1), dim3(64), 0, c->stream, c->g, S, nz, periodic_z ? 1 : 0, da, db, dc, c->d_lamx, c->d_lamy, mode_spec, c->d_nullw, 1);
  if (pend_mean_mask) if (int e = op_force_from_partials(c, pend_mean_mask, pend_mean_part, pend_mean_nblk)) return e;
  if (pipe) {
    for (int ch = 0; ch < NCH; ++ch) {
      { ProfScope ps(c, "fft_y_bwd");
        const dim3 gy((ncol_y + CB8 - 1) / CB8, (kpc + ykchunk_c - 1) / ykchunk_c);
        if (use16y) launch_y16(1, gy, ykchunk_c, kpc * ch, kpc * (ch + 1));
        else if (c->ykind) LAUNCH(c, (k_fft_y8<1, 1>), gy, dim3(sp->y8_threads), sp->shy8, c->stream, c->g, n2g, ncol_y, ykchunk_c, (const cpx *)c->d_twy, (const cpx *)c->scr_twyd, S, mode_spec, kpc * ch, kpc * (ch + 1));
        else LAUNCH(c, (k_fft_y8r<1>), gy, dim3(sp->y8_threads), sp->shy8r, c->stream, c->g, n2g, ncol, ykchunk_c, (const cpx *)c->d_twy, S, mode_spec, kpc * ch, kpc * (ch + 1)); }
      if (int e = exchange_chunk(1, ch)) return e;
    }
    for (int ch = 0; ch < NCH; ++ch) {
      HIPCHK(c, hipStreamWaitEvent(c->stream, ev_arrived[ch], 0));
      ProfScope ps(c, "fft_x_bwd");
      const long rb = (long)n[1] * kpc * ch, re = rb + rows_c;
      if (c->xkind) LAUNCH(c, (k_fft_x8<1, 1>), dim3(xblocks_c), dim3(sp->x8_threads), sp->shx8, c->stream, c->g, nh, xiters_c, (const cpx *)c->d_twx, (const cpx *)c->d_twx_post, (const cpx *)c->d_twy_post, pp, c->normfft, S, slab_spec, FillArgs{}, rb, re);
      else LAUNCH(c, (k_fft_x8<1, 0>), dim3(xblocks_c), dim3(sp->x8_threads), sp->shx8, c->stream, c->g, nh, xiters_c, (const cpx *)c->d_twx, (const cpx *)c->d_twx_post, (const cpx *)c->d_twy_post, pp, c->normfft, S, slab_spec, FillArgs{}, rb, re);
    }
  } else {
  { ProfScope ps(c, "fft_y_bwd");
    if (c->ykind >= 5) LAUNCH(c, k_dst1<1>, dim3(ncol, n[2]), dim3(256), (size_t)2 * (VS->p1y.N + 1) * sizeof(cpx), c->stream, c->g, VS->p1y, ncol, (const cpx *)VS->tw1y, pp, 1., S, mode_spec, c->ykind, 1);
    else if (c->ykind == 3) LAUNCH(c, k_fft_y4<0>, dim3((ncol + sp->CBy4 - 1) / sp->CBy4, n[2]), dim3(256), sp->shy4, c->stream, c->g, sp->py4, sp->CBy4, ncol, (const cpx *)c->d_twy4, (const cpx *)c->d_tw4y, S, mode_spec);
    else if (c->ykind == 4) LAUNCH(c, k_fft_y4<1>, dim3((ncol + sp->CBy4 - 1) / sp->CBy4, n[2]), dim3(256), sp->shy4, c->stream, c->g, sp->py4, sp->CBy4, ncol, (const cpx *)c->d_twy4, (const cpx *)c->d_tw4y, S, mode_spec);
    else if (use16y) launch_y16(1, dim3((ncol_y + 7) / 8, ychunks), ykchunk, 0, -1);
    else if (use8y && c->ykind) LAUNCH(c, (k_fft_y8<1, 1>), dim3((ncol_y + CB8 - 1) / CB8, ychunks), dim3(sp->y8_threads), sp->shy8, c->stream, c->g, n2g, ncol_y, ykchunk, (const cpx *)c->d_twy, (const cpx *)c->scr_twyd, S, mode_spec);
    else if (use8y) LAUNCH(c, (k_fft_y8r<1>), dim3((ncol + CB8 - 1) / CB8, ychunks), dim3(sp->y8_threads), sp->shy8r, c->stream, c->g, n2g, ncol, ykchunk, (const cpx *)c->d_twy, S, mode_spec);
    else LAUNCH(c, k_fft_y<1>, dim3((ncol + sp->CBy - 1) / sp->CBy, n[2]), dim3(256), sp->shy, c->stream, c->g, sp->py, sp->CBy, ncol, c->ykind, (const cpx *)c->d_twy, (const cpx *)c->scr_twyd, S, mode_spec); }
  if (dist) { ProfScope ps(c, "alltoall"); if (c->comm.a2a(c->comm.user, 1, a2a_count)) { c->err = "alltoall callback failed"; return 1; } }
  { ProfScope ps(c, "fft_x_bwd");
    if (use8x && c->xkind) LAUNCH(c, (k_fft_x8<1, 1>), dim3(xblocks_b), dim3(sp->x8_threads), sp->shx8, c->stream, c->g, nh, xiters_b,
                                   (const cpx *)c->d_twx, (const cpx *)c->d_twx_post, (const cpx *)c->d_twy_post, pp, c->normfft, S, slab_spec);
    else if (use8x) LAUNCH(c, (k_fft_x8<1, 0>), dim3(xblocks_b), dim3(sp->x8_threads), sp->shx8, c->stream, c->g, nh, xiters_b,
                                   (const cpx *)c->d_twx, (const cpx *)c->d_twx_post, (const cpx *)c->d_twy_post, pp, c->normfft, S, slab_spec);
    else if (c->xkind >= 5) LAUNCH(c, k_dst1<2>, dim3((unsigned)nrows), dim3(256), (size_t)2 * (VS->p1x.N + 1) * sizeof(cpx), c->stream, c->g, VS->p1x, 0, (const cpx *)VS->tw1x, pp, c->normfft, S, slab_spec, c->xkind, 1);
    else if (c->xkind == 3) LAUNCH(c, (k_fft_x4<1, 0>), dim3((unsigned)((nrows + sp->Rx - 1) / sp->Rx)), dim3(256), sp->shx, c->stream, c->g, sp->px, sp->Rx,
                       (const cpx *)c->d_twx, (const cpx *)c->d_tw4x, pp, c->normfft, S, slab_spec);
    else if (c->xkind == 4) LAUNCH(c, (k_fft_x4<1, 1>), dim3((unsigned)((nrows + sp->Rx - 1) / sp->Rx)), dim3(256), sp->shx, c->stream, c->g, sp->px, sp->Rx,
                       (const cpx *)c->d_twx, (const cpx *)c->d_tw4x, pp, c->normfft, S, slab_spec);
    else LAUNCH(c, k_fft_x<1>, dim3((unsigned)((nrows + sp->Rx - 1) / sp->Rx)), dim3(256), sp->shx, c->stream, c->g, sp->px, sp->Rx, c->xkind,
                       (const cpx *)c->d_twx, (const cpx *)c->d_twx_post, (const cpx *)c->d_twy_post, pp, c->normfft, S, slab_spec); }
  }
  LAUNCHCHK(c);
  return 0;
}

// the kernels the pressure solve of this context takes (cales_describe_plan): the selections of solve_field above, by name
std::string solver_path_name(cales_ctx *c) {
  SolverPlans *sp = find_plans(c);
  if (!sp) return "unset";
  static const char *kinds[] = {"PP", "NN", "DD", "ND", "DN"};
  const bool use8x = sp->x8 && c->xkind <= 1, use8y = sp->y8 && c->ykind <= 1;
  const bool perz = CBP(c, 0, 3) == 'P' && CBP(c, 1, 3) == 'P';
  const int nz = c->n[2];
  const bool hasd = CBP(c, 0, 3) == 'D' || CBP(c, 1, 3) == 'D';
  const bool tile = !(c->xkind && !c->ykind) && (perz || hasd || !c->fl.keep_null_mode) && !(nz < (perz ? 4 : 2) || nz > 1024 || c->fl.gaussel_march);
  std::string s = std::string("x:") + kinds[c->xkind] + (use8x ? "/radix8" : c->xkind >= 3 ? "/dct4" : "/mixed_radix");
  s += std::string(",y:") + kinds[c->ykind] + (use8y ? (c->ykind ? "/radix8" : "/radix8_register_ends") : c->ykind >= 3 ? "/dct4" : "/mixed_radix");
  s += std::string(",z:") + (c->xkind && !c->ykind ? "thomas_hermitian" : tile ? (perz ? "lds_tile_periodic" : "lds_tile") : "thomas_march");
  if (c->nyq_ok) s += ",modes_0_and_n1/2:one_column";
  if (c->fl.keep_null_mode) s += ",null_mode:reference_order";
  return s;
}

int op_solver(cales_ctx *c) {
  return solve_field(c, c->f[CALES_PP], c->d_a, c->d_b, c->d_c, c->n[2], 1., CBP(c, 0, 3) == 'P' && CBP(c, 1, 3) == 'P', true);
}

// z-only Helmholtz systems have the same matrix for every column (no eigenvalue shift): the pivots z_l and c'_l of the
// Thomas recurrence (solver.f90:160-178, same operations in the same order) are computed once by one thread, and the
// sweeps of the field read them instead of dividing and of storing a c' per cell.
__global__ void k_thomas_coef(int n, const real *__restrict__ a, const real *__restrict__ b, const real *__restrict__ c,
                              real *__restrict__ zz, real *__restrict__ dd) {
  if (threadIdx.x || blockIdx.x) return;
  real z = 1. / (b[0] + CALES_EPS), d = c[0] * z;
  zz[0] = z; dd[0] = d;
  for (int l = 1; l < n; ++l) { z = 1. / (b[l] - a[l] * d + CALES_EPS); d = c[l] * z; zz[l] = z; dd[l] = d; }
}
__global__ __launch_bounds__(256) void k_gaussel_cols(Geom g, int nz, const real *__restrict__ a, const real *__restrict__ zz,
                                                      const real *__restrict__ dd, real *__restrict__ p) {
  const int i = blockIdx.x * 64 + threadIdx.x + 1, j = blockIdx.y * 4 + threadIdx.y + 1;
  if (i > g.n1 || j > g.n2) return;
  const size_t e0 = g.ix(i, j, 1), st = (size_t)g.s12;
  real v = p[e0] * zz[0];
  p[e0] = v;
  for (int l = 1; l < nz; ++l) { v = (p[e0 + l * st] - a[l] * v) * zz[l]; p[e0 + l * st] = v; }
  for (int l = nz - 2; l >= 0; --l) { v = p[e0 + l * st] - dd[l] * v; p[e0 + l * st] = v; }
}

// The same sweeps for the z-implicit time step (cales_step): the r.h.s. of rk.f90:108-118 and main.f90:422-433 is formed on
// the fly in the reference's order -- (u - hf12*dudtd) + f + rhs_b -- so the separate passes over u (9 + 2 words) disappear.
// nq = nz or nz+1: plane nz+1 (the wall face of w) is not an unknown but still receives the first two terms.
__global__ __launch_bounds__(256) void k_gaussel_cols_rhs(Geom g, int nz, int nq, const real *__restrict__ a, const real *__restrict__ zz,
                                                          const real *__restrict__ dd, real *__restrict__ p, const real *__restrict__ dud,
                                                          real hf12, const real *__restrict__ force, const real *__restrict__ rb, int has_lo,
                                                          int has_hi) {
  const int i = blockIdx.x * 64 + threadIdx.x + 1, j = blockIdx.y * 4 + threadIdx.y + 1;
  if (i > g.n1 || j > g.n2) return;
  const size_t e0 = g.ix(i, j, 1), st = (size_t)g.s12, pq = (size_t)(i - 1) + (size_t)g.n1 * (j - 1);
  const bool forced = force != nullptr; const real f = forced ? force[0] : 0.;
  auto rhs = [&](int l) {
    real t = p[e0 + l * st] - hf12 * dud[e0 + l * st];
    if (forced) t = t + f;
    if (l == 0 && has_lo) t = t + rb[pq];
    if (l == nz - 1 && has_hi) t = t + rb[pq + (size_t)g.n1 * g.n2];
    return t;
  };
  real v = rhs(0) * zz[0];
  p[e0] = v;
  for (int l = 1; l < nz; ++l) { v = (rhs(l) - a[l] * v) * zz[l]; p[e0 + l * st] = v; }
  if (nq > nz) p[e0 + (size_t)nz * st] = rhs(nz);
  for (int l = nz - 2; l >= 0; --l) { v = p[e0 + l * st] - dd[l] * v; p[e0 + l * st] = v; }
}

// z-implicit Helmholtz solve of one velocity component (solver.f90:182-233 with aa,bb,cc of main.f90:435-437)
__global__ void k_scale_abc(int n, real alpha, const real *a, const real *b, const real *c, real *aa, real *bb, real *cc) {
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k < n) { aa[k] = a[k] * alpha; bb[k] = b[k] * alpha + 1.; cc[k] = c[k] * alpha; }
}
int op_rhs_b_velz(cales_ctx *c, int ivel, real alpha, real *planes = nullptr, int *has = nullptr);
void rhs_b_velz_args(cales_ctx *c, int ivel, real alpha, RhsBz *R, int *has);
int op_rhs_b_velxy(cales_ctx *c, int ivel, real alpha);
int op_helmholtz_z(cales_ctx *c, int ivel, real alpha) {
  if (c->C.impdiff != 2) { c->err = "helmholtz_z needs impdiff = 2"; return 1; }
  ProfScope ps(c, "helmholtz_z");
  const int *n = c->n; const int n3 = n[2];
  const bool fused = c->defer_imp_rhs;
  int has[2] = {0, 0};
  const char *bcz0 = &c->cbcvel[6 * (ivel - 1) + 4];
  const int q0 = (ivel == 3 && bcz0[1] == 'D') ? 1 : 0;
  // the in-LDS sweep evaluates the boundary term itself (RhsBz): no pass, no plane
  const bool tile_path = !(bcz0[0] == 'P' && bcz0[1] == 'P') && !c->fl.helmholtz_z_per_column && n3 - q0 >= 2 && n3 - q0 <= 512 &&
                         n3 <= 64 * (n3 - q0 <= 128 ? 2 : n3 - q0 <= 256 ? 4 : 8) && !c->fl.gaussel_march;
  RhsBz RB[2] = {};
  if (fused && tile_path) rhs_b_velz_args(c, ivel, alpha, RB, has);
  else if (int e = op_rhs_b_velz(c, ivel, alpha, fused ? c->scr2 : nullptr, fused ? has : nullptr)) return e;
  real *abc = c->d_red + 64 + 16 * (n3 + 2);     // scaled coefficients live behind the reduction partials
  // the table of this (component, alpha) from an earlier sweep (in-LDS form only: the others read the scaled coefficients themselves)
  real *hz = nullptr; bool hz_ready = false;
  if (tile_path) {
    if (!c->d_hztab) { const hipError_t e = hipMalloc(&c->d_hztab, (size_t)12 * 3 * 64 * 16 * sizeof(real)); if (e != hipSuccess) c->d_hztab = nullptr; }
    if (c->d_hztab) {
      int sl = -1;
      for (int q4 = 0; q4 < 4; ++q4) { const auto &h4 = c->hz_tab[ivel - 1][q4]; if (h4.ok && h4.alpha == alpha && h4.nz == n3 - q0) sl = q4; }
      hz_ready = sl >= 0;
      if (sl < 0) { sl = c->hz_next[ivel - 1]; c->hz_next[ivel - 1] = (sl + 1) % 4; c->hz_tab[ivel - 1][sl].alpha = alpha; c->hz_tab[ivel - 1][sl].nz = n3 - q0; c->hz_tab[ivel - 1][sl].ok = true; }
      hz = c->d_hztab + (size_t)(4 * (ivel - 1) + sl) * 3 * 64 * 16;
    }
  }
  if (!hz_ready) LAUNCH(c, k_scale_abc, dim3((n3 + 63) / 64), dim3(64), 0, c->stream, n3, alpha, c->d_av[ivel - 1], c->d_bv[ivel - 1], c->d_cv[ivel - 1], abc, abc + n3, abc + 2 * n3);
  const char *bcz = &c->cbcvel[6 * (ivel - 1) + 4];
  const int q = (ivel == 3 && bcz[1] == 'D') ? 1 : 0;
  const bool periodic = bcz[0] == 'P' && bcz[1] == 'P';
  dim3 b(64, 4), gr((n[0] + 63) / 64, (n[1] + 3) / 4);
  real *fld = c->f[CALES_U + ivel - 1];
  Spec S; S.blocked = 0; S.cw = 0; S.n2l = n[1]; S.n3 = n3;
  if (periodic) LAUNCH(c, (k_gaussel<real, 1>), gr, b, 0, c->stream, c->g, n3 - q, n[0], n[1], 1, 0, n[0], S, 1., abc, abc + n3, abc + 2 * n3, (const real *)nullptr, (const real *)nullptr, fld, c->scr1, c->scr2, 0);
  else if (c->fl.helmholtz_z_per_column) LAUNCH(c, (k_gaussel<real, 0>), gr, b, 0, c->stream, c->g, n3 - q, n[0], n[1], 1, 0, n[0], S, 1., abc, abc + n3, abc + 2 * n3, (const real *)nullptr, (const real *)nullptr, fld, c->scr1, c->scr2, 0);
  else if (c->P >= 1 && n3 - q >= 2 && n3 - q <= 512 && n3 <= 64 * (n3 - q <= 128 ? 2 : n3 - q <= 256 ? 4 : 8) && !c->fl.gaussel_march && [&]() {
             // the in-LDS tile of the pressure solve on the real field: u, dudtd in, u out (3 words instead of 5)
             TileMap T{}; T.nolam = 1; T.nq = n3;
             if (fused) { T.dud = c->f[CALES_DUDTD + ivel - 1] + 1; T.hf12 = c->hf12; T.force = c->C.is_forced[ivel - 1] ? c->d_force + (ivel - 1) : nullptr;
                          T.blo = RB[0]; T.bhi = RB[1]; T.has_lo = has[0]; T.has_hi = has[1]; }
             return gaussel_tile<2>(c, n3 - q, n[0], n[1], 1., abc, abc + n3, abc + 2 * n3, fld + 1, 0, T, hz, hz_ready); }()) {}
  else {
    real *zz = abc + 3 * n3, *dd = abc + 4 * n3;       // behind the scaled coefficients (cales_create reserves 6 (n3+2) doubles)
    LAUNCH(c, k_thomas_coef, dim3(1), dim3(64), 0, c->stream, n3 - q, abc, abc + n3, abc + 2 * n3, zz, dd);
    if (fused) LAUNCH(c, k_gaussel_cols_rhs, gr, b, 0, c->stream, c->g, n3 - q, n3, abc, zz, dd, fld, c->f[CALES_DUDTD + ivel - 1], c->hf12,
                                  c->C.is_forced[ivel - 1] ? c->d_force + (ivel - 1) : (const real *)nullptr, c->scr2, has[0], has[1]);
    else LAUNCH(c, k_gaussel_cols, gr, b, 0, c->stream, c->g, n3 - q, abc, zz, dd, fld);
  }
  LAUNCHCHK(c);
  return 0;
}

// 3-D implicit diffusion (_IMPDIFF without _IMPDIFF_1D, main.f90:423-491): (1 + alpha L) q = q* by the same transforms as the
// pressure solve, with the transform set of the velocity component: kinds, eigenvalues and normalisation follow its BC pairs and its
// staggering (initsolver.f90:66-98, find_fft with c_or_f, fft.f90:192-245):
//   across the component ('c'):  PP 0 | NN 1 (REDFT10/01) | DD 2 (RODFT10/01) | ND 3 (REDFT11) | DN 4 (RODFT11)      -- the pressure's kernels
//   along it ('f'):              PP 0 | NN 6 (REDFT00) | DD 5 (RODFT00, one unknown less) | ND 1 (REDFT10/01) | DN 7 (RODFT01 / RODFT10)
// Built on first use. (As in the reference, the face-centred NN and ND sets are not exact inverses of the discrete operator -- REDFT00 of n
// points goes with eigenvalues of period n, REDFT10 with half-integer ones; measured on the oracle: residual of (1 + alpha L) x = r of a
// few per cent -- while PP, DD and DN are exact to round-off. They are provided because the reference provides them.)
static int velset_build(cales_ctx *c, SolverPlans &sp, int iv, VelSet &V) {
  const int n1 = c->C.ng[0], n2g = c->C.ng[1];
  const char *bc = &c->cbcvel[6 * iv];
  auto kind = [&](int d) -> int {
    const std::string b = std::string(1, bc[2 * d]) + bc[2 * d + 1];
    const bool own = d == iv;
    if (b == "PP") return 0;
    if (b == "NN") return own ? 6 : 1;
    if (b == "DD") return own ? 5 : 2;
    if (b == "ND") return own ? 1 : 3;
    if (b == "DN") return own ? 7 : 4;
    return -1;
  };
  V.xkind = kind(0); V.ykind = kind(1);
  if (V.xkind < 0 || V.ykind < 0) { c->err = "helmholtz: unknown velocity BC pair in x or y"; return 1; }
  if (V.xkind && !V.ykind && c->cbcvel[6 * iv + 4] == 'P') { c->err = "helmholtz: non-periodic x with periodic y and z is not provided"; return 1; }
  if (V.xkind && c->cbcvel[6 * iv + 4] == 'P' && c->P > 1) { c->err = "helmholtz: a non-periodic x with periodic z needs one rank"; return 1; }
  if ((V.ykind == 3 || V.ykind == 4) && (n2g % 2)) { c->err = "helmholtz: ND/DN in y need an even ng(2)"; return 1; }
  std::vector<real> lx(n1 + 2, 0.), ly(n2g);
  const std::string bx = std::string(1, bc[0]) + bc[1], by = std::string(1, bc[2]) + bc[3];
  hs_eigenvalues(n1, bx.c_str(), iv == 0 ? 'f' : 'c', lx.data()); hs_eigenvalues(n2g, by.c_str(), iv == 1 ? 'f' : 'c', ly.data());
  for (auto &v : lx) v = v * (c->dli[0] * c->dli[0]);
  for (auto &v : ly) v = v * (c->dli[1] * c->dli[1]);
  if (V.ykind == 2) std::reverse(ly.begin(), ly.end());      // see solver_setup
  HIPCHK(c, hipMalloc(&V.lamx, (n1 + 2) * sizeof(real))); HIPCHK(c, hipMalloc(&V.lamy, n2g * sizeof(real)));
  HIPCHK(c, hipMemcpy(V.lamx, lx.data(), (n1 + 2) * sizeof(real), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(V.lamy, ly.data(), n2g * sizeof(real), hipMemcpyHostToDevice));
  // fft.f90:99,136,142: normfft = prod norm(1) (n + norm(2) - ix): 1 n (PP), 2 n ('c' pairs, 'f' ND/DN), 2 (n + 1 - 1) ('f' DD), 2 (n - 1) ('f' NN)
  auto nrm = [](int kd, int nn) -> real { return kd == 0 ? (real)nn : kd == 6 ? 2. * (nn - 1) : 2. * nn; };
  V.normfft = 1. / (nrm(V.xkind, n1) * nrm(V.ykind, n2g));
  auto mk = [&](int N, real **dev) -> int {
    std::vector<real> t(2 * (size_t)N); const real pi = std::acos(-1.0);
    for (int q = 0; q < N; ++q) { const real ang = -2. * pi * q / N; t[2 * q] = std::cos(ang); t[2 * q + 1] = std::sin(ang); }
    HIPCHK(c, hipMalloc(dev, t.size() * sizeof(real)));
    HIPCHK(c, hipMemcpy(*dev, t.data(), t.size() * sizeof(real), hipMemcpyHostToDevice));
    return 0;
  };
  auto next = [](int kd, int nn) { return kd == 5 ? 2 * nn : kd == 6 ? 2 * (nn - 1) : 4 * nn; };      // points of the symmetric extension (k_dst1)
  if (V.xkind >= 5) { const int N = next(V.xkind, n1);
    if (N < 2 || !make_plan(N, V.p1x) || (size_t)2 * (N + 1) * sizeof(cpx) > 150 * 1024) { c->err = "helmholtz: x line not supported by the face-centred transform kernel"; return 1; } if (mk(N, &V.tw1x)) return 1; }
  if (V.ykind >= 5) { const int N = next(V.ykind, n2g);
    if (N < 2 || !make_plan(N, V.p1y) || (size_t)2 * (N + 1) * sizeof(cpx) > 150 * 1024) { c->err = "helmholtz: y line not supported by the face-centred transform kernel"; return 1; } if (mk(N, &V.tw1y)) return 1; }
  if (V.xkind >= 5 || V.ykind >= 5) { HIPSOFT(c, hipFuncSetAttribute((const void *)k_dst1<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); HIPSOFT(c, hipFuncSetAttribute((const void *)k_dst1<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); HIPSOFT(c, hipFuncSetAttribute((const void *)k_dst1<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); }
  // the DCT-IV / DST-IV kernels (kinds 3, 4) read tables that solver_setup makes only when the pressure needs them
  const real pi = std::acos(-1.0);
  if ((V.xkind == 3 || V.xkind == 4) && !c->d_tw4x) {
    const int nh = n1 / 2; std::vector<real> t(4 * (size_t)nh);
    for (int q = 0; q < nh; ++q) { const real a1 = -pi * (4. * q + 1.) / (4. * n1), a2 = -pi * q / (real)n1;
                                   t[2 * q] = std::cos(a1); t[2 * q + 1] = std::sin(a1); t[2 * (nh + q)] = std::cos(a2); t[2 * (nh + q) + 1] = std::sin(a2); }
    HIPCHK(c, hipMalloc(&c->d_tw4x, t.size() * sizeof(real)));
    HIPCHK(c, hipMemcpy(c->d_tw4x, t.data(), t.size() * sizeof(real), hipMemcpyHostToDevice));
  }
  if ((V.ykind == 3 || V.ykind == 4) && !c->d_tw4y) {
    if (!make_plan(n2g / 2, sp.py4)) { c->err = "helmholtz: ng(2)/2 must factor into primes <= 127"; return 1; }
    sp.CBy4 = 4; sp.shy4 = (size_t)2 * sp.CBy4 * 2 * (n2g / 2 + 1) * sizeof(cpx);
    while (sp.shy4 > 60 * 1024 && sp.CBy4 > 1) { sp.CBy4 /= 2; sp.shy4 = (size_t)2 * sp.CBy4 * 2 * (n2g / 2 + 1) * sizeof(cpx); }
    if (sp.shy4 > 64 * 1024) { c->err = "helmholtz: y line too long for the LDS-resident DCT-IV"; return 1; }
    const int nh2 = n2g / 2; std::vector<real> t(4 * (size_t)nh2);
    for (int q = 0; q < nh2; ++q) { const real a1 = -pi * (4. * q + 1.) / (4. * n2g), a2 = -pi * q / (real)n2g;
                                    t[2 * q] = std::cos(a1); t[2 * q + 1] = std::sin(a1); t[2 * (nh2 + q)] = std::cos(a2); t[2 * (nh2 + q) + 1] = std::sin(a2); }
    HIPCHK(c, hipMalloc(&c->d_tw4y, t.size() * sizeof(real)));
    HIPCHK(c, hipMemcpy(c->d_tw4y, t.data(), t.size() * sizeof(real), hipMemcpyHostToDevice));
    if (!c->d_twy4) { if (mk(n2g / 2, &c->d_twy4)) return 1; }
  }
  V.ready = true;
  return 0;
}
int op_helmholtz(cales_ctx *c, int ivel, real alpha) {
  if (c->C.impdiff != 1) { c->err = "helmholtz needs impdiff = 1"; return 1; }
  PlanSlot *slot = find_slot(c);
  if (!slot) { c->err = "solver not initialised"; return 1; }
  VelSet &V = slot->vs[ivel - 1];
  if (!V.ready) if (int e = velset_build(c, slot->sp, ivel - 1, V)) return e;
  ProfScope ps(c, "helmholtz_xyz");
  const int n3 = c->n[2];
  if (int e = op_rhs_b_velxy(c, ivel, alpha)) return e;      // main.f90:424-431: boundary terms of the x and y faces, then z
  if (int e = op_rhs_b_velz(c, ivel, alpha)) return e;
  real *abc = c->d_red + 64 + 16 * (n3 + 2);
  LAUNCH(c, k_scale_abc, dim3((n3 + 63) / 64), dim3(64), 0, c->stream, n3, alpha, c->d_av[ivel - 1], c->d_bv[ivel - 1], c->d_cv[ivel - 1], abc, abc + n3, abc + 2 * n3);
  const char *bcz = &c->cbcvel[6 * (ivel - 1) + 4];
  const int q = (ivel == 3 && bcz[1] == 'D') ? 1 : 0;
  // the solve runs with the component's kinds, eigenvalues and normalisation in place of the pressure's
  const int xk = c->xkind, yk = c->ykind; real *lx = c->d_lamx, *ly = c->d_lamy; const real nf = c->normfft;
  c->xkind = V.xkind; c->ykind = V.ykind; c->d_lamx = V.lamx; c->d_lamy = V.lamy; c->normfft = V.normfft; c->cur_velset = &V;
  const int e = solve_field(c, c->f[CALES_U + ivel - 1], abc, abc + n3, abc + 2 * n3, n3 - q, alpha, bcz[0] == 'P' && bcz[1] == 'P', false);
  c->xkind = xk; c->ykind = yk; c->d_lamx = lx; c->d_lamy = ly; c->normfft = nf; c->cur_velset = nullptr;
  return e;
}
