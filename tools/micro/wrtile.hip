// Micro-benchmark: plane-marching tiles that read NR fields and write NW fields, with (A) 62-wide unaligned row segments
// as the tile kernels use, versus (B) 64-wide segments aligned to 512 B. Build: hipcc --offload-arch=gfx950 -O3 wrtile.hip -o wrtile
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define NR 3
#define NW 12
struct Args { const double *in[NR]; double *out[NW]; long s1, s12; int n1, n2, n3, w, x0, kchunk; };
template <int TY>
__global__ __launch_bounds__(64 * (TY + 2)) void k(Args A) {
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int i = blockIdx.x * A.w + tx + A.x0, j = blockIdx.y * TY + ty;
  const int kbeg = blockIdx.z * A.kchunk + 1, kend = min(kbeg + A.kchunk - 1, A.n3);
  const bool ldok = i < A.s1 && j <= A.n2 + 1;
  const bool outok = (A.w == 64 || (tx >= 1 && tx <= 62)) && ty >= 1 && ty <= TY && i <= A.n1 + A.x0 + (A.w == 64 ? -1 : 0) + 1 && j <= A.n2;
  const size_t c0 = (size_t)j * A.s1 + i;
  double acc = 0.;
  for (int kk = kbeg; kk <= kend; ++kk) {
    const size_t idx = c0 + (size_t)kk * A.s12;
    double v = 0.;
    if (ldok) { for (int q = 0; q < NR; ++q) v += A.in[q][idx]; }
    acc += v;
    if (outok) { for (int q = 0; q < NW; ++q) A.out[q][idx] = acc + q; }
  }
}
int main() {
  const int n = 512;
  for (int variant = 0; variant < 3; ++variant) {
    // 0: pitch n+2, 62-wide tiles (current); 1: pitch 528, interior starts at a 128-B boundary, 64-wide tiles; 2: pitch n+2, 64-wide tiles (misaligned by 8 B)
    long s1 = variant == 1 ? 528 : n + 2; long s12 = s1 * (n + 2); size_t ntot = (size_t)s12 * (n + 2);
    Args A; A.s1 = s1; A.s12 = s12; A.n1 = A.n2 = A.n3 = n; A.w = variant == 0 ? 62 : 64; A.x0 = variant == 1 ? 16 : (variant == 2 ? 1 : 0);
    std::vector<double *> ptr;
    for (int q = 0; q < NR + NW; ++q) { double *p; hipMalloc(&p, ntot * 8 + 4096); hipMemset(p, 0, ntot * 8); ptr.push_back(p); }
    for (int q = 0; q < NR; ++q) A.in[q] = ptr[q];
    for (int q = 0; q < NW; ++q) A.out[q] = ptr[NR + q];
    for (int ty : {6, 14}) {
      A.kchunk = 64;
      dim3 b(64, ty + 2), g((n + A.w - 1) / A.w, (n + ty - 1) / ty, n / A.kchunk);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int it = 0; it < 5; ++it) { if (ty == 6) hipLaunchKernelGGL(k<6>, g, b, 0, 0, A); else hipLaunchKernelGGL(k<14>, g, b, 0, 0, A); }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        if (rep == 2) printf("variant %d TY %d: %.3f ms  %.2f TB/s (algorithmic %d words/cell)\n", variant, ty, ms, (NR + NW) * 8.0 * n * n * n / ms / 1e9, NR + NW);
      }
    }
    for (auto p : ptr) hipFree(p);
  }
  return 0;
}
