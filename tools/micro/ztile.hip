// Micro-benchmark: how fast can whole z-columns be brought into LDS and written back, for tiles W doubles wide (one or more
// 128-B lines per plane)?  Decides whether a tridiagonal sweep that keeps its columns in LDS (2 words/cell instead of the 5.3 of
// the marching Thomas kernel, which stores c' and d') can pay.  Build: hipcc --offload-arch=gfx950 -O3 ztile.hip -o ztile
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int W, int NT>
__global__ __launch_bounds__(NT) void k_ztile(double *p, long s1, long s12, int n1w, int n2, int n3) {
  extern __shared__ double sh[];
  const int t = threadIdx.x, x = t % W, kk = t / W;
  constexpr int KP = NT / W;
  const long base = (long)(blockIdx.y + 1) * s1 + 16 + (long)blockIdx.x * W + x;
  for (int k = kk; k < n3; k += KP) sh[k * W + x] = p[base + (long)(k + 1) * s12];
  __syncthreads();
  // stand-in for the solve: every thread touches its column chunk
  for (int k = kk; k < n3; k += KP) p[base + (long)(k + 1) * s12] = sh[(n3 - 1 - k) * W + x] + 1.;
}
// the same tile with 16-byte accesses: eight lanes per 128-B line, half as many load / store instructions
template <int NT>
__global__ __launch_bounds__(NT) void k_ztile16(double *p, long s1, long s12, int n1w, int n2, int n3) {
  extern __shared__ double sh[];
  const int t = threadIdx.x, x = t % 8, kk = t / 8;
  constexpr int KP = NT / 8;
  const long base = (long)(blockIdx.y + 1) * s1 + 16 + (long)blockIdx.x * 16 + 2 * x;
  for (int k = kk; k < n3; k += KP) { const double2 v = *(const double2 *)(p + base + (long)(k + 1) * s12); sh[k * 16 + 2 * x] = v.x; sh[k * 16 + 2 * x + 1] = v.y; }
  __syncthreads();
  for (int k = kk; k < n3; k += KP) { double2 v; v.x = sh[(n3 - 1 - k) * 16 + 2 * x] + 1.; v.y = sh[(n3 - 1 - k) * 16 + 2 * x + 1] + 1.; *(double2 *)(p + base + (long)(k + 1) * s12) = v; }
}
template <int NT>
static void run16(double *p, long s1, long s12, int n, const char *name) {
  dim3 g(n / 16 + 1, n), b(NT);
  size_t lds = (size_t)16 * n * 8;
  hipFuncSetAttribute((const void *)k_ztile16<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((k_ztile16<NT>), g, b, lds, 0, p, s1, s12, n, n, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    if (rep == 2) printf("%s 16-byte accesses NT=%d: %.3f ms  %.2f TB/s\n", name, NT, ms, 16.0 * n * n * n / ms / 1e9);
  }
}
// the marching pattern for reference: 64 lanes along x, in place, 2 words/cell
__global__ __launch_bounds__(256) void k_march(double *p, long s1, long s12, int n3) {
  const long base = (long)(blockIdx.y + 1) * s1 + 16 + (long)blockIdx.x * 256 + threadIdx.x;
  double acc = 0.;
  for (int k = 1; k <= n3; ++k) { acc = 0.5 * acc + p[base + (long)k * s12]; p[base + (long)k * s12] = acc; }
}
template <int W, int NT>
static void run(double *p, long s1, long s12, int n, const char *name) {
  dim3 g(n / W + (W <= 16 ? 1 : 0), n), b(NT);       // n+2 doubles of a spectral row -> one more tile when W is small
  size_t lds = (size_t)W * n * 8;
  hipFuncSetAttribute((const void *)k_ztile<W, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((k_ztile<W, NT>), g, b, lds, 0, p, s1, s12, n, n, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    if (rep == 2) printf("%s W=%d NT=%d: %.3f ms  %.2f TB/s\n", name, W, NT, ms, 16.0 * n * n * n / ms / 1e9);
  }
}
int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 512; long s1 = n + 16, s12 = s1 * (n + 2); size_t ntot = (size_t)s12 * (n + 2);
  double *p; hipMalloc(&p, ntot * 8 + 65536); hipMemset(p, 0, ntot * 8);
  printf("n = %d\n", n);
  run<16, 256>(p, s1, s12, n, "ztile"); run<16, 512>(p, s1, s12, n, "ztile"); run<16, 1024>(p, s1, s12, n, "ztile");
  if (n <= 512) { run<32, 512>(p, s1, s12, n, "ztile"); run<32, 1024>(p, s1, s12, n, "ztile"); }
  run16<256>(p, s1, s12, n, "ztile"); run16<512>(p, s1, s12, n, "ztile"); run16<1024>(p, s1, s12, n, "ztile");
  // the same tiles if the spectrum were laid out [j][k][m] (z stride = one row, y stride = one plane)
  run<16, 512>(p, s12, s1, n, "ztile, rows and planes swapped"); run<16, 1024>(p, s12, s1, n, "ztile, rows and planes swapped");
  {
    dim3 g(2, n), b(256); hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(k_march, g, b, 0, 0, p, s1, s12, n);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
      if (rep == 2) printf("march 256 lanes in x: %.3f ms  %.2f TB/s\n", ms, 16.0 * n * n * n / ms / 1e9);
    }
  }
  return 0;
}
