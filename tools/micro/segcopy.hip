// What the column-tile access pattern of the y transforms (k_fft_y8 / k_fft_y8r) can reach at best: in-place copy of a spectrum laid out like the
// library's (planes of 512 rows of 264 complex slots, 257 used), each block owning CB adjacent complex columns of one plane = 512 segments of CB x 16 B
// at a pitch of 4224 B, eight elements per thread in flight, no arithmetic, no LDS.   hipcc --offload-arch=gfx950 -O3 -o segcopy segcopy.hip && ./segcopy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int CB>
__global__ __launch_bounds__(512) void segcopy(double2 *__restrict__ p, int ncols, int pitch, int nrows, int nplanes, int kchunk) {
  const int T = 512 / CB, c = threadIdx.x % CB, t = threadIdx.x / CB, m = min(blockIdx.x * CB + c, ncols - 1);
  const int NE = nrows / T;
  for (int k = blockIdx.y * kchunk; k < min((int)(blockIdx.y + 1) * kchunk, nplanes); ++k) {
    double2 v[32];
    double2 *base = p + (size_t)k * nrows * pitch + m;
#pragma unroll
    for (int e = 0; e < 32; ++e) if (e < NE) v[e] = base[(size_t)(t + e * T) * pitch];
#pragma unroll
    for (int e = 0; e < 32; ++e) if (e < NE) { v[e].x += 1.; base[(size_t)(t + e * T) * pitch] = v[e]; }
  }
}
int main() {
  const int nrows = 512, nplanes = 512, pitch = 264, ncols = 257;
  const size_t n = (size_t)nplanes * nrows * pitch;
  double2 *p; CK(hipMalloc(&p, n * 16 + 4096)); CK(hipMemset(p, 0, n * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char *nm, auto f) {
    f(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for (int r = 0; r < 10; ++r) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("%-28s %.3f ms  %.0f GB/s (useful bytes: 257 of 264 slots, read + write)\n", nm, ms, 2. * nplanes * nrows * ncols * 16. / ms / 1e6);
  };
  for (int kch : {1, 4}) {
    char nm[64];
    snprintf(nm, sizeof nm, "4 columns (64 B), kchunk %d", kch);  run(nm, [&] { hipLaunchKernelGGL(segcopy<4>, dim3((ncols + 3) / 4, (nplanes + kch - 1) / kch), dim3(512), 0, 0, p, ncols, pitch, nrows, nplanes, kch); });
    snprintf(nm, sizeof nm, "8 columns (128 B), kchunk %d", kch);  run(nm, [&] { hipLaunchKernelGGL(segcopy<8>, dim3((ncols + 7) / 8, (nplanes + kch - 1) / kch), dim3(512), 0, 0, p, ncols, pitch, nrows, nplanes, kch); });
    snprintf(nm, sizeof nm, "16 columns (256 B), kchunk %d", kch); run(nm, [&] { hipLaunchKernelGGL(segcopy<16>, dim3((ncols + 15) / 16, (nplanes + kch - 1) / kch), dim3(512), 0, 0, p, ncols, pitch, nrows, nplanes, kch); });
    snprintf(nm, sizeof nm, "32 columns (512 B), kchunk %d", kch); run(nm, [&] { hipLaunchKernelGGL(segcopy<32>, dim3((ncols + 31) / 32, (nplanes + kch - 1) / kch), dim3(512), 0, 0, p, ncols, pitch, nrows, nplanes, kch); });
  }
  return 0;
}
