// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for the access widths of this library (MI355X_MICROARCH.md, "HBM": the x2 rule
// for FETCH_SIZE is stated for 16 B/lane streams only). Streams arrays of KNOWN size (far beyond the 256 MiB Infinity Cache) with
// 8 B/lane (what the FP64 tile kernels do) and 16 B/lane loads/stores; run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`
// (separate passes) and divide the counters by the byte counts printed here (profiles/summarize.py --calib).
//   hipcc --offload-arch=gfx950 -O3 -o calib calib.hip && ./calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void calib_copy8(const double *__restrict__ a, double *__restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void calib_copy16(const double2 *__restrict__ a, double2 *__restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void calib_read8(const double *__restrict__ a, double *__restrict__ out, size_t n) {
  double s = 0.;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += a[i];
  if (s == 1.2345e300) out[blockIdx.x] = s;      // never true: no write traffic
}
__global__ __launch_bounds__(256) void calib_read16(const double2 *__restrict__ a, double *__restrict__ out, size_t n) {
  double s = 0.;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { double2 v = a[i]; s += v.x + v.y; }
  if (s == 1.2345e300) out[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void calib_write8(double *__restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = 1.5;
}
// rows of 512 doubles at a pitch of 528 starting 15 doubles into the allocation: the layout of the library's fields
__global__ __launch_bounds__(256) void calib_copy8_rows(const double *__restrict__ a, double *__restrict__ b, int nrow) {
  for (int r = blockIdx.x; r < nrow; r += gridDim.x)
    for (int i = threadIdx.x; i < 512; i += 256) b[16 + (size_t)r * 528 + i] = a[16 + (size_t)r * 528 + i];
}

int main() {
  const size_t n = (size_t)1 << 28;      // 2 GiB per array
  double *a, *b; CK(hipMalloc(&a, n * 8 + 4096)); CK(hipMalloc(&b, n * 8 + 4096));
  CK(hipMemset(a, 0, n * 8)); CK(hipMemset(b, 0, n * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int nb = 256 * 16;
  auto timeit = [&](const char *name, double rd, double wr, auto f) {
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int r = 0; r < 5; ++r) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("{\"kernel\": \"%s\", \"read_bytes\": %.0f, \"write_bytes\": %.0f, \"ms\": %.4f, \"GBps\": %.1f}\n", name, rd, wr, ms, (rd + wr) / ms / 1e6);
  };
  timeit("calib_copy8", n * 8., n * 8., [&] { hipLaunchKernelGGL(calib_copy8, dim3(nb), dim3(256), 0, 0, a, b, n); });
  timeit("calib_copy16", n * 8., n * 8., [&] { hipLaunchKernelGGL(calib_copy16, dim3(nb), dim3(256), 0, 0, (const double2 *)a, (double2 *)b, n / 2); });
  timeit("calib_read8", n * 8., 0., [&] { hipLaunchKernelGGL(calib_read8, dim3(nb), dim3(256), 0, 0, a, b, n); });
  timeit("calib_read16", n * 8., 0., [&] { hipLaunchKernelGGL(calib_read16, dim3(nb), dim3(256), 0, 0, (const double2 *)a, b, n / 2); });
  timeit("calib_write8", 0., n * 8., [&] { hipLaunchKernelGGL(calib_write8, dim3(nb), dim3(256), 0, 0, b, n); });
  const int nrow = (int)(n / 528) - 1;
  timeit("calib_copy8_rows", nrow * 4096., nrow * 4096., [&] { hipLaunchKernelGGL(calib_copy8_rows, dim3(nb), dim3(256), 0, 0, a, b, nrow); });
  return 0;
}
