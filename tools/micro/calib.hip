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

// The copy the guide quotes its 6.29 TB/s for (MI355X_MICROARCH.md, "HBM"): 16 B per lane, four loads in flight per lane before the first store,
// grid-stride; `NT` = non-temporal stores and loads (the output is not read again: no reason to keep it in L2 / Infinity Cache).
template <int NT>
__global__ __launch_bounds__(256) void calib_copy16x4(const double2 *__restrict__ a, double2 *__restrict__ b, size_t n) {
  const size_t st = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * st < n; i += 4 * st) {
    double2 v0, v1, v2, v3;
    typedef double v2d __attribute__((ext_vector_type(2)));
    if (NT) { const v2d *aa = reinterpret_cast<const v2d *>(a); v2d *bb = reinterpret_cast<v2d *>(b);
              const v2d w0 = __builtin_nontemporal_load(aa + i), w1 = __builtin_nontemporal_load(aa + i + st), w2 = __builtin_nontemporal_load(aa + i + 2 * st), w3 = __builtin_nontemporal_load(aa + i + 3 * st);
              __builtin_nontemporal_store(w0, bb + i); __builtin_nontemporal_store(w1, bb + i + st); __builtin_nontemporal_store(w2, bb + i + 2 * st); __builtin_nontemporal_store(w3, bb + i + 3 * st); }
    else { v0 = a[i]; v1 = a[i + st]; v2 = a[i + 2 * st]; v3 = a[i + 3 * st]; b[i] = v0; b[i + st] = v1; b[i + 2 * st] = v2; b[i + 3 * st] = v3; }
  }
  for (; i < n; i += st) b[i] = a[i];
}
// the same with 8 B per lane (what a kernel with one FP64 value per lane and field can do at best): eight loads in flight
__global__ __launch_bounds__(256) void calib_copy8x8(const double *__restrict__ a, double *__restrict__ b, size_t n) {
  const size_t st = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 7 * st < n; i += 8 * st) {
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = a[i + q * st];
#pragma unroll
    for (int q = 0; q < 8; ++q) b[i + q * st] = v[q];
  }
  for (; i < n; i += st) b[i] = a[i];
}

// every block owns one contiguous piece (DRAM pages are walked in order, not touched by 4096 blocks at once), 16 B per lane, 4 loads in flight
__global__ __launch_bounds__(256) void calib_copy16_chunks(const double2 *__restrict__ a, double2 *__restrict__ b, size_t n) {
  const size_t per = (n + gridDim.x - 1) / gridDim.x, beg = (size_t)blockIdx.x * per, end = beg + per < n ? beg + per : n;
  size_t i = beg + threadIdx.x;
  for (; i + 768 < end; i += 1024) { const double2 v0 = a[i], v1 = a[i + 256], v2 = a[i + 512], v3 = a[i + 768]; b[i] = v0; b[i + 256] = v1; b[i + 512] = v2; b[i + 768] = v3; }
  for (; i < end; i += 256) b[i] = a[i];
}
// several streams at once, as the library's multi-field passes (k_correc_cell: 5 fields in, 4 out): 4 in, 4 out, 8 B per lane each
__global__ __launch_bounds__(256) void calib_copy8_4streams(const double *__restrict__ a, double *__restrict__ b, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const double v0 = a[i], v1 = a[i + n4], v2 = a[i + 2 * n4], v3 = a[i + 3 * n4];
    b[i] = v0; b[i + n4] = v1; b[i + 2 * n4] = v2; b[i + 3 * n4] = v3;
  }
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
  for (int mult : {8, 16, 32, 64}) {      // blocks per CU-count: the best of these is the copy ceiling the fractions are quoted against
    char nm[64];
    snprintf(nm, sizeof nm, "calib_copy16x4_g%d", mult);
    timeit(nm, n * 8., n * 8., [&] { hipLaunchKernelGGL(calib_copy16x4<0>, dim3(256 * mult), dim3(256), 0, 0, (const double2 *)a, (double2 *)b, n / 2); });
    snprintf(nm, sizeof nm, "calib_copy16x4nt_g%d", mult);
    timeit(nm, n * 8., n * 8., [&] { hipLaunchKernelGGL(calib_copy16x4<1>, dim3(256 * mult), dim3(256), 0, 0, (const double2 *)a, (double2 *)b, n / 2); });
    snprintf(nm, sizeof nm, "calib_copy8x8_g%d", mult);
    timeit(nm, n * 8., n * 8., [&] { hipLaunchKernelGGL(calib_copy8x8, dim3(256 * mult), dim3(256), 0, 0, a, b, n); });
  }
  for (int mult : {4, 8, 16, 64}) {
    char nm[64];
    snprintf(nm, sizeof nm, "calib_copy16_chunks_g%d", mult);
    timeit(nm, n * 8., n * 8., [&] { hipLaunchKernelGGL(calib_copy16_chunks, dim3(256 * mult), dim3(256), 0, 0, (const double2 *)a, (double2 *)b, n / 2); });
    snprintf(nm, sizeof nm, "calib_copy8_4streams_g%d", mult);
    timeit(nm, n * 8., n * 8., [&] { hipLaunchKernelGGL(calib_copy8_4streams, dim3(256 * mult), dim3(256), 0, 0, a, b, n / 4); });
  }
  const int nrow = (int)(n / 528) - 1;
  timeit("calib_copy8_rows", nrow * 4096., nrow * 4096., [&] { hipLaunchKernelGGL(calib_copy8_rows, dim3(nb), dim3(256), 0, 0, a, b, nrow); });
  return 0;
}
