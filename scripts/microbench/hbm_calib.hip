// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS code base's access
// pattern (8 B per lane, lanes on consecutive doubles), as guides/MI355X_MICROARCH.md "HBM"
// prescribes: the counters are only calibrated for 16 B/lane streaming reads, so measure a known
// byte count first.  Two kernels over a 2 GiB buffer (past the 256 MiB Infinity Cache):
//   calib_read8   reads  every double once  (2 GiB read,  ~0 written)
//   calib_write8  writes every double once  (2 GiB written, ~0 read)
// Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`; the correction factors are
// bytes_known / (counter * 1024).
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void
calib_read8(const double* __restrict__ a, long n, double* out)
{
  double s = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    s += a[i];
  if (s == 12345.678)
    out[0] = s;
}

__global__ void
calib_write8(double* __restrict__ a, long n)
{
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    a[i] = (double)i;
}

int
main()
{
  const long n = 1L << 28; // 2 GiB of doubles
  double *a = nullptr, *out = nullptr;
  if (hipMalloc(&a, n * sizeof(double)) != hipSuccess || hipMalloc(&out, 8) != hipSuccess)
    return 1;
  hipMemset(a, 0, n * sizeof(double));
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(calib_read8, dim3(256 * 16), dim3(256), 0, 0, a, n, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    std::printf("calib_read8  %ld bytes in %.3f ms = %.1f GB/s\n", n * 8, ms, n * 8 / ms * 1e-6);
  }
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(calib_write8, dim3(256 * 16), dim3(256), 0, 0, a, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    std::printf("calib_write8 %ld bytes in %.3f ms = %.1f GB/s\n", n * 8, ms, n * 8 / ms * 1e-6);
  }
  hipFree(a);
  hipFree(out);
  return 0;
}
