// Rate of kernel-issued stores into pinned, device-mapped host memory (the result mirrors of pqp_batch_enable_host_results)
// by store width and by bytes per wavefront: hipcc --offload-arch=gfx950 -O3 scripts/host_store_rate.hip -o build/ab/host_store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template<int W> // doubles per lane per store
__global__ void fill(double* dst, long per_wave, int reps_compute)
{
  const long wave = (long)blockIdx.x;
  double v = (double)threadIdx.x;
  for (int r = 0; r < reps_compute; ++r) // (a little dependent work before the stores, as an epilogue would follow a solve)
    v = fma(v, 1.0000001, 0.5);
  double* p = dst + wave * per_wave;
  if constexpr (W == 1) {
    for (long k = threadIdx.x; k < per_wave; k += 64)
      p[k] = v;
  } else if constexpr (W == 2) {
    double2* q = (double2*)p;
    for (long k = threadIdx.x; k < per_wave / 2; k += 64)
      q[k] = make_double2(v, v);
  } else {
    double4* q = (double4*)p;
    for (long k = threadIdx.x; k < per_wave / 4; k += 64)
      q[k] = make_double4(v, v, v, v);
  }
}
template<int W>
static void run(double* dev_view, long waves, long per_wave, int reps, const char* what)
{
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float best = 1e30f;
  for (int it = 0; it < 5; ++it) {
    hipEventRecord(a);
    hipLaunchKernelGGL(fill<W>, dim3((unsigned)waves), dim3(64), 0, 0, dev_view, per_wave, reps);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  const double bytes = (double)waves * per_wave * 8;
  printf("%-28s %d B/lane/store  %6.1f MB  %.3f ms  %.1f GB/s\n", what, W * 8, bytes / 1e6, best, bytes / best / 1e6);
}
int main()
{
  const long waves = 4096, per_wave = 640; // 5 KB per wavefront, 20 MB: the C5 mirrors
  double* host = nullptr;
  hipHostMalloc((void**)&host, waves * per_wave * 8 * 4, hipHostMallocMapped);
  double* dv = nullptr;
  hipHostGetDevicePointer((void**)&dv, host, 0);
  double* dmem = nullptr;
  hipMalloc((void**)&dmem, waves * per_wave * 8 * 4);
  run<1>(dmem, waves, per_wave, 0, "device memory");
  run<1>(dv, waves, per_wave, 0, "pinned host");
  run<2>(dv, waves, per_wave, 0, "pinned host");
  run<4>(dv, waves, per_wave, 0, "pinned host");
  run<1>(dv, waves * 4, per_wave, 0, "pinned host, 80 MB");
  run<4>(dv, waves * 4, per_wave, 0, "pinned host, 80 MB");
  run<1>(dv, waves, per_wave, 20000, "pinned host, after compute");
  // the same bytes by the copy engine
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(a);
    hipMemcpyAsync(host, dmem, waves * per_wave * 8, hipMemcpyDeviceToHost, 0);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("hipMemcpyAsync D2H           %6.1f MB  %.3f ms  %.1f GB/s\n", waves * per_wave * 8 / 1e6, ms, waves * per_wave * 8 / ms / 1e6);
  }
  return 0;
}
