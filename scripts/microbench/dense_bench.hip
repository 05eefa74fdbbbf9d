// Stand-alone micro-benchmark of the dense building blocks (ldlt_factor / ldlt_solve / gemv)
// at the benchmark's shapes and occupancy: 2048 workgroups of 256 threads, 37 KB of LDS each
// (4 workgroups per CU), every workgroup on its own matrix.  Build + run: see scripts/gpu_micro.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "pqp_block.hpp"
using namespace pqp;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256, 4) void k_factor(double* Ms, const double* src, int ld, int m, int reps) {
  HIP_DYNAMIC_SHARED(double, smem)
  lptr s = (lptr)smem;
  gptr M = (gptr)(Ms + (long)blockIdx.x * ld * ld);
  cgptr S = (cgptr)(src + (long)blockIdx.x * ld * ld);
  for (int r = 0; r < reps; ++r) {
    for (int o = threadIdx.x; o < m * ld; o += 256) M[o] = S[o];
    __syncthreads();
    ldlt_factor<256, false>(M, ld, m, s, s + 1024);
  }
}
__global__ __launch_bounds__(256, 4) void k_factor_reg(double* Ms, const double* src, int ld, int m, int reps, long long* cyc) {
  HIP_DYNAMIC_SHARED(double, smem)
  lptr s = (lptr)smem;
  gptr M = (gptr)(Ms + (long)blockIdx.x * ld * ld);
  cgptr S = (cgptr)(src + (long)blockIdx.x * ld * ld);
  PQP_LDS long long* prof = (PQP_LDS long long*)(s + 2048);
  if (threadIdx.x < 8) prof[threadIdx.x] = 0;
  __syncthreads();
  for (int r = 0; r < reps; ++r) {
    auto load = [&](int i, int j) -> double { return S[(long)j * ld + i]; };
    ldlt_factor_reg<256, 7>(load, M, ld, m, s, s + 1024, prof);
  }
  if (threadIdx.x < 4) cyc[blockIdx.x * 4 + threadIdx.x] = prof[threadIdx.x];
}
__global__ __launch_bounds__(256, 4) void k_copy(double* Ms, const double* src, int ld, int m, int reps) {
  gptr M = (gptr)(Ms + (long)blockIdx.x * ld * ld);
  cgptr S = (cgptr)(src + (long)blockIdx.x * ld * ld);
  for (int r = 0; r < reps; ++r) {
    for (int o = threadIdx.x; o < m * ld; o += 256) M[o] = S[o];
    __syncthreads();
  }
}
__global__ __launch_bounds__(256, 4) void k_solve(const double* Ms, int ld, int m, int reps, double* out) {
  HIP_DYNAMIC_SHARED(double, smem)
  lptr s = (lptr)smem;
  cgptr M = (cgptr)(Ms + (long)blockIdx.x * ld * ld);
  for (int o = threadIdx.x; o < 512; o += 256) { s[o] = 1.0; s[512 + o] = 1.0 + 1e-3 * o; }
  __syncthreads();
  for (int r = 0; r < reps; ++r)
    ldlt_solve<256>(M, ld, m, s, s + 512, s + 1024);
  if (threadIdx.x == 0) out[blockIdx.x] = s[512];
}
__global__ __launch_bounds__(256, 4) void k_gemv(const double* Ms, int ld, int K, int J, int reps, double* out) {
  HIP_DYNAMIC_SHARED(double, smem)
  lptr s = (lptr)smem;
  cgptr M = (cgptr)(Ms + (long)blockIdx.x * ld * ld);
  for (int o = threadIdx.x; o < 512; o += 256) s[o] = 1.0 + 1e-3 * o;
  __syncthreads();
  for (int r = 0; r < reps; ++r)
    gemv<256>(M, ld, K, J, s, s + 512, s + 1024, nullptr, 0, nullptr, 0);
  if (threadIdx.x == 0) out[blockIdx.x] = s[512];
}

int main() {
  const int B = 2048, ld = 150, reps = 10;
  const size_t per = (size_t)ld * ld;
  std::vector<double> h(per * B);
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < ld; ++i)
      for (int j = 0; j < ld; ++j)
        h[b * per + i * ld + j] = (i == j) ? ld + 1.0 : 1.0 / (1.0 + ((i * 7 + j * 13 + b) % 17));
  // symmetrise
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < ld; ++i)
      for (int j = 0; j < i; ++j)
        h[b * per + i * ld + j] = h[b * per + j * ld + i];
  double *src, *M, *out;
  CHECK(hipMalloc(&src, per * B * 8));
  CHECK(hipMalloc(&M, per * B * 8));
  CHECK(hipMalloc(&out, B * 8));
  CHECK(hipMemcpy(src, h.data(), per * B * 8, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t lds = 37 * 1024;
  float ms;
  for (int m : {32, 64, 85, 100, 150}) {
    hipLaunchKernelGGL(k_copy, dim3(B), dim3(256), 0, 0, M, src, ld, m, reps);
    hipEventRecord(e0); hipLaunchKernelGGL(k_copy, dim3(B), dim3(256), 0, 0, M, src, ld, m, reps); hipEventRecord(e1);
    CHECK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1); float copy_ms = ms;
    hipLaunchKernelGGL(k_factor, dim3(B), dim3(256), lds, 0, M, src, ld, m, reps);
    hipEventRecord(e0); hipLaunchKernelGGL(k_factor, dim3(B), dim3(256), lds, 0, M, src, ld, m, reps); hipEventRecord(e1);
    CHECK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1);
    printf("factor m=%3d: %.3f ms per pass of %d matrices (copy-in alone %.3f) -> %.1f us per 1024-wide round\n", m, ms / reps, B, copy_ms / reps, 1e3 * (ms - copy_ms) / reps / 2);
    if (m <= 112) {
      long long* cyc; CHECK(hipMalloc(&cyc, B * 4 * 8));
      hipLaunchKernelGGL(k_factor_reg, dim3(B), dim3(256), lds, 0, M, src, ld, m, reps, cyc);
      hipEventRecord(e0); hipLaunchKernelGGL(k_factor_reg, dim3(B), dim3(256), lds, 0, M, src, ld, m, reps, cyc); hipEventRecord(e1);
      CHECK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1);
      std::vector<long long> hc(B * 4); CHECK(hipMemcpy(hc.data(), cyc, B * 4 * 8, hipMemcpyDeviceToHost));
      double c0 = 0, c1 = 0, c3 = 0; for (int b = 0; b < B; ++b) { c0 += hc[b * 4]; c1 += hc[b * 4 + 1]; c3 += hc[b * 4 + 3]; }
      printf("factor_reg m=%3d: %.3f ms per pass -> %.1f us per round; cycles per call: load %.0f loop %.0f (%.0f/col) writeback %.0f\n", m, ms / reps, 1e3 * ms / reps / 2, c0 / B / reps, c1 / B / reps, c1 / B / reps / m, c3 / B / reps);
      hipFree(cyc);
    }
    hipEventRecord(e0); hipLaunchKernelGGL(k_solve, dim3(B), dim3(256), lds, 0, M, ld, m, reps, out); hipEventRecord(e1);
    CHECK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1);
    printf("solve  m=%3d: %.3f ms per pass\n", m, ms / reps);
  }
  for (int K : {50, 100}) for (int J : {50, 100, 150}) {
    hipEventRecord(e0); hipLaunchKernelGGL(k_gemv, dim3(B), dim3(256), lds, 0, src, ld, K, J, reps, out); hipEventRecord(e1);
    CHECK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1);
    printf("gemv K=%3d J=%3d: %.3f ms per pass (%.1f GB/s)\n", K, J, ms / reps, (double)K * J * 8 * B / (ms / reps * 1e-3) / 1e9);
  }
  return 0;
}
