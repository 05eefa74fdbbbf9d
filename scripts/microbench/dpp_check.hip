// On-device check of the DPP wavefront reductions of pqp_block.hpp against host sums (the SIMT
// emulator runs their shuffle twins, so the DPP forms are only ever exercised on the GPU).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I proxsuite_amd/csrc \
//         scripts/microbench/dpp_check.hip -o scripts/microbench/dpp_check && scripts/microbench/dpp_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "pqp_block.hpp"

__global__ void
reduce_kernel(const double* in, double* out)
{
  const double v = in[blockIdx.x * 64 + threadIdx.x];
  double* o = out + blockIdx.x * 4 * 64;
  o[threadIdx.x] = pqp::wave_sum(v);
  o[64 + threadIdx.x] = pqp::wave_max(v);
  o[128 + threadIdx.x] = pqp::wave_min(v);
  o[192 + threadIdx.x] = pqp::row16_sum(v);
}

int
main()
{
  const int W = 64;
  std::vector<double> in(W * 64), out(W * 4 * 64);
  unsigned long long s = 88172645463325252ull;
  for (auto& x : in) {
    s ^= s << 13, s ^= s >> 7, s ^= s << 17;
    x = double(long(s % 2000001) - 1000000) / 1000.0;
  }
  double *din, *dout;
  if (hipMalloc(&din, in.size() * 8) != hipSuccess || hipMalloc(&dout, out.size() * 8) != hipSuccess)
    return std::printf("no device\n"), 2;
  hipMemcpy(din, in.data(), in.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(reduce_kernel, dim3(W), dim3(64), 0, 0, din, dout);
  hipMemcpy(out.data(), dout, out.size() * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int w = 0; w < W; ++w) {
    const double* x = &in[w * 64];
    double sum = 0, mx = -INFINITY, mn = INFINITY;
    for (int l = 0; l < 64; ++l)
      sum += x[l], mx = std::fmax(mx, x[l]), mn = std::fmin(mn, x[l]);
    for (int l = 0; l < 64; ++l) {
      const double* o = &out[w * 256];
      if (std::fabs(o[l] - sum) > 1e-9 * (1 + std::fabs(sum)) || o[64 + l] != mx || o[128 + l] != mn)
        ++bad;
    }
    for (int r = 0; r < 4; ++r) {
      double rs = 0;
      for (int l = 0; l < 16; ++l)
        rs += x[r * 16 + l];
      if (std::fabs(out[w * 256 + 192 + r * 16 + 15] - rs) > 1e-9 * (1 + std::fabs(rs)))
        ++bad;
    }
  }
  std::printf("dpp_check: %s (%d mismatches)\n", bad ? "FAILED" : "ok", bad);
  return bad ? 1 : 0;
}
