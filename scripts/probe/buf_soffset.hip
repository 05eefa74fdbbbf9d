// Does raw-buffer range checking on gfx950 include soffset?  base fixed, per-row soffset, num_records = soffset + chi*8.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ void k(const double* p, int ld, int rows, double* out)
{
  const int lane = threadIdx.x;
  for (int r = 0; r < rows; ++r) {
    const int soff = r * ld * 8;
    const int chi = r + 1; // columns [0, r]
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, soff + chi * 8, 0x00020000);
    v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, soff, 0);
    d2 d = __builtin_bit_cast(d2, v);
    out[(long)r * 128 + 2 * lane] = d.x;
    out[(long)r * 128 + 2 * lane + 1] = d.y;
  }
}
int main()
{
  const int ld = 100, rows = 100;
  double* h = new double[ld * rows];
  for (int i = 0; i < ld * rows; ++i)
    h[i] = 1.0 + i;
  double *d, *o;
  hipMalloc(&d, ld * rows * 8);
  hipMalloc(&o, rows * 128 * 8);
  hipMemcpy(d, h, ld * rows * 8, hipMemcpyHostToDevice);
  hipMemset(o, 0xff, rows * 128 * 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, ld, rows, o);
  double* ho = new double[rows * 128];
  hipMemcpy(ho, o, rows * 128 * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < 128; ++c) {
      const double want = (c <= r) ? h[r * ld + c] : 0.0;
      if (ho[r * 128 + c] != want) {
        if (bad < 10)
          printf("row %d col %d: got %g want %g\n", r, c, ho[r * 128 + c], want);
        ++bad;
      }
    }
  printf("soffset-included range check: %s (%d mismatches)\n", bad ? "NO" : "YES", bad);
  return 0;
}
