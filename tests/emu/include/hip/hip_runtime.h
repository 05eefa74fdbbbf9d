// TEST INFRASTRUCTURE ONLY: stands in for <hip/hip_runtime.h> when the device sources are
// compiled with g++ for the CPU emulator (tests/emu/build.py puts this directory first on
// the include path).  Never on the product's include path.
#ifndef PQP_EMU_FAKE_HIP_RUNTIME_H
#define PQP_EMU_FAKE_HIP_RUNTIME_H
// address-space qualifiers and the inlining policy are meaningless on the host
#define PQP_LDS
#define PQP_GLOBAL
#define PQP_CALL inline
// scalarisation is the identity on the host
#define __builtin_amdgcn_readfirstlane(x) (x)
#define PQP_OPAQUE_SCALAR(v) ((void)0)
#define PQP_OPAQUE_VECTOR(v) ((void)0)
#include "../../hip_emu.hpp"
// FP64 matrix core (v_mfma_f64_16x16x4_f64) as a wave collective built from lane shuffles,
// with the hardware's operand layout (see pqp_block.hpp): a = A[l & 15][l >> 4],
// b = B[l >> 4][l & 15], result r of lane l = D[(l >> 4) + 4 r][l & 15].
#define PQP_EMULATED_MFMA
struct pqp_d4
{
  double v[4];
  double& operator[](int i) { return v[i]; }
  const double& operator[](int i) const { return v[i]; }
};
inline void
wave_sync()
{
  hipemu::wave_barrier();
}
inline double
lane_bcast(double v, int src)
{
  return __shfl(v, src);
}
inline pqp_d4
mfma_f64_16x16x4(double a, double b, pqp_d4 c)
{
  const int lane = int(threadIdx.x) & 63;
  const int col = lane & 15, rbase = lane >> 4;
  double bk[4];
  for (int k = 0; k < 4; ++k)
    bk[k] = __shfl(b, col + 16 * k);
  for (int r = 0; r < 4; ++r) {
    const int row = rbase + 4 * r;
    double acc = c[r];
    for (int k = 0; k < 4; ++k) {
      double aik = __shfl(a, row + 16 * k);
      acc = std::fma(aik, bk[k], acc);
    }
    c[r] = acc;
  }
  return c;
}
#endif
