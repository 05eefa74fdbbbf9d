// Workgroup-cooperative building blocks of the batched ProxQP kernels (gfx950).
//
// Execution model: ONE workgroup of NT threads (NT/64 wavefronts) owns ONE QP.
// Per-QP vectors live in LDS; per-QP matrices live in HBM as contiguous row-major
// panels and are always walked so that consecutive lanes read consecutive
// addresses of one row (coalesced 512-B wave transactions):
//   * every mat-vec is "thread-per-output, loop over the inner index"
//     (out[j] = sum_k M[k][j] v[k]); both orientations of A and C are stored
//     so that A*x, A^T*y, C*x, C^T*z all take this form and need no cross-lane
//     reduction;
//   * the inner index is split across the wavefronts of the workgroup and the
//     partial sums meet in LDS.
// Scalars that steer control flow are computed redundantly by every thread
// from LDS broadcasts, so all branches are workgroup-uniform.
#ifndef PQP_BLOCK_HPP
#define PQP_BLOCK_HPP

#include <hip/hip_runtime.h>

// Explicit address spaces: the shared routines below are real (non-inlined)
// functions so that the persistent solve kernel stays small enough for the
// instruction cache; typing their pointer parameters keeps LDS traffic on ds_*
// and HBM traffic on global_* instructions instead of flat_*.
#ifndef PQP_LDS
#define PQP_LDS __attribute__((address_space(3)))
#define PQP_GLOBAL __attribute__((address_space(1)))
#endif
// Inlining policy of the shared dense routines (see DESIGN.md "code size vs
// registers"): real calls keep the kernel small but force every uniform scalar
// of the caller into VGPRs across the call; inlining does the opposite.
#ifndef PQP_CALL
#define PQP_CALL __forceinline__
#endif

namespace pqp {

typedef PQP_LDS double* lptr;
typedef const PQP_LDS double* clptr;
typedef PQP_LDS int* liptr;
typedef const PQP_LDS int* cliptr;
typedef PQP_GLOBAL double* gptr;
typedef const PQP_GLOBAL double* cgptr;

constexpr int WAVE = 64;
constexpr int PQP_NB = 16; // panel width of the blocked LDL^T / substitutions

// ---------------------------------------------------------------------------
// Wave-uniform scalars.  Every thread of a workgroup computes the solver's control
// scalars (mu, rho, BCL thresholds, residual norms, counters ...) redundantly, so
// they are uniform by construction -- but fp64 arithmetic runs on the VALU and would
// leave ~50 of them in VGPRs of every lane for the whole kernel.  uni() moves a value
// to the scalar register file (v_readfirstlane), where the compiler keeps it in
// SGPRs / spills it to VGPR *lanes* instead of scratch memory, and where it makes
// branch conditions provably uniform (scalar branches instead of exec masking).
// ---------------------------------------------------------------------------
__device__ __forceinline__ int
uni(int v)
{
  return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ double
uni(double v)
{
  union
  {
    double d;
    int i[2];
  } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
  u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
  return u.d;
}
__device__ __forceinline__ long
uni(long v)
{
  union
  {
    long l;
    int i[2];
  } u;
  u.l = v;
  u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
  u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
  return u.l;
}
__device__ __forceinline__ bool
uni(bool v)
{
  return __builtin_amdgcn_readfirstlane(v ? 1 : 0) != 0;
}
// a double that is re-scalarised on every assignment
struct UD
{
  double v;
  __device__ __forceinline__ UD()
    : v(0)
  {
  }
  __device__ __forceinline__ UD(double x)
    : v(uni(x))
  {
  }
  __device__ __forceinline__ UD& operator=(double x)
  {
    v = uni(x);
    return *this;
  }
  __device__ __forceinline__ operator double() const { return v; }
  __device__ __forceinline__ UD& operator+=(double x) { return *this = v + x; }
  __device__ __forceinline__ UD& operator*=(double x) { return *this = v * x; }
};

// keeps a value (and the loads that produced it) alive without storing it
__device__ __forceinline__ void
keep_alive(double v)
{
#ifndef PQP_EMULATED_MFMA
  asm volatile("" ::"v"(v));
#else
  (void)v;
#endif
}

// Wavefront reductions.  On the device they are pure VALU: the AMDGPU backend's own scan pattern on
// DPP moves (row_shr 1/2/4/8 inside the 16-lane rows, row_bcast15 onto rows 1 and 3, row_bcast31
// onto rows 2 and 3) leaves the result in lane 63, which v_readlane hands to every lane.  The
// xor-butterfly on __shfl_xor they replace goes through the LDS crossbar (ds_bpermute): six
// dependent LDS round trips per value, on the pipe the solver's vectors live on.
#ifndef PQP_EMULATED_MFMA
template<int CTRL, int ROW_MASK>
__device__ __forceinline__ double
dpp_move(double old, double v)
{
  // lanes whose source lane does not exist (or whose row is masked out) keep `old`
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double
readlane_f64(double v, int src)
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
#define PQP_WAVE_REDUCE(OP, IDENT)                                                                \
  v = OP(v, dpp_move<0x111, 0xf>(IDENT, v)); /* row_shr:1 */                                       \
  v = OP(v, dpp_move<0x112, 0xf>(IDENT, v)); /* row_shr:2 */                                       \
  v = OP(v, dpp_move<0x114, 0xf>(IDENT, v)); /* row_shr:4 */                                       \
  v = OP(v, dpp_move<0x118, 0xf>(IDENT, v)); /* row_shr:8 */                                       \
  v = OP(v, dpp_move<0x142, 0xa>(IDENT, v)); /* row_bcast:15 -> rows 1, 3 */                       \
  v = OP(v, dpp_move<0x143, 0xc>(IDENT, v)); /* row_bcast:31 -> rows 2, 3 */                       \
  return readlane_f64(v, 63);
__device__ __forceinline__ double
pqp_add(double a, double b)
{
  return a + b;
}
__device__ __forceinline__ double
wave_sum(double v)
{
  PQP_WAVE_REDUCE(pqp_add, 0.0)
}
__device__ __forceinline__ double
wave_max(double v)
{
  PQP_WAVE_REDUCE(fmax, -__builtin_inf())
}
__device__ __forceinline__ double
wave_min(double v)
{
  PQP_WAVE_REDUCE(fmin, __builtin_inf())
}
#undef PQP_WAVE_REDUCE
// sum over each 16-lane row; the total is valid in the row's LAST lane (lane & 15 == 15)
__device__ __forceinline__ double
row16_sum(double v)
{
  v += dpp_move<0x111, 0xf>(0.0, v);
  v += dpp_move<0x112, 0xf>(0.0, v);
  v += dpp_move<0x114, 0xf>(0.0, v);
  v += dpp_move<0x118, 0xf>(0.0, v);
  return v;
}
#else
__device__ __forceinline__ double
wave_sum(double v)
{
#pragma unroll
  for (int o = WAVE / 2; o > 0; o >>= 1)
    v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double
wave_max(double v)
{
#pragma unroll
  for (int o = WAVE / 2; o > 0; o >>= 1)
    v = fmax(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double
wave_min(double v)
{
#pragma unroll
  for (int o = WAVE / 2; o > 0; o >>= 1)
    v = fmin(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double
row16_sum(double v)
{
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  return v; // every lane of the row, its last one included
}
#endif

// Block reductions with a parity-toggled LDS scratch: ONE barrier per reduction.
// `red` points at 2 * RED_VALS * (NT/64) doubles.  Every thread of the block must call
// these in the same order (the parity lives in a register).
constexpr int RED_VALS = 16; // values one fused reduction can carry
template<int NT>
struct Reducer
{
  static constexpr int NW = NT / WAVE;
  lptr red;
  int parity;
  __device__ __forceinline__ Reducer(lptr scratch)
    : red(scratch)
    , parity(0)
  {
  }
  __device__ __forceinline__ lptr slot() { return red + parity * RED_VALS * NW; }

  // NS sums and NM maxima in ONE barrier interval (a barrier interval of this kernel costs a few
  // thousand cycles under load, the extra wave reductions a few hundred)
  template<int NS, int NM>
  __device__ __forceinline__ void mixed(double (&sv)[NS > 0 ? NS : 1], double (&mv)[NM > 0 ? NM : 1])
  {
    static_assert(NS + NM <= RED_VALS, "too many values for one fused reduction");
#pragma unroll
    for (int i = 0; i < NS; ++i)
      sv[i] = wave_sum(sv[i]);
#pragma unroll
    for (int i = 0; i < NM; ++i)
      mv[i] = wave_max(mv[i]);
    lptr s = slot();
    if ((threadIdx.x & (WAVE - 1)) == 0) {
      const int w = threadIdx.x / WAVE;
#pragma unroll
      for (int i = 0; i < NS; ++i)
        s[i * NW + w] = sv[i];
#pragma unroll
      for (int i = 0; i < NM; ++i)
        s[(NS + i) * NW + w] = mv[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      double r = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w)
        r += s[i * NW + w];
      sv[i] = uni(r);
    }
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      double r = s[(NS + i) * NW];
#pragma unroll
      for (int w = 1; w < NW; ++w)
        r = fmax(r, s[(NS + i) * NW + w]);
      mv[i] = uni(r);
    }
    parity ^= 1;
  }

  __device__ __forceinline__ double sum(double v)
  {
    v = wave_sum(v);
    lptr s = slot();
    if ((threadIdx.x & (WAVE - 1)) == 0)
      s[threadIdx.x / WAVE] = v;
    __syncthreads();
    double r = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w)
      r += s[w];
    parity ^= 1;
    return uni(r);
  }
  __device__ __forceinline__ double max(double v)
  {
    v = wave_max(v);
    lptr s = slot();
    if ((threadIdx.x & (WAVE - 1)) == 0)
      s[threadIdx.x / WAVE] = v;
    __syncthreads();
    double r = s[0];
#pragma unroll
    for (int w = 1; w < NW; ++w)
      r = fmax(r, s[w]);
    parity ^= 1;
    return uni(r);
  }
  __device__ __forceinline__ double min(double v)
  {
    v = wave_min(v);
    lptr s = slot();
    if ((threadIdx.x & (WAVE - 1)) == 0)
      s[threadIdx.x / WAVE] = v;
    __syncthreads();
    double r = s[0];
#pragma unroll
    for (int w = 1; w < NW; ++w)
      r = fmin(r, s[w]);
    parity ^= 1;
    return uni(r);
  }
  // two sums in one barrier
  __device__ __forceinline__ void sum2(double& a, double& b)
  {
    a = wave_sum(a);
    b = wave_sum(b);
    lptr s = slot();
    if ((threadIdx.x & (WAVE - 1)) == 0) {
      s[threadIdx.x / WAVE] = a;
      s[NW + threadIdx.x / WAVE] = b;
    }
    __syncthreads();
    double ra = 0, rb = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      ra += s[w];
      rb += s[NW + w];
    }
    a = uni(ra);
    b = uni(rb);
    parity ^= 1;
  }
  // up to three maxima in one barrier
  __device__ __forceinline__ void max3(double& a, double& b, double& c)
  {
    a = wave_max(a);
    b = wave_max(b);
    c = wave_max(c);
    lptr s = slot();
    if ((threadIdx.x & (WAVE - 1)) == 0) {
      s[threadIdx.x / WAVE] = a;
      s[NW + threadIdx.x / WAVE] = b;
      s[2 * NW + threadIdx.x / WAVE] = c;
    }
    __syncthreads();
    double ra = s[0], rb = s[NW], rc = s[2 * NW];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      ra = fmax(ra, s[w]);
      rb = fmax(rb, s[NW + w]);
      rc = fmax(rc, s[2 * NW + w]);
    }
    a = uni(ra);
    b = uni(rb);
    c = uni(rc);
    parity ^= 1;
  }
  // four sums in one barrier
  __device__ __forceinline__ void sum4(double& a, double& b, double& c, double& d)
  {
    a = wave_sum(a);
    b = wave_sum(b);
    c = wave_sum(c);
    d = wave_sum(d);
    lptr s = slot();
    if ((threadIdx.x & (WAVE - 1)) == 0) {
      int w = threadIdx.x / WAVE;
      s[w] = a;
      s[NW + w] = b;
      s[2 * NW + w] = c;
      s[3 * NW + w] = d;
    }
    __syncthreads();
    double ra = 0, rb = 0, rc = 0, rd = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      ra += s[w];
      rb += s[NW + w];
      rc += s[2 * NW + w];
      rd += s[3 * NW + w];
    }
    a = uni(ra);
    b = uni(rb);
    c = uni(rc);
    d = uni(rd);
    parity ^= 1;
  }
};

// infinity norm of an LDS vector
template<int NT>
__device__ __forceinline__ double
block_inf_norm(Reducer<NT>& R, clptr v, int n)
{
  double m = 0;
  for (int i = threadIdx.x; i < n; i += NT)
    m = fmax(m, fabs(v[i]));
  return R.max(m);
}

// ---------------------------------------------------------------------------
// gemv:  out[j] = sum_{k<K} M[row(k)*ld + col(j)] * v[k]     for j < J
// M in HBM (row-major, row contiguous), v / out / part in LDS.  The J outputs are
// tiled over 64-lane column chunks, the K range is split across the wavefronts that
// share a chunk and the partial sums meet in `part` (gemv_part_len() doubles).
// Optional gathers pick an active subset of rows / columns without copies:
//   index(i) = i                         when i <  split
//            = split + map[i - split]    when i >= split      (map in LDS)
// pass map == nullptr for the identity.  Two barriers inside (after the partial
// sums, after `out` is written); `out` may alias `v`.
// ---------------------------------------------------------------------------
__host__ __device__ inline int
gemv_part_len(int nt, int jmax)
{
  return (nt > jmax ? nt : jmax) + WAVE;
}

template<int NT>
__device__ PQP_CALL void
gemv(cgptr M, int ld, int K, int J, clptr v, lptr out, lptr part, cliptr rowmap, int rowsplit,
     cliptr colmap, int colsplit, int tri = 0, lptr out2 = nullptr, clptr div = nullptr)
{
  // out2 / div (optional epilogue, compile-time constant at every call site): out2[j] = out[j] / div[j]
  // in the same pass that writes out[j] -- saves the caller a loop and a barrier interval
  // tri = +1: M[k][j] == 0 for k > j (only k <= j is read);  tri = -1: M[k][j] == 0 for k < j.
  // Honoured on the plain (no row gather) k-split path; the skipped terms are exact zeros.
  constexpr int NW = NT / WAVE;
  const int lane = threadIdx.x & (WAVE - 1);
  const int wid = threadIdx.x / WAVE;
  const int JC = (J + WAVE - 1) / WAVE; // column chunks
  int KS = 1;                           // k-splits per chunk
  if (JC > 0 && JC <= NW) {
    KS = NW / JC;
    const int chunk = wid % JC;
    const int ks = wid / JC;
    if (ks < KS) {
      const int j = chunk * WAVE + lane;
      if (j < J) {
        int cj = j;
        if (colmap && j >= colsplit)
          cj = colsplit + colmap[j - colsplit];
        cgptr col = M + cj;
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        int k = ks;
        if (rowmap) {
          // gathered rows: resolve the row indices, then 8 / 4 independent loads
          for (; k + 7 * KS < K; k += 8 * KS) {
            double m[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              int ku = k + u * KS;
              int ru = (ku < rowsplit) ? ku : rowsplit + rowmap[ku - rowsplit];
              m[u] = col[(long)ru * ld];
            }
#pragma unroll
            for (int u = 0; u < 8; u += 4) {
              a0 = fma(m[u], v[k + u * KS], a0);
              a1 = fma(m[u + 1], v[k + (u + 1) * KS], a1);
              a2 = fma(m[u + 2], v[k + (u + 2) * KS], a2);
              a3 = fma(m[u + 3], v[k + (u + 3) * KS], a3);
            }
          }
          if (k < K) {
            // at most 8 gathered rows are left: one batch of clamped, masked loads
            double m[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int ku = (k + u * KS < K) ? (k + u * KS) : k;
              const int ru = (ku < rowsplit) ? ku : rowsplit + rowmap[ku - rowsplit];
              m[u] = col[(long)ru * ld];
            }
#pragma unroll
            for (int u = 0; u < 8; u += 4) {
              a0 = fma(m[u], (k + u * KS < K) ? v[k + u * KS] : 0.0, a0);
              a1 = fma(m[u + 1], (k + (u + 1) * KS < K) ? v[k + (u + 1) * KS] : 0.0, a1);
              a2 = fma(m[u + 2], (k + (u + 2) * KS < K) ? v[k + (u + 2) * KS] : 0.0, a2);
              a3 = fma(m[u + 3], (k + (u + 3) * KS < K) ? v[k + (u + 3) * KS] : 0.0, a3);
            }
          }
        } else {
          const long step = (long)KS * ld;
          int Kj = K; // per-lane row range [k, Kj)
          if (tri > 0)
            Kj = (cj + 1 < K) ? (cj + 1) : K;
          if (tri < 0 && cj > k)
            k += ((cj - k + KS - 1) / KS) * KS;
          cgptr p = col + (long)k * ld;
          // 16, then 8, independent loads issued back to back before their first use: the
          // kernel is bound by HBM round trips, so bytes in flight per lane is the lever
#ifndef PQP_GEMV_DEEP_256
#define PQP_GEMV_DEEP_256 1 // (0 in the translation unit of the 128-VGPR C2 kernel: 8 loads in flight)
#endif
          // 16 loads in flight per lane where the register budget allows
          constexpr bool DEEP = (NT == 256) ? (PQP_GEMV_DEEP_256 != 0) : true;
          for (; DEEP && k + 15 * KS < Kj; k += 16 * KS) {
            double m[16];
#pragma unroll
            for (int u = 0; u < 16; ++u)
              m[u] = p[u * step];
#pragma unroll
            for (int u = 0; u < 16; u += 4) {
              a0 = fma(m[u], v[k + u * KS], a0);
              a1 = fma(m[u + 1], v[k + (u + 1) * KS], a1);
              a2 = fma(m[u + 2], v[k + (u + 2) * KS], a2);
              a3 = fma(m[u + 3], v[k + (u + 3) * KS], a3);
            }
            p += 16 * step;
          }
          for (; k + 7 * KS < Kj; k += 8 * KS) {
            double m0 = p[0], m1 = p[step], m2 = p[2 * step], m3 = p[3 * step];
            double m4 = p[4 * step], m5 = p[5 * step], m6 = p[6 * step], m7 = p[7 * step];
            a0 = fma(m0, v[k], a0);
            a1 = fma(m1, v[k + KS], a1);
            a2 = fma(m2, v[k + 2 * KS], a2);
            a3 = fma(m3, v[k + 3 * KS], a3);
            a0 = fma(m4, v[k + 4 * KS], a0);
            a1 = fma(m5, v[k + 5 * KS], a1);
            a2 = fma(m6, v[k + 6 * KS], a2);
            a3 = fma(m7, v[k + 7 * KS], a3);
            p += 8 * step;
          }
          if (k < Kj) {
            // at most 7 rows are left: ONE batch of clamped, masked loads (a row-at-a-time tail
            // would cost a memory round trip per row)
            double m[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
              m[u] = p[((k + u * KS < Kj) ? u : 0) * step];
#pragma unroll
            for (int u = 0; u < 8; u += 4) {
              a0 = fma(m[u], (k + u * KS < Kj) ? v[k + u * KS] : 0.0, a0);
              a1 = fma(m[u + 1], (k + (u + 1) * KS < Kj) ? v[k + (u + 1) * KS] : 0.0, a1);
              a2 = fma(m[u + 2], (k + (u + 2) * KS < Kj) ? v[k + (u + 2) * KS] : 0.0, a2);
              a3 = fma(m[u + 3], (k + (u + 3) * KS < Kj) ? v[k + (u + 3) * KS] : 0.0, a3);
            }
          }
        }
        part[ks * J + j] = (a0 + a1) + (a2 + a3);
      }
    }
  } else {
    // more chunks than waves: each wave loops over its chunks, no k-split
    for (int chunk = wid; chunk < JC; chunk += NW) {
      const int j = chunk * WAVE + lane;
      if (j < J) {
        int cj = j;
        if (colmap && j >= colsplit)
          cj = colsplit + colmap[j - colsplit];
        cgptr col = M + cj;
        double a0 = 0, a1 = 0;
        for (int k = 0; k < K; ++k) {
          int rk = k;
          if (rowmap && k >= rowsplit)
            rk = rowsplit + rowmap[k - rowsplit];
          if (k & 1)
            a1 = fma(col[(long)rk * ld], v[k], a1);
          else
            a0 = fma(col[(long)rk * ld], v[k], a0);
        }
        part[j] = a0 + a1;
      }
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < J; j += NT) {
    double s = part[j];
    for (int q = 1; q < KS; ++q)
      s += part[q * J + j];
    out[j] = s;
    if (out2)
      out2[j] = s / div[j];
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------
// gemv_dual: BOTH products of a row-major matrix M (R rows x n columns) in ONE pass over it:
//   rowout[r] = sum_j M[r][j] * v[j]      r < R
//   colout[j] = sum_r w[r] * M[r][j]      j < n          (COLS = false: skipped, w / colout unused)
// This is what the KKT residual needs of A and C (A dx with A^T dy, C dx with C^T dz,
// reference solver.hpp:243-318); a pair of gemv calls reads the matrix and a transposed copy.
// Lane layout: 16 lanes share a row (one 128- or 256-byte segment per load), a wavefront covers four
// rows per step, each lane keeps the 8 column accumulators of its 16-column stripes; columns
// beyond 128 are handled in further blocks of 128.  Row sums close with four DPP row shifts
// inside the 16-lane group, column sums with two across the groups and an LDS pass across the
// wavefronts, block of 128 columns by block.  `part`: gemv_dual_part_len() doubles of LDS.  v, w must not alias
// the outputs.  Two barriers per column block with COLS, one in all otherwise.  (The column pass is a template flag and not a null
// test of w: LDS offset 0 is a valid address -- the first vector of the carve-up lives there.)
// ---------------------------------------------------------------------------
__host__ __device__ inline int
gemv_dual_part_len(int nt, int n)
{
  return (nt / WAVE) * (n < 128 ? n : 128);
}

// GATHER = true: row r of the product is row  r < rowsplit ? r : rowsplit + rowmap[r - rowsplit]
// of M (the gemv convention; rowmap in LDS).
// two adjacent doubles in one 16-byte load (address 16-byte aligned)
struct Pair
{
  double x, y;
};
__device__ __forceinline__ Pair
load_pair(cgptr p)
{
#ifndef PQP_EMULATED_MFMA
  typedef double pqp_d2 __attribute__((ext_vector_type(2)));
  const pqp_d2 t = *reinterpret_cast<const PQP_GLOBAL pqp_d2*>(p);
  return Pair{ t.x, t.y };
#else
  return Pair{ p[0], p[1] };
#endif
}

// W = doubles per lane and load: 2 (16-byte loads, 32-column stripes; needs even ld / n and a
// 16-byte aligned M) or 1.
// Epilogues (compile-time constants at the call sites; they fold an element-wise loop and its barrier
// interval of the caller into the pass):
//   EPI_ROW_RSUB : rowout[r] = (row sum) - rowout[r]           (in place)
//   EPI_ROW_DIV  : rowout[r] = (row sum) / ea[r]
//   EPI_COL_SUBDIV: colout[j] = (ea[j] - (column sum)) / eb[j]
enum
{
  EPI_NONE = 0,
  EPI_ROW_RSUB = 1,
  EPI_ROW_DIV = 2,
  EPI_COL_SUBDIV = 3
};
// TRIL = true: M is LOWER triangular with explicit zeros above the diagonal (the inverse factors W_S / W_P): a
// 16 W-column stripe of a row step is only loaded when some row of the wavefront's step reaches it (wave-uniform
// test, as in symv_lower); the skipped elements are exact zeros, so every sum keeps its bits.
template<int NT, bool COLS, bool GATHER, bool ROWS, int W, bool TRIL = false>
__device__ __forceinline__ void
gemv_dual_impl(cgptr M, int ld, int R, int n, clptr v, clptr w, lptr rowout, lptr colout, lptr part,
               cliptr rowmap, int rowsplit, int epi, clptr ea, clptr eb)
{
  constexpr int NW = NT / WAVE;
  constexpr int CH = 8 / W; // stripes of 16 * W columns per lane and column block (128 columns)
  const int lane = threadIdx.x & (WAVE - 1);
  const int wid = threadIdx.x / WAVE;
  const int g = lane >> 4, s = lane & 15;
  for (int c0 = 0; c0 < n; c0 += 128) {
    double vv[CH][W], acc[CH][W];
    int off[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int col = c0 + 16 * W * c + W * s;
      off[c] = (col < n) ? col : (n - W); // clamped: the load stays unconditional
#pragma unroll
      for (int e = 0; e < W; ++e) {
        vv[c][e] = (ROWS && col + e < n) ? v[off[c] + e] : 0.0;
        acc[c][e] = 0.0;
      }
    }
    // U row steps per trip: the loads of all of them are issued before the first use
#ifndef PQP_DUAL_U
#define PQP_DUAL_U 2
#endif
    constexpr int U = PQP_DUAL_U;
    for (int base = 0; base < R; base += U * 4 * NW) {
      int r[U];
      bool valid[U];
      double m[U][CH][W];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        r[u] = base + u * 4 * NW + 4 * wid + g;
        valid[u] = r[u] < R;
        int rsrc = valid[u] ? r[u] : (R - 1);
        if (GATHER && rsrc >= rowsplit)
          rsrc = rowsplit + rowmap[rsrc - rowsplit];
        cgptr row = M + (long)rsrc * ld;
        const int rmax = base + u * 4 * NW + 4 * wid + 3; // last row of this wavefront's step (wave-uniform)
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          if (c0 + 16 * W * c < n && (!TRIL || c0 + 16 * W * c <= rmax)) { // stripe test is wave-uniform
            if (W == 2) {
              const Pair t = load_pair(row + off[c]);
              m[u][c][0] = t.x;
              m[u][c][W - 1] = t.y;
            } else {
              m[u][c][0] = row[off[c]];
            }
          } else {
#pragma unroll
            for (int e = 0; e < W; ++e)
              m[u][c][e] = 0.0;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (ROWS) {
          double p0 = 0, p1 = 0;
#pragma unroll
          for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int e = 0; e < W; ++e) {
              if ((c * W + e) & 1)
                p1 = fma(m[u][c][e], vv[c][e], p1);
              else
                p0 = fma(m[u][c][e], vv[c][e], p0);
            }
          const double pr = row16_sum(p0 + p1);
          if (valid[u] && s == 15) {
            double o = (c0 == 0) ? ((epi == EPI_ROW_RSUB) ? pr - rowout[r[u]] : pr) : rowout[r[u]] + pr;
            if (epi == EPI_ROW_DIV && c0 + 128 >= n)
              o /= ea[r[u]];
            rowout[r[u]] = o;
          }
        }
        if (COLS) {
          const double wr = valid[u] ? w[valid[u] ? r[u] : 0] : 0.0;
#pragma unroll
          for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int e = 0; e < W; ++e)
              acc[c][e] = fma(wr, m[u][c][e], acc[c][e]);
        }
      }
    }
    if (COLS) {
      // the column sums of this block meet across the wavefronts (the scratch is reused by the next block)
      const int bw = (n - c0 < 128) ? (n - c0) : 128;
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int e = 0; e < W; ++e) {
          double a = acc[c][e];
          a += __shfl_xor(a, 16);
          a += __shfl_xor(a, 32);
          const int cb = 16 * W * c + W * s + e;
          if (g == 0 && cb < bw)
            part[wid * bw + cb] = a;
        }
      __syncthreads();
      for (int j = threadIdx.x; j < bw; j += NT) {
        double a = part[j];
#pragma unroll
        for (int q = 1; q < NW; ++q)
          a += part[q * bw + j];
        colout[c0 + j] = (epi == EPI_COL_SUBDIV) ? (ea[c0 + j] - a) / eb[c0 + j] : a;
      }
      if (c0 + 128 < n)
        __syncthreads();
    }
  }
  __syncthreads();
}

// ROWS = false: column sums only (v / rowout unused).
template<int NT, bool COLS = true, bool GATHER = false, bool ROWS = true, bool TRIL = false>
__device__ PQP_CALL void
gemv_dual(cgptr M, int ld, int R, int n, clptr v, clptr w, lptr rowout, lptr colout, lptr part,
          cliptr rowmap = nullptr, int rowsplit = 0, int epi = EPI_NONE, clptr ea = nullptr, clptr eb = nullptr)
{
  static_assert(!(TRIL && GATHER), "the triangular skip assumes rows in storage order");
  const bool wide = (((ld | n) & 1) == 0) && ((reinterpret_cast<unsigned long long>(M) & 15ull) == 0);
  if (wide)
    gemv_dual_impl<NT, COLS, GATHER, ROWS, 2, TRIL>(M, ld, R, n, v, w, rowout, colout, part, rowmap, rowsplit, epi, ea, eb);
  else
    gemv_dual_impl<NT, COLS, GATHER, ROWS, 1, TRIL>(M, ld, R, n, v, w, rowout, colout, part, rowmap, rowsplit, epi, ea, eb);
}

// ---------------------------------------------------------------------------
// FP64 matrix core: D(16x16) += A(16x4) * B(4x16), one wavefront, v_mfma_f64_16x16x4_f64.
// Operand layout (lane l of 64): a = A[l & 15][l >> 4], b = B[l >> 4][l & 15];
// result register r (0..3) of lane l = D[(l >> 4) + 4 r][l & 15].
// The emulator build (tests/emu) supplies the same function from lane shuffles.
// ---------------------------------------------------------------------------
// an optimisation barrier on a wave-uniform integer held in scalar registers
#ifndef PQP_OPAQUE_SCALAR
#define PQP_OPAQUE_SCALAR(v) asm volatile("" : "+s"(v))
#define PQP_OPAQUE_VECTOR(v) asm volatile("" : "+v"(v))
#endif

#ifndef PQP_EMULATED_MFMA
typedef double pqp_d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ pqp_d4
mfma_f64_16x16x4(double a, double b, pqp_d4 c)
{
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
#endif

// ---------------------------------------------------------------------------
// Register-resident LDL^T for m <= 16*MB, NT = 256 threads as a 16 x 16 grid.
//
// An HBM-resident factorisation pays a memory round trip per column chunk; at sizes
// of m ~ 50..110 that is ~100 dependent memory latencies.  Here the lower
// triangle lives in VGPRs for the whole factorisation: thread (ti, tj) owns
// the element (16*bi + ti, 16*bj + tj) of every 16 x 16 block bi >= bj
// (block-cyclic, MB*(MB+1)/2 doubles per thread), `load(i, j)` is called once
// per element up front (all loads in flight together), and each of the m
// right-looking column steps costs one barrier: the 16 threads that own
// column k publish it through LDS, everybody applies the rank-1 update
// a_ij -= a_ik a_jk / d_k to its own registers.  The block-column index kb is
// unrolled so every register index is static.
// Output: the upper mirror U[j][i] = l_ij (i > j) and U[j][j] = d_j in global
// memory, d[] in LDS.
// `cbuf`: 4 * 16 * MB doubles of LDS (raw + scaled column, double-buffered).
// ---------------------------------------------------------------------------
// value of `v` in lane `src` (uniform) of the calling wavefront, through scalar registers
#ifndef PQP_EMULATED_MFMA
__device__ __forceinline__ double
lane_bcast(double v, int src)
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
#endif

template<int NT, int MB, typename LoadFn>
__device__ PQP_CALL void
ldlt_factor_reg(LoadFn load, gptr U, int ld, int m, lptr d, lptr cbuf, PQP_LDS long long* prof = nullptr)
{
  long long t0 = 0;
#define PQP_PROF(slot)                                                                            \
  if (prof && threadIdx.x == 0) {                                                                  \
    long long t1 = clock64();                                                                      \
    prof[slot] += t1 - t0;                                                                         \
    t0 = t1;                                                                                       \
  }
  if (prof && threadIdx.x == 0)
    t0 = clock64();
  static_assert(NT == 256, "ldlt_factor_reg lays the workgroup out as 16 x 16 threads");
  constexpr int NB = 16;
  const int ti = threadIdx.x & (NB - 1), tj = threadIdx.x / NB;
  const int mb = (m + NB - 1) / NB;
  double a[MB * (MB + 1) / 2];
  // every load is issued unconditionally on clamped indices and masked afterwards: a
  // conditional load becomes a branch whose join waits for the data, which would turn the
  // gather into MB*(MB+1)/2 dependent round trips
#pragma unroll
  for (int bi = 0; bi < MB; ++bi)
#pragma unroll
    for (int bj = 0; bj <= bi; ++bj) {
      const int i = bi * NB + ti, j = bj * NB + tj;
      const int ic = (i < m) ? i : m - 1;
      const int jc = (j < ic) ? j : ic;
      a[bi * (bi + 1) / 2 + bj] = load(ic, jc);
    }
#pragma unroll
  for (int bi = 0; bi < MB; ++bi)
#pragma unroll
    for (int bj = 0; bj <= bi; ++bj) {
      const int i = bi * NB + ti, j = bj * NB + tj;
      if (!(i < m && j <= i))
        a[bi * (bi + 1) / 2 + bj] = 0.0;
    }
  int par = 0;
  if (prof) {
    // drain the loads so that the gather is billed to slot 0
    double sink = 0;
#pragma unroll
    for (int q = 0; q < MB * (MB + 1) / 2; ++q)
      sink += a[q];
    if (sink == 1.2345e300)
      d[0] = sink;
  }
  PQP_PROF(0)
#pragma unroll
  for (int kb = 0; kb < MB; ++kb) {
    if (kb < mb) {
      const int nk = (m - kb * NB < NB) ? (m - kb * NB) : NB;
      for (int kk = 0; kk < nk; ++kk) {
        const int k = kb * NB + kk;
        lptr cb = cbuf + par * (2 * MB * NB);
        // the pivot d_k sits in lane 16 (kk & 3) + kk of the wave that owns column k: a scalar
        // read (every wave reads its own lane; only the owner wave uses the value)
        const double dkw = lane_bcast(a[kb * (kb + 1) / 2 + kb], ((kk & 3) << 4) + kk);
        if (tj == kk) {
          // owners publish the raw column (a_ik) and the scaled one (l_ik = a_ik / d_k): the
          // other threads then need neither the division nor a multiply per element
          const double inv = 1.0 / dkw;
#pragma unroll
          for (int bi = kb; bi < MB; ++bi)
            if (bi < mb) {
              const double raw = a[bi * (bi + 1) / 2 + kb];
              const double scaled = raw * inv;
              cb[bi * NB + ti] = raw;
              cb[MB * NB + bi * NB + ti] = scaled;
              if (bi * NB + ti > k)
                a[bi * (bi + 1) / 2 + kb] = scaled;
            }
          if (ti == kk)
            d[k] = dkw;
        }
        __syncthreads();
        double ci[MB], cj[MB];
#pragma unroll
        for (int b = kb; b < MB; ++b) {
          ci[b] = (b < mb) ? cb[b * NB + ti] : 0.0;
          cj[b] = (b < mb) ? cb[MB * NB + b * NB + tj] : 0.0;
        }
#pragma unroll
        for (int bi = kb; bi < MB; ++bi)
          if (bi < mb) { // uniform: block rows past the matrix are skipped
#pragma unroll
            for (int bj = kb + 1; bj <= bi; ++bj)
              a[bi * (bi + 1) / 2 + bj] = fma(-ci[bi], cj[bj], a[bi * (bi + 1) / 2 + bj]);
          }
        if (tj > kk) { // the rest of block column kb
#pragma unroll
          for (int bi = kb; bi < MB; ++bi)
            if (bi < mb)
              a[bi * (bi + 1) / 2 + kb] = fma(-ci[bi], cj[kb], a[bi * (bi + 1) / 2 + kb]);
        }
        par ^= 1;
      }
    }
  }
  PQP_PROF(1)
#pragma unroll
  for (int bi = 0; bi < MB; ++bi)
#pragma unroll
    for (int bj = 0; bj <= bi; ++bj) {
      const int i = bi * NB + ti, j = bj * NB + tj;
      if (i < m && j <= i)
        U[(long)j * ld + i] = a[bi * (bi + 1) / 2 + bj];
    }
  __syncthreads();
  PQP_PROF(3)
#undef PQP_PROF
}

// ---------------------------------------------------------------------------
// Register-resident LDL^T that hands back the INVERSE factor: on exit
//     Wl[i][j] = (L^{-1})_ij  (j < i; unit diagonal written, strict upper never touched), d[] in LDS,
// so that  S^{-1} = W^T D^{-1} W  is applied with two chain-free mat-vecs and edited in place by
// the rank-1 routines of the solver (append a row, delete a row).
// Same right-looking elimination as ldlt_factor_reg, run Gauss-Jordan style on [S | I]: the
// elimination of column k, a_ij -= a_ik l_jk, is also applied to the identity block,
//     E_ic -= l_ik E_kc   (i > k, c <= k),
// which leaves E = L^{-1}.  Row k of E is final at step k and rides through LDS beside column k
// of A under the SAME barrier, so the inverse costs no extra synchronisation.  E uses the
// transposed thread mapping -- thread (ti, tj) owns E[16 bi + tj][16 bj + ti] -- so that its
// multipliers l_ik are the `cj` values the A update already holds, and the final stores are
// coalesced along a row of W.  Liveness: block (bi, bj) of E is born at block step bj and
// stored at block step bi; block column kb of A dies after block step kb (L itself is not kept).
// `cbuf`: 6 * 16 * MB doubles of LDS (raw column, scaled column, E row; double-buffered).
// ---------------------------------------------------------------------------
template<int NT, int MB, typename LoadFn>
__device__ PQP_CALL void
ldlt_inverse_reg(LoadFn load, gptr Wl, int ld, int m, lptr d, lptr cbuf)
{
  static_assert(NT == 256, "ldlt_inverse_reg lays the workgroup out as 16 x 16 threads");
  constexpr int NB = 16;
  const int ti = threadIdx.x & (NB - 1), tj = threadIdx.x / NB;
  const int mb = (m + NB - 1) / NB;
  double a[MB * (MB + 1) / 2];
  double e[MB * (MB + 1) / 2];
#pragma unroll
  for (int bi = 0; bi < MB; ++bi)
#pragma unroll
    for (int bj = 0; bj <= bi; ++bj) {
      const int i = bi * NB + ti, j = bj * NB + tj;
      const int ic = (i < m) ? i : m - 1;
      const int jc = (j < ic) ? j : ic;
      a[bi * (bi + 1) / 2 + bj] = load(ic, jc);
      e[bi * (bi + 1) / 2 + bj] = 0.0;
    }
#pragma unroll
  for (int bi = 0; bi < MB; ++bi)
#pragma unroll
    for (int bj = 0; bj <= bi; ++bj) {
      const int i = bi * NB + ti, j = bj * NB + tj;
      if (!(i < m && j <= i))
        a[bi * (bi + 1) / 2 + bj] = 0.0;
    }
  int par = 0;
#pragma unroll
  for (int kb = 0; kb < MB; ++kb) {
    if (kb < mb) {
      const int nk = (m - kb * NB < NB) ? (m - kb * NB) : NB;
      for (int kk = 0; kk < nk; ++kk) {
        const int k = kb * NB + kk;
        lptr cb = cbuf + par * (3 * MB * NB);
        const double dkw = lane_bcast(a[kb * (kb + 1) / 2 + kb], ((kk & 3) << 4) + kk);
        if (tj == kk) {
          const double inv = 1.0 / dkw;
#pragma unroll
          for (int bi = kb; bi < MB; ++bi)
            if (bi < mb) {
              const double raw = a[bi * (bi + 1) / 2 + kb];
              cb[bi * NB + ti] = raw;
              cb[MB * NB + bi * NB + ti] = raw * inv;
            }
          // row k of E (thread (ti, tj = kk) owns E[16 kb + kk][16 bj + ti])
#pragma unroll
          for (int bj = 0; bj <= kb; ++bj) {
            const int c = bj * NB + ti;
            cb[2 * MB * NB + bj * NB + ti] = (c < k) ? e[kb * (kb + 1) / 2 + bj] : ((c == k) ? 1.0 : 0.0);
          }
          if (ti == kk)
            d[k] = dkw;
        }
        __syncthreads();
        double ci[MB], cj[MB], er[MB];
#pragma unroll
        for (int b = kb; b < MB; ++b) {
          ci[b] = (b < mb) ? cb[b * NB + ti] : 0.0;
          cj[b] = (b < mb) ? cb[MB * NB + b * NB + tj] : 0.0;
        }
#pragma unroll
        for (int b = 0; b <= kb; ++b)
          er[b] = cb[2 * MB * NB + b * NB + ti];
#pragma unroll
        for (int bi = kb; bi < MB; ++bi)
          if (bi < mb) { // uniform: block rows past the matrix are skipped
#pragma unroll
            for (int bj = kb + 1; bj <= bi; ++bj)
              a[bi * (bi + 1) / 2 + bj] = fma(-ci[bi], cj[bj], a[bi * (bi + 1) / 2 + bj]);
            // E[i][c] -= l_ik E[k][c] for the rows i = 16 bi + tj > k of this thread
            const double lik = (bi * NB + tj > k) ? cj[bi] : 0.0;
#pragma unroll
            for (int bj = 0; bj <= kb; ++bj)
              e[bi * (bi + 1) / 2 + bj] = fma(-lik, er[bj], e[bi * (bi + 1) / 2 + bj]);
          }
        if (tj > kk) { // the rest of block column kb of A
#pragma unroll
          for (int bi = kb; bi < MB; ++bi)
            if (bi < mb)
              a[bi * (bi + 1) / 2 + kb] = fma(-ci[bi], cj[kb], a[bi * (bi + 1) / 2 + kb]);
        }
        par ^= 1;
      }
      // block row kb of E is final: rows 16 kb + tj, columns 16 bj + ti (coalesced along ti)
      const int row = kb * NB + tj;
      if (row < m) {
#pragma unroll
        for (int bj = 0; bj <= kb; ++bj) {
          const int c = bj * NB + ti;
          if (c < row)
            Wl[(long)row * ld + c] = e[kb * (kb + 1) / 2 + bj];
          else if (c == row)
            Wl[(long)row * ld + c] = 1.0;
        }
      }
    }
  }
  __syncthreads();
}

// inclusive prefix sum of one double per thread over the workgroup (threads past `count`
// contribute 0).  `scratch`: NT / 64 doubles of LDS, free again after the call.  Two barriers.
template<int NT>
__device__ __forceinline__ double
block_scan_inclusive(double v, lptr scratch)
{
  constexpr int NW = NT / WAVE;
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
#pragma unroll
  for (int o = 1; o < WAVE; o <<= 1) {
    const double up = __shfl_up(v, o);
    if (lane >= o)
      v += up;
  }
  if (lane == WAVE - 1)
    scratch[wid] = v;
  __syncthreads();
  double off = 0.0;
#pragma unroll
  for (int w = 0; w < NW; ++w)
    if (w < wid)
      off += scratch[w];
  __syncthreads();
  return v + off;
}

// (the explicit inverse W = L^{-1} is computed on the matrix cores: tri_inverse_mfma /
// tri_inverse_mfma_rows below)

// ---------------------------------------------------------------------------
// Inverses of the unit-lower diagonal blocks L_bb of an upper-mirror factor (the output of
// ldlt_factor_reg), written back into the diagonal blocks in the FULL layout that tri_inverse
// expects: strict lower = inv(L_bb), strict upper = inv(L_bb)^T, diagonal (d_j) untouched.
// L_bb = I + N with N strictly lower, N^16 = 0, so
//     inv(L_bb) = (I - N)(I + N^2)(I + N^4)(I + N^8)
// exactly: six 16x16 products on the matrix cores per block, no substitution chain.  Every
// matrix is carried as the pair (X, X^T) of tiles in the MFMA result layout, because a tile used
// as the A operand stands for its transpose and as the B operand for itself.
// One block per wavefront at a time.
// ---------------------------------------------------------------------------
struct TilePair
{
  pqp_d4 x, xt;
};
__device__ __forceinline__ TilePair
tile_mul(const TilePair& a, const TilePair& b)
{
  // c = a * b : A operand = (tile a.xt), B operand = (tile b.x);  c^T = b^T a^T
  TilePair c;
  c.x = pqp_d4{ 0.0, 0.0, 0.0, 0.0 };
  c.xt = pqp_d4{ 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    c.x = mfma_f64_16x16x4(a.xt[q], b.x[q], c.x);
    c.xt = mfma_f64_16x16x4(b.x[q], a.xt[q], c.xt);
  }
  return c;
}
template<int NT>
__device__ PQP_CALL void
diag_block_inverses_mfma(gptr F, int ld, int n)
{
  constexpr int NB = 16;
  constexpr int NWV = NT / WAVE;
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  const int lr = lane & 15, lk = lane >> 4;
  const int nbk = (n + NB - 1) / NB;
  for (int j = w; j < nbk; j += NWV) {
    const int j0 = j * NB;
    TilePair N, P;
    bool diag[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = lk + 4 * q, col = lr;
      const int gr = j0 + row, gc = j0 + col;
      const int grc = (gr < n) ? gr : n - 1, gcc = (gc < n) ? gc : n - 1;
      // upper mirror: U[c][r] = L[r][c] for r > c
      const double up = F[(long)grc * ld + gcc];  // (row, col), meaningful for row < col: L[col][row]
      const double lo = F[(long)gcc * ld + grc];  // (col, row): for row > col this is U[col][row] = L[row][col]
      const bool in = gr < n && gc < n;
      N.x[q] = (row > col && in) ? lo : 0.0;  // N[row][col]
      N.xt[q] = (row < col && in) ? up : 0.0; // N^T[row][col] = N[col][row]
      diag[q] = (row == col);
      P.x[q] = (diag[q] ? 1.0 : 0.0) - N.x[q]; // I - N
      P.xt[q] = (diag[q] ? 1.0 : 0.0) - N.xt[q];
    }
    TilePair S = tile_mul(N, N); // N^2
#pragma unroll
    for (int rep = 0; rep < 3; ++rep) {
      TilePair T = S; // I + N^(2,4,8)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (diag[q]) {
          T.x[q] += 1.0;
          T.xt[q] += 1.0;
        }
      P = tile_mul(P, T);
      if (rep < 2)
        S = tile_mul(S, S);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = lk + 4 * q, col = lr;
      const int gr = j0 + row, gc = j0 + col;
      if (gr < n && gc < n && row != col)
        F[(long)gr * ld + gc] = (row > col) ? P.x[q] : P.xt[q];
    }
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------
// tri_inverse on the FP64 matrix cores, n <= 16 * MB.  Input: the FULL layout of the
// factorisation routines (L in both mirrors, inv(L_bb) / inv(L_bb)^T in the diagonal blocks, d_j on
// the diagonal); output: WL = L^{-1} (row-major, lower) and WU = WL^T.
// Block forward substitution, one block COLUMN j of W per wavefront:
//     W_jj = inv(L_jj),    W_ij = -inv(L_ii) * sum_{k=j}^{i-1} L_ik W_kj     (i > j)
// Every product is a 16x16x16 tile product = 4 MFMAs.  The W_kj tiles of the column stay in
// registers in the MFMA result layout, which IS the B-operand layout, so the chain over i has no
// memory round trip through W; only L_ik and inv(L_ii) are loaded (coalesced, from the upper
// mirror).  The transposed tile for WU comes from the same operand registers with the roles of
// A and B swapped, so both outputs are written with coalesced stores.
// ---------------------------------------------------------------------------
template<int NT, int MB>
__device__ PQP_CALL void
tri_inverse_mfma(cgptr F, int ld, int n, gptr WL, gptr WU)
{
  constexpr int NB = 16;
  constexpr int NWV = NT / WAVE;
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  const int lr = lane & 15, lk = lane >> 4;
  const int nbk = (n + NB - 1) / NB;
  // WL's strict upper and WU's strict lower triangle are never written by anybody and stay at the
  // zero they were allocated with: only the unit diagonal is (re)written here
  for (int k = threadIdx.x; k < n; k += NT) {
    WL[(long)k * ld + k] = 1.0;
    WU[(long)k * ld + k] = 1.0;
  }
  __syncthreads();
  for (int j = w; j < nbk; j += NWV) {
    const int j0 = j * NB;
    pqp_d4 Wt[MB];
    {
      // diagonal tile: strict lower of F's block = inv(L_jj), strict upper = its transpose
      pqp_d4 tl = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = lk + 4 * q, col = lr;
        const int gr = j0 + row, gc = j0 + col;
        const int grc = (gr < n) ? gr : n - 1, gcc = (gc < n) ? gc : n - 1;
        const double v = F[(long)grc * ld + gcc];
        const bool in = gr < n && gc < n;
        tl[q] = (row > col && in) ? v : ((row == col) ? 1.0 : 0.0);
        if (in && row != col) {
          if (row > col)
            WL[(long)gr * ld + gc] = v;
          else
            WU[(long)gr * ld + gc] = v;
        }
      }
#pragma unroll
      for (int kk = 0; kk < MB; ++kk)
        if (kk == j)
          Wt[kk] = tl;
    }
#pragma unroll
    for (int i = 1; i < MB; ++i) {
      if (i > j && i < nbk) {
        const int i0 = i * NB;
        const int ir = i0 + lr;
        const int irc = (ir < n) ? ir : n - 1;
        pqp_d4 T = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
        for (int k = 0; k < i; ++k) {
          if (k >= j) {
            double a[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
              a[q] = F[(long)(k * NB + 4 * q + lk) * ld + irc]; // L[i0+lr][k0+4q+lk] (upper mirror)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              T = mfma_f64_16x16x4((ir < n) ? a[q] : 0.0, Wt[k][q], T);
          }
        }
        double iv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = 4 * q + lk, r = lr; // inv(L_ii)[r][c], read from the transposed upper half
          const int gc = i0 + c, gr = i0 + r;
          const int gcc = (gc < n) ? gc : n - 1, grc = (gr < n) ? gr : n - 1;
          const double v = F[(long)gcc * ld + grc];
          iv[q] = (r > c && gc < n && gr < n) ? v : ((r == c) ? 1.0 : 0.0);
        }
        pqp_d4 Wij = { 0.0, 0.0, 0.0, 0.0 }, WijT = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          Wij = mfma_f64_16x16x4(iv[q], T[q], Wij);   // inv(L_ii) * T
          WijT = mfma_f64_16x16x4(T[q], iv[q], WijT); // T^T * inv(L_ii)^T
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          Wij[q] = -Wij[q];
          const int rowl = i0 + lk + 4 * q; // WL[i-block row][j-block col]
          if (rowl < n)
            WL[(long)rowl * ld + j0 + lr] = Wij[q];
          const int rowu = j0 + lk + 4 * q; // WU[j-block row][i-block col]
          if (ir < n)
            WU[(long)rowu * ld + ir] = -WijT[q];
        }
        Wt[i] = Wij;
      }
    }
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------
// wave-level ordering point for wave-synchronous LDS exchange (one wavefront writes LDS and the
// same wavefront reads other lanes' values): the hardware executes a wave's LDS operations in
// order, so only the compiler has to be stopped from reordering; the emulator supplies a real
// wave barrier.
// ---------------------------------------------------------------------------
#ifndef PQP_EMULATED_MFMA
__device__ __forceinline__ void
wave_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#endif

// ---------------------------------------------------------------------------
// Blocked LDL^T on the FP64 matrix cores for any m <= NT, matrix in HBM / L2 (full symmetric
// row-major input, leading dimension ld; only the upper triangle is read).  Restates what the
// reference's factorisation computes (include/proxsuite/linalg/dense/factorize.hpp:89-148, 215-280:
// D from the diagonal recurrence, L = unit lower) with a static pivot order.  Outputs:
//   upper mirror  M[k][i] = L[i][k] (i > k),   M[j][j] = d_j,   d[] in LDS;
//   FULL: lower mirror outside the diagonal blocks, diagonal blocks = inv(L_bb) / inv(L_bb)^T.
// Left-looking over 16-column panels kb, three phases per panel, one barrier after each:
//   P1  every wavefront: tiles U(kb, x), x >= kb, of the upper triangle get
//         U(kb,x) -= sum_{p<kb} U(p,kb)^T D_p U(p,x)
//       (4 MFMAs per (x, p); operands are rows of U, loaded coalesced in the MFMA layouts);
//   P2  wavefront 0: the 16x16 diagonal tile is factorised with one ROW per lane, pivot rows
//       travelling through scalar registers (v_readlane), no LDS traffic and no barrier inside;
//       inv(L_11) from the Neumann product (diag_block_inverses_mfma's identity);
//   P3  every wavefront: U(kb,x) <- D_1^{-1} inv(L_11) U(kb,x) for x > kb (4 MFMAs per tile),
//       the transposed tile for the lower mirror from the swapped operands.
// `top`: 2*256 + 32 doubles of LDS.
// ---------------------------------------------------------------------------
template<int NT, bool FULL>
__device__ PQP_CALL void
ldlt_factor_mfma(gptr M, int ld, int m, lptr d, lptr top)
{
  constexpr int NB = 16;
  constexpr int NWV = NT / WAVE;
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  const int lr = lane & 15, lk = lane >> 4;
  const int nbk = (m + NB - 1) / NB;
  lptr tile = top;        // 16 x 16 diagonal tile, row-major
  lptr mop = top + 256;   // TRSM operand in register order [q][lane]
  lptr dinv = top + 512;  // 16 inverse pivots
  for (int kb = 0; kb < nbk; ++kb) {
    const int k0 = kb * NB;
    // ---- P1: update block row kb of the upper triangle
    // Addressing: row (p * 16 + 4 q + lk) of M = a wave-uniform base (scalar registers) + a per-lane byte offset that does
    // not change in the loop, so the loads carry no vector address arithmetic.  Operand lanes beyond column m hold
    // clamped (finite) duplicates and are NOT zeroed: lane lr of the A operand only reaches row lr of the tile, lane lr of
    // the B operand only its column lr, and rows / columns beyond m are never stored (P2 masks the diagonal tile it reads).
    for (int x = kb + w; x < nbk; x += NWV) {
      const int x0 = x * NB;
      const int xc = x0 + lr;
      const int xcc = (xc < m) ? xc : m - 1;
      const int kcol = k0 + lr;
      const int kcc = (kcol < m) ? kcol : m - 1;
      const unsigned offk = (unsigned)(lk * ld + kcc) * 8u, offx = (unsigned)(lk * ld + xcc) * 8u;
      const PQP_GLOBAL char* Mb = reinterpret_cast<const PQP_GLOBAL char*>(M);
      pqp_d4 acc;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = k0 + lk + 4 * q;
        const int rowc = (row < m) ? row : m - 1;
        const double v = M[(long)rowc * ld + xcc];
        acc[q] = (row < m && xc < m) ? v : 0.0;
      }
      for (int p = 0; p < kb; p += 2) {
        double ap[2][4], bp[2][4], dp[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int pb = ((p + h < kb) ? (p + h) : p) * NB + 4 * q; // wave-uniform row of the group; a full block: < m
            const PQP_GLOBAL char* rb = Mb + (long)pb * ld * 8;
            ap[h][q] = *reinterpret_cast<const PQP_GLOBAL double*>(rb + offk);
            bp[h][q] = *reinterpret_cast<const PQP_GLOBAL double*>(rb + offx);
            dp[h][q] = d[pb + lk];
          }
#pragma unroll
        for (int h = 0; h < 2; ++h)
          if (p + h < kb) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              acc = mfma_f64_16x16x4(-ap[h][q] * dp[h][q], bp[h][q], acc);
          }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = k0 + lk + 4 * q;
        if (row < m && xc < m)
          M[(long)row * ld + xc] = acc[q];
        if (x == kb)
          tile[(lk + 4 * q) * NB + lr] = acc[q];
      }
    }
    __syncthreads();
    // ---- P2: wavefront 0 factorises the diagonal tile, one row per lane (lanes 0..15)
    if (w == 0) {
      const int nb = (m - k0 < NB) ? (m - k0) : NB;
      const int r = lane & 15;
      double a[NB];
#pragma unroll
      for (int c = 0; c < NB; ++c) {
        const double v = tile[r * NB + c];
        a[c] = (r < nb && c < nb) ? v : ((r == c) ? 1.0 : 0.0); // identity padding
      }
#pragma unroll
      for (int c = 0; c < NB; ++c) {
        const double dc = lane_bcast(a[c], c);
        const double l = a[c] / dc;
#pragma unroll
        for (int cp = c + 1; cp < NB; ++cp) {
          const double mcp = lane_bcast(a[c], cp); // A[cp][c] before scaling
          if (r >= cp)
            a[cp] = fma(-l, mcp, a[cp]);
        }
        if (r > c)
          a[c] = l;
      }
      wave_sync();
      if (lane < NB) {
#pragma unroll
        for (int c = 0; c < NB; ++c)
          tile[r * NB + c] = (c < r) ? a[c] : 0.0; // strict lower N of L_11
        double dr = 1.0;
#pragma unroll
        for (int c = 0; c < NB; ++c)
          if (c == r)
            dr = a[c];
        dinv[r] = 1.0 / dr;
        if (r < nb)
          d[k0 + r] = dr;
      }
      wave_sync();
      TilePair N, Pm;
      bool dg[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = lk + 4 * q, col = lr;
        N.x[q] = (row > col) ? tile[row * NB + col] : 0.0;
        N.xt[q] = (row < col) ? tile[col * NB + row] : 0.0;
        dg[q] = (row == col);
        Pm.x[q] = (dg[q] ? 1.0 : 0.0) - N.x[q];
        Pm.xt[q] = (dg[q] ? 1.0 : 0.0) - N.xt[q];
      }
      TilePair S = tile_mul(N, N);
#pragma unroll
      for (int rep = 0; rep < 3; ++rep) {
        TilePair T = S;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (dg[q]) {
            T.x[q] += 1.0;
            T.xt[q] += 1.0;
          }
        Pm = tile_mul(Pm, T);
        if (rep < 2)
          S = tile_mul(S, S);
      }
      const double dcol = dinv[lr];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // A operand of  D_1^{-1} inv(L_11)  =  registers of the tile  inv(L_11)^T D_1^{-1}
        mop[q * WAVE + lane] = Pm.xt[q] * dcol;
        const int row = lk + 4 * q, col = lr;
        const int gr = k0 + row, gc = k0 + col;
        if (gr < m && gc < m && row != col) {
          if (FULL)
            M[(long)gr * ld + gc] = (row > col) ? Pm.x[q] : Pm.xt[q];
          else if (row < col)
            M[(long)gr * ld + gc] = N.xt[q]; // L_11^T in the upper mirror
        }
        if (gr < m && row == col)
          M[(long)gr * ld + gc] = d[gr];
      }
    }
    __syncthreads();
    // ---- P3: the panel to the right of the diagonal tile
    for (int x = kb + 1 + w; x < nbk; x += NWV) {
      const int x0 = x * NB;
      const int xc = x0 + lr;
      const int xcc = (xc < m) ? xc : m - 1;
      double a[4], b[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        a[q] = mop[q * WAVE + lane];
        b[q] = M[(long)(k0 + lk + 4 * q) * ld + xcc]; // k0 block is full here (x > kb exists)
      }
      pqp_d4 res = { 0.0, 0.0, 0.0, 0.0 }, resT = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double bv = (xc < m) ? b[q] : 0.0;
        res = mfma_f64_16x16x4(a[q], bv, res);
        if (FULL)
          resT = mfma_f64_16x16x4(bv, a[q], resT);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (xc < m)
          M[(long)(k0 + lk + 4 * q) * ld + xc] = res[q];
        if (FULL) {
          const int row = x0 + lk + 4 * q;
          if (row < m)
            M[(long)row * ld + k0 + lr] = resT[q];
        }
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// tri_inverse on the matrix cores for any n (the register-resident variant above needs
// n <= 16 * MB): block ROW i of W = L^{-1} at a time, its tiles spread over the wavefronts,
//     W_ij = -inv(L_ii) * sum_{k=j}^{i-1} L_ik W_kj,
// W_kj read back from WL (rows k < i are complete: one barrier per block row).
// ---------------------------------------------------------------------------
template<int NT, bool WRITE_WU = true>
__device__ PQP_CALL void
tri_inverse_mfma_rows(cgptr F, int ld, int n, gptr WL, gptr WU)
{
  constexpr int NB = 16;
  constexpr int NWV = NT / WAVE;
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  const int lr = lane & 15, lk = lane >> 4;
  const int nbk = (n + NB - 1) / NB;
  // WL's strict upper and WU's strict lower triangle are never written by anybody and stay at the
  // zero they were allocated with: only the unit diagonal is (re)written here
  for (int k = threadIdx.x; k < n; k += NT) {
    WL[(long)k * ld + k] = 1.0;
    if (WRITE_WU)
      WU[(long)k * ld + k] = 1.0;
  }
  __syncthreads();
  for (int j = w; j < nbk; j += NWV) {
    const int j0 = j * NB;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = lk + 4 * q, col = lr;
      const int gr = j0 + row, gc = j0 + col;
      if (gr < n && gc < n && row != col) {
        const double v = F[(long)gr * ld + gc];
        if (row > col)
          WL[(long)gr * ld + gc] = v;
        else if (WRITE_WU)
          WU[(long)gr * ld + gc] = v;
      }
    }
  }
  __syncthreads();
  for (int i = 1; i < nbk; ++i) {
    const int i0 = i * NB;
    const int ir = i0 + lr;
    const int irc = (ir < n) ? ir : n - 1;
    for (int j = w; j < i; j += NWV) {
      const int j0 = j * NB;
      pqp_d4 T = { 0.0, 0.0, 0.0, 0.0 };
      // (addressing as in ldlt_factor_mfma's P1: wave-uniform row base + loop-invariant lane offset; the A-operand lanes of
      // rows beyond n hold clamped duplicates and only reach rows of the tile that are never stored)
      const unsigned offa = (unsigned)(lk * ld + irc) * 8u, offb = (unsigned)(lk * ld + j0 + lr) * 8u;
      const PQP_GLOBAL char* Fb = reinterpret_cast<const PQP_GLOBAL char*>(F);
      const PQP_GLOBAL char* Wb = reinterpret_cast<const PQP_GLOBAL char*>(WL);
      for (int k = j; k < i; k += 2) {
        double a[2][4], b[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const long rowoff = (long)(((k + h < i) ? (k + h) : k) * NB + 4 * q) * ld * 8; // wave-uniform; full block: < n
            a[h][q] = *reinterpret_cast<const PQP_GLOBAL double*>(Fb + rowoff + offa); // L[i0+lr][kr]   (upper mirror)
            b[h][q] = *reinterpret_cast<const PQP_GLOBAL double*>(Wb + rowoff + offb); // W[kr][j0+lr]
          }
#pragma unroll
        for (int h = 0; h < 2; ++h)
          if (k + h < i) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              T = mfma_f64_16x16x4(a[h][q], b[h][q], T);
          }
      }
      double iv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 4 * q + lk, r = lr;
        const int gc = i0 + c, gr = i0 + r;
        const int gcc = (gc < n) ? gc : n - 1, grc = (gr < n) ? gr : n - 1;
        const double v = F[(long)gcc * ld + grc];
        iv[q] = (r > c && gc < n && gr < n) ? v : ((r == c) ? 1.0 : 0.0);
      }
      pqp_d4 Wij = { 0.0, 0.0, 0.0, 0.0 }, WijT = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        Wij = mfma_f64_16x16x4(iv[q], T[q], Wij);
        WijT = mfma_f64_16x16x4(T[q], iv[q], WijT);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rowl = i0 + lk + 4 * q;
        if (rowl < n)
          WL[(long)rowl * ld + j0 + lr] = -Wij[q];
        if (WRITE_WU && ir < n)
          WU[(long)(j0 + lk + 4 * q) * ld + ir] = -WijT[q];
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// LDS-tiled GEMMs of the Z / G build (Solver::build_ZG): the workgroup computes one TM x TN block of the result at a
// time, every wavefront a 32 x 32 sub-block (2 x 2 MFMA tiles), and the operand rows of a slab of KS = 16 steps of the
// reduction index are staged ONCE per block through LDS for all wavefronts -- the per-wavefront form (each wavefront
// streaming its own operand panels from L2 / HBM) re-read every operand byte about twenty times at C4 and seven at C2
// (profiles/r04_pmc_c4_waste_by_phase.txt: 139 MB of traffic for 20 MB of operands).  The next slab travels in
// registers while the current one is multiplied (one staging buffer in LDS).  Both operands are "reduction-major":
// element (j, i) of an operand is row j of a row-major matrix, so the staging loads are coalesced.
//   NT = 256: 2 x 2 wavefronts, TM = TN = 64;   512: 2 x 4, TM = 64, TN = 128;   1024: 4 x 4, TM = TN = 128.
// Accumulation order = ascending reduction index, four per MFMA, exactly as in the per-wavefront form: the results
// are the same bits (terms that were skipped there are exact zeros here).
// `stage`: zg_stage_doubles(NT) doubles of LDS.
// ---------------------------------------------------------------------------
constexpr int ZG_KS = 16; // reduction steps per slab
template<int NT>
struct ZgTile
{
  static constexpr int NWV = NT / WAVE;
  static constexpr int WR = (NWV == 16) ? 4 : 2; // wavefront grid
  static constexpr int WC = NWV / WR;
  static constexpr int TM = 32 * WR, TN = 32 * WC;
  static constexpr int TMP = TM + 8, TNP = TN + 8; // padded row strides (doubles) of the staged slabs
  static constexpr int SLAB = ZG_KS * (TMP + TNP);
  static constexpr int SCRATCH = NWV * 16 * 17; // per-wavefront 16 x 16 transposition tiles (stride 17)
  static constexpr int STAGE = SLAB > SCRATCH ? SLAB : SCRATCH;
  static constexpr int PER_A = (ZG_KS * TM + NT - 1) / NT; // staged elements per thread and operand
  static constexpr int PER_B = (ZG_KS * TN + NT - 1) / NT;
};
__host__ __device__ inline int
zg_stage_doubles(int nt)
{
  return nt == 256 ? ZgTile<256>::STAGE : (nt == 512 ? ZgTile<512>::STAGE : ZgTile<1024>::STAGE);
}

// one workgroup block: acc[h][g] += sum_j A(j, R0 + ...) B(j, C0 + ...) over j in [0, jend); the loaders return the
// operand element (j, i) or 0 outside the matrix.  Leaves the 2 x 2 tiles of this wavefront in acc.
template<int NT, typename LoadA, typename LoadB>
__device__ __forceinline__ void
zg_block(LoadA loadA, LoadB loadB, int R0, int C0, int jend, lptr stage, pqp_d4 (&acc)[2][2])
{
  using T = ZgTile<NT>;
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  const int lr = lane & 15, lk = lane >> 4;
  const int wr = w / T::WC, wc = w - wr * T::WC;
  lptr As = stage, Bs = stage + ZG_KS * T::TMP;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int g = 0; g < 2; ++g)
      acc[h][g] = pqp_d4{ 0.0, 0.0, 0.0, 0.0 };
  double ra[T::PER_A], rb[T::PER_B];
  auto fetch = [&](int j0) {
#pragma unroll
    for (int i = 0; i < T::PER_A; ++i) {
      const int e = threadIdx.x + i * NT;
      const int jj = e / T::TM, kk = e - jj * T::TM;
      ra[i] = (e < ZG_KS * T::TM && j0 + jj < jend) ? loadA(j0 + jj, R0 + kk) : 0.0;
    }
#pragma unroll
    for (int i = 0; i < T::PER_B; ++i) {
      const int e = threadIdx.x + i * NT;
      const int jj = e / T::TN, cc = e - jj * T::TN;
      rb[i] = (e < ZG_KS * T::TN && j0 + jj < jend) ? loadB(j0 + jj, C0 + cc) : 0.0;
    }
  };
  fetch(0);
  for (int j0 = 0; j0 < jend; j0 += ZG_KS) {
    __syncthreads(); // the previous slab has been consumed by every wavefront
#pragma unroll
    for (int i = 0; i < T::PER_A; ++i) {
      const int e = threadIdx.x + i * NT;
      const int jj = e / T::TM, kk = e - jj * T::TM;
      if (e < ZG_KS * T::TM)
        As[jj * T::TMP + kk] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < T::PER_B; ++i) {
      const int e = threadIdx.x + i * NT;
      const int jj = e / T::TN, cc = e - jj * T::TN;
      if (e < ZG_KS * T::TN)
        Bs[jj * T::TNP + cc] = rb[i];
    }
    __syncthreads();
    if (j0 + ZG_KS < jend)
      fetch(j0 + ZG_KS); // in flight while this slab is multiplied
#pragma unroll
    for (int q = 0; q < ZG_KS / 4; ++q) {
      double a[2], b[2];
#pragma unroll
      for (int h = 0; h < 2; ++h)
        a[h] = As[(4 * q + lk) * T::TMP + wr * 32 + h * 16 + lr];
#pragma unroll
      for (int g = 0; g < 2; ++g)
        b[g] = Bs[(4 * q + lk) * T::TNP + wc * 32 + g * 16 + lr];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int g = 0; g < 2; ++g)
          acc[h][g] = mfma_f64_16x16x4(a[h], b[g], acc[h][g]);
    }
  }
  __syncthreads(); // the staging buffer is free again (the callers transpose through it)
}

// transposed copy of one 16 x 16 MFMA result tile through this wavefront's own LDS tile: returns t with
// t[r] = element (row = lr, column = lk + 4 r) of the tile whose lane layout is  acc[r] = element (lk + 4 r, lr)
__device__ __forceinline__ pqp_d4
zg_transpose(const pqp_d4& acc, lptr sc)
{
  const int lane = threadIdx.x & (WAVE - 1);
  const int lr = lane & 15, lk = lane >> 4;
  wave_sync();
#pragma unroll
  for (int r = 0; r < 4; ++r)
    sc[(lk + 4 * r) * 17 + lr] = acc[r];
  wave_sync();
  pqp_d4 t;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    t[r] = sc[lr * 17 + lk + 4 * r];
  return t;
}

// ---------------------------------------------------------------------------
// Symmetric rank-K accumulation on the FP64 matrix cores:
//     out[i][j] = base[i][j] + alpha * sum_{k < K} M[row(k)][i] * M[row(k)][j]        i, j < m
// M row-major (leading dimension ldm), rows optionally gathered through `rowmap` (LDS; row(k) =
// rowmap[k]); out / base row-major m x m (leading dimension ld), base may be null (zero) or alias
// out.  One 16x16 lower tile per wavefront at a time; the mirror tile comes from the same operand
// registers with the roles swapped, so both are written with coalesced stores and the result is
// exactly symmetric.  Used by the PrimalLDLT engine for A^T A and C_J^T C_J.
// ---------------------------------------------------------------------------
template<int NT>
__device__ PQP_CALL void
syrk_mfma(cgptr M, int ldm, int K, cliptr rowmap, int m, double alpha, cgptr base, gptr out, int ld)
{
  constexpr int NWV = NT / WAVE;
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  const int lr = lane & 15, lk = lane >> 4;
  const int MT = (m + 15) / 16;
  const int tiles = MT * (MT + 1) / 2;
  for (int t = w; t < tiles; t += NWV) {
    int ct = 0, rem = t;
    while (rem >= ct + 1) {
      rem -= ct + 1;
      ++ct;
    }
    const int dt = rem; // dt <= ct
    const int c0 = ct * 16, d0 = dt * 16;
    const int c = c0 + lr, dcol = d0 + lr;
    const bool c_ok = c < m, d_ok = dcol < m;
    const int cc = c_ok ? c : m - 1, dc = d_ok ? dcol : m - 1;
    pqp_d4 acc1 = { 0.0, 0.0, 0.0, 0.0 }, acc2 = { 0.0, 0.0, 0.0, 0.0 };
    constexpr int DEPTH = 8;
    for (int k0 = 0; k0 < K; k0 += 4 * DEPTH) {
      double a[DEPTH], b[DEPTH];
#pragma unroll
      for (int u = 0; u < DEPTH; ++u) {
        const int k = k0 + 4 * u + lk;
        const int kc = (k < K) ? k : K - 1;
        const long row = rowmap ? (long)rowmap[kc] : (long)kc;
        a[u] = M[row * ldm + cc];
        b[u] = M[row * ldm + dc];
      }
#pragma unroll
      for (int u = 0; u < DEPTH; ++u) {
        const int k = k0 + 4 * u + lk;
        const double av = (k < K && c_ok) ? a[u] : 0.0;
        const double bv = (k < K && d_ok) ? b[u] : 0.0;
        acc1 = mfma_f64_16x16x4(av, bv, acc1);
        acc2 = mfma_f64_16x16x4(bv, av, acc2);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cr = c0 + lk + 4 * q; // row of the (ct, dt) tile
      if (ct != dt) {
        if (cr < m && d_ok) {
          const long o = (long)cr * ld + dcol;
          out[o] = (base ? base[o] : 0.0) + alpha * acc1[q];
        }
        const int dr = d0 + lk + 4 * q; // row of the mirror tile
        if (dr < m && c_ok) {
          const long o = (long)dr * ld + c;
          out[o] = (base ? base[o] : 0.0) + alpha * acc2[q];
        }
      } else if (cr < m && d_ok && cr >= dcol) {
        const long o = (long)cr * ld + dcol, ot = (long)dcol * ld + cr;
        const double v = (base ? base[o] : 0.0) + alpha * acc1[q];
        out[o] = v;
        out[ot] = v;
      }
    }
  }
  __syncthreads();
}

// exclusive prefix count of a per-thread flag over the block; returns this
// thread's rank among the set flags and the total through `total`.
// `cnt` is LDS scratch of NT/64 + 1 ints.
template<int NT>
__device__ __forceinline__ int
block_rank(bool flag, liptr cnt, int& total)
{
  constexpr int NW = NT / WAVE;
  unsigned long long m = __ballot(flag ? 1 : 0);
  const int lane = threadIdx.x & (WAVE - 1);
  const int wid = threadIdx.x / WAVE;
  int before = __popcll(m & ((lane == 0) ? 0ull : (~0ull >> (WAVE - lane))));
  if (lane == 0)
    cnt[wid] = __popcll(m);
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    int c = cnt[w];
    if (w < wid)
      off += c;
    tot += c;
  }
  total = uni(tot);
  __syncthreads();
  return off + before;
}

} // namespace pqp

#endif
