// The kernels of libproxqp_hip.so and their launchers.  Compiled once per kernel family
// (-DPQP_TU=1..9, objects built in parallel by proxsuite_amd/_build.py); PQP_TU undefined or 0
// compiles everything in one translation unit (the CPU emulator build of tests/emu does that).
//   1  pqp_solve_kernel<256, 4, 1>    no box constraints, dense Hessian (the C2 kernel): 128 VGPRs per
//                                     lane, FOUR workgroups per CU -- launches that fill the device
//   7  pqp_solve_kernel<256, 3, 1>    the same solve with 168 VGPRs per lane, three workgroups per CU:
//                                     lower latency per QP, used when the launch leaves CUs idle anyway
//   2  pqp_solve_kernel<256, 3, 0>
//   3  pqp_solve_kernel<512, 2, .>    256 VGPRs, one workgroup per CU
//   8  pqp_solve_kernel<512, 4, .>    128 VGPRs, two workgroups per CU: launches of more workgroups than CUs
//   4  pqp_solve_kernel<1024, ., .>
//   5  pqp_backward_kernel<.>
//   6  pqp_setup_kernel<.>, pqp_order_kernel, the dispatchers
#ifndef PQP_TU
#define PQP_TU 0
#endif
//  12  pqp_solve_kernel<256, 3, 2>    the diagonal-structure solver alone (H diagonal / zero, no equality, every inequality
//                                     row on one variable: BASELINE.json configs[4]): launches whose QPs ALL have it
//   9  pqp_solve_hbm_kernel<1024, .>  shapes whose per-QP vectors exceed the CU's 160 KiB of LDS: the
//                                     SAME solver with its "LDS" pointers typed as global memory and
//                                     carved out of a per-workgroup slice of an HBM scratch buffer (the
//                                     workgroup barrier orders global accesses within a workgroup as it
//                                     does LDS ones).  Slow per QP -- every vector access is an L1/L2 round
//                                     trip -- but it removes the size ceiling below 1024 rows.
#if PQP_TU == 9
#define PQP_VECTORS_IN_HBM 1
#define PQP_LDS __attribute__((address_space(1)))
#define PQP_GLOBAL __attribute__((address_space(1)))
#endif
// 8 instead of 16 matrix loads in flight per lane in gemv for the 128-VGPR kernel (pqp_block.hpp)
#if PQP_TU == 1 && !defined(PQP_GEMV_DEEP_256)
#define PQP_GEMV_DEEP_256 0
#endif
// the translation units whose kernels run at 128 VGPRs per lane keep 4 instead of 8 MFMA k-steps of
// operand loads in flight in the Z / G build (+2 % at C2 and C4, profiles/r02_ab_compiler_flags.txt)
#if (PQP_TU == 1 || PQP_TU == 4 || PQP_TU == 8 || PQP_TU == 9) && !defined(PQP_ZG_DEPTH)
#define PQP_ZG_DEPTH 4
#endif
// (the one-wavefront diagonal kernel is not short of scalar registers: its per-QP pointers are ordinary values -- the
// optimisation barriers that keep them from being hoisted in the workgroup kernels would pin them to SGPRs inside
// lane-divergent code here)
#if PQP_TU == 16 || PQP_TU == 17
#define PQP_OPAQUE_SCALAR(v)
#define PQP_OPAQUE_VECTOR(v)
#endif
//  17  pqp_dwave_kernel<WPS>          the DENSE solver as one wavefront per QP (pqp_dwave.hpp): dense Hessian, no box, n, n_eq,
//                                     n_in <= 128 -- the iteration of a solve; launches that fill the device
//  18  pqp_prologue_kernel<256>       the factorisation prologue of those solves (Solver::prologue), 256 threads per QP
#include "pqp_host.hpp"
#include "pqp_dwave.hpp"

#define PQP_TU_HAS(k) (PQP_TU == 0 || PQP_TU == (k))

// Waves per SIMD the register allocator must leave room for (512 / WPS VGPRs per lane): the
// knob that trades spills against resident workgroups per CU.  Compile-time only.
#ifndef PQP_WPS_256_DENSE
#define PQP_WPS_256_DENSE 4 // throughput variant of the C2 kernel (LDS: 4 x 40.9 KB fits the CU's 160 KB)
#endif
#ifndef PQP_WPS_256
#define PQP_WPS_256 3
#endif
#ifndef PQP_WPS_512
#define PQP_WPS_512 2
#endif
#ifndef PQP_WPS_512_DENSE
#define PQP_WPS_512_DENSE 4
#endif
#ifndef PQP_WPS_1024
#define PQP_WPS_1024 4
#endif

template<int NT, int WPS, int SPEC>
__global__ __launch_bounds__(NT, WPS) void
pqp_solve_kernel(pqp::Batch batch, long first, const int* __restrict__ order)
{
  HIP_DYNAMIC_SHARED(double, smem)
  // `order` (optional) is the dispatch order of the QPs: workgroups are handed out in blockIdx
  // order, so listing the expensive QPs first shortens the tail of the launch
  const long slot = order ? (long)order[blockIdx.x] : (long)blockIdx.x;
  pqp::solve_body<NT, SPEC>(batch, first + slot, (pqp::lptr)smem);
}

template<int NT, int WPS, int SPEC>
static int
launch_solve(pqp_batch* h)
{
  // (a kernel narrower than the handle's default width carves its own, smaller, LDS layout)
  const size_t lds = (NT == h->nt) ? h->lds_solve : pqp::lds_bytes(h->dev.d, NT);
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&pqp_solve_kernel<NT, WPS, SPEC>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  HIP_TRY(hipEventRecord(h->ev0, h->stream));
  const bool whole = h->range_first == 0 && h->range_count == h->dev.B;
  const int* order = h->subset_order ? h->subset_order : ((h->lpt && h->order_valid && whole) ? h->d_order : nullptr);
  hipLaunchKernelGGL((pqp_solve_kernel<NT, WPS, SPEC>), dim3((unsigned)h->range_count), dim3(NT), lds,
                     h->stream, h->dev, h->range_first, order);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(h->ev1, h->stream));
  return PQP_OK;
}

// SPEC = 1: no box constraints and a dense Hessian, both known at compile time
int pqp_launch_solve_256_s1(pqp_batch* h);
int pqp_launch_solve_256_s1_lat(pqp_batch* h);
int pqp_launch_solve_256_s1_one(pqp_batch* h);
int pqp_launch_solve_256_s0_one(pqp_batch* h);
int pqp_launch_solve_256_s1_two(pqp_batch* h);
int pqp_launch_solve_256_s0(pqp_batch* h);
int pqp_launch_solve_256_s2(pqp_batch* h);
int pqp_launch_solve_diag_wave(pqp_batch* h);
int pqp_diag_wave_slots(int dim);
int pqp_launch_dense_wave(pqp_batch* h);
int pqp_launch_prologue(pqp_batch* h, long first, long count, hipStream_t stream);
int pqp_launch_solve_512(pqp_batch* h, bool common);
int pqp_launch_solve_512_dense(pqp_batch* h, bool common);
int pqp_launch_solve_1024(pqp_batch* h, bool common);
int pqp_launch_solve_hbm(pqp_batch* h, bool common);

#if PQP_TU_HAS(1)
int
pqp_launch_solve_256_s1(pqp_batch* h)
{
  return launch_solve<256, PQP_WPS_256_DENSE, 1>(h);
}
#endif

#if PQP_TU_HAS(7)
int
pqp_launch_solve_256_s1_lat(pqp_batch* h)
{
  return launch_solve<256, PQP_WPS_256, 1>(h);
}
#endif
// One workgroup per CU: a launch of no more workgroups than the device has CUs (C1: 128 QPs, a single QP::solve())
// leaves every QP a CU of its own, so the whole register file of a SIMD may go to its wavefront -- no spills at all
// (100 spilled VGPRs at three per CU).  C1: 1.318 -> 1.245 ms (profiles/r04_ab_small_launch_budget.txt).
#if PQP_TU_HAS(13)
int
pqp_launch_solve_256_s1_one(pqp_batch* h)
{
  return launch_solve<256, 1, 1>(h);
}
#endif
#if PQP_TU_HAS(15)
int
pqp_launch_solve_256_s1_two(pqp_batch* h) // (up to two workgroups per CU: 256 VGPRs; 384 / 512 QPs of the C2 shape -3 %)
{
  return launch_solve<256, 2, 1>(h);
}
#endif
#if PQP_TU_HAS(14)
int
pqp_launch_solve_256_s0_one(pqp_batch* h) // (the same for the general kernel: boxes, sparse Hessian types, PrimalLDLT)
{
  return launch_solve<256, 1, 0>(h);
}
#endif
#if PQP_TU_HAS(2)
int
pqp_launch_solve_256_s0(pqp_batch* h)
{
  return launch_solve<256, PQP_WPS_256, 0>(h);
}
#endif
#ifndef PQP_WPS_256_DIAG
#define PQP_WPS_256_DIAG 2 // (two workgroups per CU at 256 VGPRs: 4.08 ms per 4096 C5 QPs against 4.29 ms with three at 168, profiles/r04_ab_diag_kernel.txt)
#endif
#if PQP_TU_HAS(12)
int
pqp_launch_solve_256_s2(pqp_batch* h)
{
  return launch_solve<256, PQP_WPS_256_DIAG, 2>(h);
}
#endif
// One WAVEFRONT per QP, every per-QP vector in registers: the diagonal-structure solver of pqp_diag.hpp
// (BASELINE.json configs[4]).  A workgroup is one wavefront; E = 4 register slots per vector serve dim <= 256, the
// range of the 256-thread kernel class.
#if PQP_TU_HAS(16)
#ifndef PQP_WPS_DIAG_WAVE
#define PQP_WPS_DIAG_WAVE 2
#endif
template<int E, int WPS>
__global__ __launch_bounds__(64, WPS) void
pqp_diag_kernel(pqp::Batch batch, long first, const int* __restrict__ order)
{
  HIP_DYNAMIC_SHARED(double, smem)
  const long slot = order ? (long)order[blockIdx.x] : (long)blockIdx.x;
  pqp::diag_solve_body<E>(batch, first + slot, (pqp::lptr)smem);
}

template<int E>
static int
launch_diag_wave(pqp_batch* h)
{
  HIP_TRY(hipEventRecord(h->ev0, h->stream));
  const bool whole = h->range_first == 0 && h->range_count == h->dev.B;
  const int* order = h->subset_order ? h->subset_order : ((h->lpt && h->order_valid && whole) ? h->d_order : nullptr);
  hipLaunchKernelGGL((pqp_diag_kernel<E, PQP_WPS_DIAG_WAVE>), dim3((unsigned)h->range_count), dim3(64),
                     pqp::diag_lds_bytes(E), h->stream, h->dev, h->range_first, order);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(h->ev1, h->stream));
  return PQP_OK;
}

// register slots per vector by dimension: 1 (dim <= 64), 2 (<= 128), 4 (<= 256) -- a slot is a pass of every element-wise
// loop over all 64 lanes, so a QP of dim 60 does a quarter of the vector instructions of one of dim 200
int
pqp_diag_wave_slots(int dim)
{
  return dim <= 64 ? 1 : (dim <= 128 ? 2 : 4);
}

int
pqp_launch_solve_diag_wave(pqp_batch* h)
{
  switch (pqp_diag_wave_slots(h->dev.d.n)) {
    case 1:
      return launch_diag_wave<1>(h);
    case 2:
      return launch_diag_wave<2>(h);
    default:
      return launch_diag_wave<4>(h);
  }
}
#endif
// One WAVEFRONT per QP for DENSE QPs (pqp_dwave.hpp) behind the 256-thread factorisation prologue: two launches on the
// handle's stream, one pair of events around both.
#if PQP_TU_HAS(18)
template<int NT>
__global__ __launch_bounds__(NT, 4) void
pqp_prologue_kernel(pqp::Batch batch, long first, const int* __restrict__ order)
{
  HIP_DYNAMIC_SHARED(double, smem)
  const long slot = order ? (long)order[blockIdx.x] : (long)blockIdx.x;
  pqp::Solver<NT, 1> S(batch, first + slot, (pqp::lptr)smem);
  S.prologue();
}

// (first, count): the QPs of the launch this call covers (slots of the launch's order); stream: where it is enqueued
int
pqp_launch_prologue(pqp_batch* h, long first, long count, hipStream_t stream)
{
  const size_t lds = (256 == h->nt) ? h->lds_solve : pqp::lds_bytes(h->dev.d, 256);
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&pqp_prologue_kernel<256>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const bool whole = h->range_first == 0 && h->range_count == h->dev.B;
  const int* order = h->subset_order ? h->subset_order : ((h->lpt && h->order_valid && whole) ? h->d_order : nullptr);
  if (order)
    hipLaunchKernelGGL((pqp_prologue_kernel<256>), dim3((unsigned)count), dim3(256), lds, stream, h->dev, h->range_first,
                       order + first);
  else
    hipLaunchKernelGGL((pqp_prologue_kernel<256>), dim3((unsigned)count), dim3(256), lds, stream, h->dev,
                       h->range_first + first, order);
  HIP_TRY(hipGetLastError());
  return PQP_OK;
}
#endif
#if PQP_TU_HAS(17)
#ifndef PQP_WPS_DENSE_WAVE
#define PQP_WPS_DENSE_WAVE 2
#endif
template<int WPS>
__global__ __launch_bounds__(64, WPS) void
pqp_dwave_kernel(pqp::Batch batch, long first, const int* __restrict__ order)
{
  HIP_DYNAMIC_SHARED(double, smem)
  const long slot = order ? (long)order[blockIdx.x] : (long)blockIdx.x;
  pqp::dwave_solve_body(batch, first + slot, (pqp::lptr)smem);
}

// Two kernels per launch: the 256-thread factorisation prologue, then the one-wavefront iteration kernel.
// (Cutting the launch into 2 - 8 chunks on streams of their own, so that a chunk's prologue runs beside the iteration
// kernel of the chunk before it, changes nothing: 7.67 - 7.93 ms against 7.69 ms per 2048 C2 QPs,
// profiles/r06_ab_dwave.txt -- the pair is bound by the bytes it moves, not by the order it moves them in.)
int
pqp_launch_dense_wave(pqp_batch* h)
{
  HIP_TRY(hipEventRecord(h->ev0, h->stream));
  if (int rc = pqp_launch_prologue(h, 0, h->range_count, h->stream))
    return rc;
  HIP_TRY(hipEventRecord(h->ev_mid, h->stream));
  h->prologue_timed = true;
  const bool whole = h->range_first == 0 && h->range_count == h->dev.B;
  const int* order = h->subset_order ? h->subset_order : ((h->lpt && h->order_valid && whole) ? h->d_order : nullptr);
  hipLaunchKernelGGL((pqp_dwave_kernel<PQP_WPS_DENSE_WAVE>), dim3((unsigned)h->range_count), dim3(64), pqp::dwave_lds_bytes(),
                     h->stream, h->dev, h->range_first, order);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(h->ev1, h->stream));
  return PQP_OK;
}
#endif
#if PQP_TU_HAS(3)
int
pqp_launch_solve_512(pqp_batch* h, bool common)
{
  return common ? launch_solve<512, PQP_WPS_512, 1>(h) : launch_solve<512, PQP_WPS_512, 0>(h);
}
#endif
#if PQP_TU_HAS(8)
int
pqp_launch_solve_512_dense(pqp_batch* h, bool common)
{
  return common ? launch_solve<512, PQP_WPS_512_DENSE, 1>(h) : launch_solve<512, PQP_WPS_512_DENSE, 0>(h);
}
#endif
#if PQP_TU_HAS(4)
int
pqp_launch_solve_1024(pqp_batch* h, bool common)
{
  return common ? launch_solve<1024, PQP_WPS_1024, 1>(h) : launch_solve<1024, PQP_WPS_1024, 0>(h);
}
#endif

#if PQP_TU == 9 || (PQP_TU == 0 && defined(PQP_EMULATED_MFMA))
// (the single-translation-unit build exists for the CPU emulator only, where address spaces are plain
// pointers: there the kernel below is the ordinary solver running on a heap slice)
template<int NT, int WPS, int SPEC>
__global__ __launch_bounds__(NT, WPS) void
pqp_solve_hbm_kernel(pqp::Batch batch, long first, const int* __restrict__ order, double* scratch, long stride)
{
  const long slot = order ? (long)order[blockIdx.x] : (long)blockIdx.x;
  pqp::solve_body<NT, SPEC>(batch, first + slot, (pqp::lptr)(scratch + (long)blockIdx.x * stride));
}

int
pqp_launch_solve_hbm(pqp_batch* h, bool common)
{
  HIP_TRY(hipEventRecord(h->ev0, h->stream));
  const bool whole = h->range_first == 0 && h->range_count == h->dev.B;
  const int* order = h->subset_order ? h->subset_order : ((h->lpt && h->order_valid && whole) ? h->d_order : nullptr);
  const long stride = (long)((h->lds_solve + 7) / 8);
  if (common)
    hipLaunchKernelGGL((pqp_solve_hbm_kernel<1024, PQP_WPS_1024, 1>), dim3((unsigned)h->range_count), dim3(1024), 0,
                       h->stream, h->dev, h->range_first, order, h->vec_scratch, stride);
  else
    hipLaunchKernelGGL((pqp_solve_hbm_kernel<1024, PQP_WPS_1024, 0>), dim3((unsigned)h->range_count), dim3(1024), 0,
                       h->stream, h->dev, h->range_first, order, h->vec_scratch, stride);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(h->ev1, h->stream));
  return PQP_OK;
}
#endif

#if PQP_TU_HAS(5)
template<int NT>
__global__ __launch_bounds__(NT, 2) void
pqp_backward_kernel(pqp::Batch batch, pqp::BackwardArgs bw)
{
  HIP_DYNAMIC_SHARED(double, smem)
  pqp::backward_body<NT>(batch, bw, (long)blockIdx.x, (pqp::lptr)smem);
}

template<int NT>
static int
launch_backward(pqp_batch* h, const pqp::BackwardArgs& bw, long count)
{
  if (h->lds_solve > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&pqp_backward_kernel<NT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_solve));
  hipLaunchKernelGGL((pqp_backward_kernel<NT>), dim3((unsigned)count), dim3(NT), h->lds_solve, h->stream,
                     h->dev, bw);
  HIP_TRY(hipGetLastError());
  return PQP_OK;
}

int
pqp_launch_backward(pqp_batch* h, const pqp::BackwardArgs& bw, long count)
{
  switch (h->nt) {
    case 256:
      return launch_backward<256>(h, bw, count);
    case 512:
      return launch_backward<512>(h, bw, count);
    default:
      return launch_backward<1024>(h, bw, count);
  }
}
#endif

#if PQP_TU_HAS(6)
template<int NT>
__global__ __launch_bounds__(NT) void
pqp_setup_kernel(pqp::Batch batch, long first)
{
  HIP_DYNAMIC_SHARED(double, smem)
  pqp::setup_body<NT>(batch, first + (long)blockIdx.x, (pqp::lptr)smem);
}

// Dispatch order for the next whole-batch launch: QP i goes to position
// rank(i) = #{ j : cycles_j > cycles_i  or  (cycles_j == cycles_i and j < i) }  (descending by the
// device cycles of the solve that just finished; O(B^2) compares, a few microseconds for B ~ 10^3-10^4).
__global__ __launch_bounds__(64) void
pqp_order_kernel(const long long* __restrict__ stats, int stride, int B, int* __restrict__ order)
{
  // keys are compared as 32-bit values (cycle counts are clamped to 2^32 - 1: an ordering
  // heuristic, exactness of huge counts does not matter)
  constexpr int TILE = 4096;
  __shared__ unsigned tile[TILE];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const long long raw = (i < B) ? stats[(long)i * stride] : 0;
  const unsigned ci = raw > 0xffffffffll ? 0xffffffffu : (raw < 0 ? 0u : (unsigned)raw);
  int rank = 0;
  for (int j0 = 0; j0 < B; j0 += TILE) {
    const int cnt = (B - j0 < TILE) ? (B - j0) : TILE;
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += blockDim.x) {
      const long long r = stats[(long)(j0 + t) * stride];
      tile[t] = r > 0xffffffffll ? 0xffffffffu : (r < 0 ? 0u : (unsigned)r);
    }
    __syncthreads();
    const int split = (i - j0 < 0) ? 0 : ((i - j0 < cnt) ? (i - j0) : cnt); // j < i  <=>  t < split
    for (int t = 0; t < split; ++t)
      rank += (tile[t] >= ci) ? 1 : 0;
    for (int t = split; t < cnt; ++t)
      rank += (tile[t] > ci) ? 1 : 0;
  }
  if (i < B)
    order[rank] = i;
}

template<int NT>
static int
launch_setup(pqp_batch* h)
{
  if (h->lds_setup > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&pqp_setup_kernel<NT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_setup));
  // only the QPs [setup_first, setup_first + setup_count) carry a queued command
  hipLaunchKernelGGL((pqp_setup_kernel<NT>), dim3((unsigned)h->setup_count), dim3(NT), h->lds_setup, h->stream,
                     h->dev, h->setup_first);
  HIP_TRY(hipGetLastError());
  return PQP_OK;
}

int
pqp_launch_setup(pqp_batch* h)
{
  switch (h->nt) {
    case 256:
      return launch_setup<256>(h);
    case 512:
      return launch_setup<512>(h);
    default:
      return launch_setup<1024>(h);
  }
}

int
pqp_launch_order(pqp_batch* h, long count)
{
  hipLaunchKernelGGL((pqp_order_kernel), dim3((unsigned)((count + 63) / 64)), dim3(64), 0, h->stream,
                     reinterpret_cast<const long long*>(h->dev.stats), (int)pqp::ST_COUNT, (int)count, h->d_order);
  HIP_TRY(hipGetLastError());
  return PQP_OK;
}

// (x, y, z, status, iter) of the QPs first .. first+count-1 packed into one row-major
// [count][n + n_eq + n_c + 2] fp64 buffer: the payload of the path's only collective (the final
// all_gather of a sharded batch), built on the device so that the gather never touches the host.
__global__ __launch_bounds__(256) void
pqp_pack_kernel(pqp::Batch batch, long first, long count, double* __restrict__ out)
{
  const int n = batch.d.n, ne = batch.d.n_eq, nc = batch.d.nc;
  const long width = (long)n + ne + nc + 2;
  const long total = count * width;
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    const long i = o / width;
    const int k = (int)(o - i * width);
    const long q = first + i;
    double v;
    if (k < n)
      v = batch.x[q * n + k];
    else if (k < n + ne)
      v = batch.y[q * ne + (k - n)];
    else if (k < n + ne + nc)
      v = batch.z[q * nc + (k - n - ne)];
    else if (k == n + ne + nc)
      v = (double)batch.info[q].status;
    else
      v = (double)batch.info[q].iter;
    out[o] = v;
  }
}

int
pqp_launch_pack(pqp_batch* h, long first, long count, double* out, hipStream_t stream)
{
  const long width = (long)h->dev.d.n + h->dev.d.n_eq + h->dev.d.nc + 2;
  long blocks = (count * width + 255) / 256;
  if (blocks > 4096)
    blocks = 4096;
  if (blocks < 1)
    blocks = 1;
  hipLaunchKernelGGL((pqp_pack_kernel), dim3((unsigned)blocks), dim3(256), 0, stream, h->dev, first, count, out);
  HIP_TRY(hipGetLastError());
  return PQP_OK;
}

// 0: the workgroup kernels; 1: every QP OF THE LAUNCH (its range or its subset; whole_batch: of the handle) has diagonal
// structure (signature + the flags the set-up kernel left) and the launch goes to the one-wavefront kernel of pqp_diag.hpp;
// 2: the same, forced to the 256-thread form of that solver by PQP_DIAG_KERNEL=workgroup (its A/B partner in the tests;
// read per launch: they switch it between two solves).  (a general QP or a never-initialised slot elsewhere in
// the handle no longer sends a launch of structured QPs to the 6.5 x slower kernel.)
int
pqp_diag_dispatch(const pqp_batch* h, bool whole_batch)
{
  const pqp::Dims& dd = h->dev.d;
  if (h->nt != 256 || h->vec_scratch || !pqp::diag_structure_signature(dd.hessian, dd.n_eq, dd.n_in, dd.box) || h->c_diag.empty())
    return 0;
  if (!whole_batch && h->subset_host) {
    for (int q : *h->subset_host)
      if (!h->c_diag[size_t(q)])
        return 0;
  } else {
    const size_t lo = whole_batch ? 0 : size_t(h->range_first);
    const size_t hi = whole_batch ? h->c_diag.size() : std::min(h->c_diag.size(), size_t(h->range_first + h->range_count));
    for (size_t q = lo; q < hi; ++q)
      if (!h->c_diag[q])
        return 0;
  }
  const char* e = std::getenv("PQP_DIAG_KERNEL");
  return (e && e[0] == 'w') || dd.n > 256 ? 2 : 1;
}

// 0: the workgroup kernels; 1: the one-wavefront dense kernel (pqp_dwave.hpp) behind the factorisation prologue.
// PQP_DENSE_KERNEL=workgroup / wave forces either (the A/B partners of the tests; read per launch).  By default the
// one-wavefront form takes the launches made of full resident rounds (see below).
size_t
pqp_dense_wave_lds_bytes()
{
  return pqp::dwave_lds_bytes();
}

int
pqp_dense_wave_dispatch(const pqp_batch* h, long count)
{
  if (h->nt != 256 || h->vec_scratch || !pqp::dwave_signature(h->dev.d))
    return 0;
  const char* e = std::getenv("PQP_DENSE_KERNEL");
  if (e && e[0] == 'w' && e[1] == 'o') // "workgroup"
    return 0;
  if (e && e[0] == 'w' && e[1] == 'a') // "wave"
    return 1;
  // Measured on C2-shaped batches, both kernels at 22 batch sizes (profiles/r06_ab_dwave.txt sections 5 and 17): the
  // one-wavefront kernel keeps 8 QPs per CU resident; it wins from 0.6 of a resident round upwards (1280 QPs on 256 CUs: +7 %;
  // 2048: +10 %) and in later rounds whenever the last one is empty or at least 0.4 full (3072: +6 %, 4096: +5 %); a thinner
  // last round is run at a lone wavefront's latency and ties or loses 1 - 2 % (2304, 2560), and below 0.6 of a round the
  // workgroup kernel's four wavefronts per QP win outright (1024 QPs: 4.4 against 5.2 ms).  Five other shapes of the
  // signature behave the same way (section 10).
  const long round = 8L * h->n_cu, rem = count % round;
  if (count < 64) // (whatever the device: a handful of QPs is a latency problem -- and the one-CU emulated device of the CPU tests
    return 0;     //  dispatches small launches as a real one does)
  if (count < round)
    return (5 * count >= 3 * round) ? 1 : 0;
  return (rem == 0 || 5 * rem >= 2 * round) ? 1 : 0;
}

int
pqp_launch_solve(pqp_batch* h)
{
  // SPEC = 1: no box constraints, dense Hessian, PrimalDualLDLT engine -- all known at compile time
  const bool common = h->dev.d.box == 0 && h->dev.d.hessian == PQP_HESSIAN_DENSE &&
                      h->dev.d.backend != PQP_BACKEND_PRIMAL_LDLT;
  h->prologue_timed = false;
  if (h->vec_scratch) // per-QP vectors beyond the LDS of a CU: the solver runs on an HBM slice per workgroup
    return pqp_launch_solve_hbm(h, common);
  switch (h->nt) {
    case 256:
      if (!common) {
        if (const int dg = pqp_diag_dispatch(h))
          return dg == 1 ? pqp_launch_solve_diag_wave(h) : pqp_launch_solve_256_s2(h);
        return (h->range_count <= (long)h->n_cu) ? pqp_launch_solve_256_s0_one(h) : pqp_launch_solve_256_s0(h);
      }
      if (pqp_dense_wave_dispatch(h, h->range_count))
        return pqp_launch_dense_wave(h);
      // more workgroups than three per CU can hold at once: the four-per-CU build; otherwise the
      // launch is latency-bound and the build with the larger register budget is faster per QP
      if (h->range_count > 3L * h->n_cu && 4 * h->lds_solve <= 160 * 1024)
        return pqp_launch_solve_256_s1(h);
      if (h->range_count <= (long)h->n_cu)
        return pqp_launch_solve_256_s1_one(h); // a CU per QP: the whole register file
      if (h->range_count <= 2L * h->n_cu)
        return pqp_launch_solve_256_s1_two(h);
      return pqp_launch_solve_256_s1_lat(h);
    case 512:
      // (same rule as for 256 threads: the smaller register budget only when it buys a second resident workgroup)
      return (h->range_count > (long)h->n_cu && 2 * h->lds_solve <= 160 * 1024) ? pqp_launch_solve_512_dense(h, common)
                                                                               : pqp_launch_solve_512(h, common);
    default:
      return pqp_launch_solve_1024(h, common);
  }
}
#endif
