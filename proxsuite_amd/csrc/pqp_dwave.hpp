// The DENSE solver as ONE WAVEFRONT PER QP (north_star: "one wavefront owns one QP instance"):
// dense Hessian, no box constraints, PrimalDualLDLT engine (the signature of BASELINE.json configs[0..2]), every
// dimension n, n_eq, n_in <= 128.
//
// The workgroup kernel pqp_solve_kernel<256, 4, 1> spends a Newton step on 256 threads meeting at ~45 barriers with the
// vectors in 40.9 KB of LDS: four QPs per CU, 2048 QPs in two ragged rounds, 0.42 of the HBM roofline, waves parked 63 %
// of their cycles (profiles/r05_pmc_c2.json).  Here
//   * a workgroup is ONE wavefront and holds every per-QP vector of the Newton step in REGISTERS, element k of a vector
//     in lane (k % 128) / 2, register k % 2 of block k / 128 ("pair layout": a lane owns two adjacent elements, so a
//     row of a row-major matrix is one 16-byte load per lane, 800 contiguous bytes per instruction at n = 100);
//   * a mat-vec is a pass over the ROWS of a matrix (mat_pass): the coefficient of row t is broadcast from its owner
//     lane through scalar registers (v_readlane) and the row accumulates into the lane's two columns -- M^T c with
//     no cross-lane reduction, no barrier, no LDS round trip -- and the same loaded registers give the row sums M x:
//     sixteen rows at a time are reduced across the wavefront by the FP64 matrix core (one v_mfma_f64_16x16x4 per row
//     against a unit selector, one more to close the group), so A x and A^T y (C x and C^T z) come from ONE read of the
//     matrix and the transposed copies A_s^T, C_s^T, Z_c are never touched;
//   * the dual Schur block keeps the inverse-factor form of the workgroup kernels (W_S, D_S; appended / deleted rows
//     by the same recurrences, the prefix sum of a deletion on the wavefront) and is re-factorised by the blocked
//     matrix-core routines of pqp_block.hpp instantiated on one wavefront;
//   * 18.5 KB of LDS per QP (seven rarely read vectors, slot lists, 8 KB of scratch): EIGHT QPs per CU, two wavefronts
//     per SIMD at 256 VGPRs -- all 2048 QPs of C2 resident in one round.
// The once-per-solve, GEMM-shaped prologue (re-applied equilibration, H_s + rho I = L D L^T, W = L^{-1}, Z, G) stays a
// 256-thread kernel of its own in front (Solver::prologue, pqp_prologue_kernel).
// Same algorithm, same decisions, same HBM state as the workgroup kernels (a QP may be solved by either, in any order).
// Sums are taken in another order, so results agree with them and with the oracle to rounding, not bit for bit.
#ifndef PQP_DWAVE_HPP
#define PQP_DWAVE_HPP

#include "pqp_diag.hpp"
#include <utility>

namespace pqp {

#ifndef PQP_DW_HB
#define PQP_DW_HB 2
#endif
#ifndef PQP_DW_SCHUR_LAZY
#define PQP_DW_SCHUR_LAZY 1 // register factorisation of the Schur block: left-looking, S gathered one block row ahead (0: all at once)
#endif
// (5 / 6 / 7 / 8 blocks: 8.07 / 7.85 / 8.15 / 7.66 ms per 2048 C2 QPs on one box, profiles/r06_ab_dwave.txt section 11: with eight
// every block of the signature (r <= 128) is factorised in registers and the memory-resident form drops out of the kernel)
#ifndef PQP_DW_SCHUR_REG_BLOCKS
#define PQP_DW_SCHUR_REG_BLOCKS 8
#endif
constexpr int DW_SCHUR_REG_BLOCKS = PQP_DW_SCHUR_REG_BLOCKS; // dual blocks of up to 16 x this many slots are factorised in registers
constexpr int DW_MAXDIM = 128; // n, n_eq, n_in <= 128 (one register block each); slots n_eq + n_in <= 256 (two blocks)

// vectors kept in LDS (read at most a few times per Newton step), 128 doubles each, linear by element
enum
{
  DLV_XP = 0,
  DLV_YP,
  DLV_ZP,
  DLV_GS,
  DLV_BS,
  DLV_US,
  DLV_LS,
  DLV_COUNT
};
constexpr int DW_SCR = 1024; // doubles of scratch: row sums / permutation staging; D_S + `top` of the blocked factorisation
constexpr int DW_INTS = 128 + 128 + 256 + 256 + 2 * INCR_MAX;

__host__ __device__ inline size_t
dwave_lds_bytes()
{
  return (size_t)DW_INTS * sizeof(int) + (size_t)(DLV_COUNT * 128 + DW_SCR) * sizeof(double) + (ST_COUNT + 2) * sizeof(long long);
}

__host__ __device__ inline bool
dwave_signature(const Dims& d)
{
  return d.box == 0 && d.hessian == PQP_HESSIAN_DENSE && d.backend != PQP_BACKEND_PRIMAL_LDLT && d.n <= DW_MAXDIM &&
         d.n_eq <= DW_MAXDIM && d.n_in <= DW_MAXDIM && d.n >= 2;
}

struct DPair
{
  double x, y;
};
// two adjacent doubles (8-byte aligned address: rows of odd length start on odd elements)
__device__ __forceinline__ DPair
dw_load_pair(cgptr p)
{
#ifndef PQP_EMULATED_MFMA
  typedef double pqp_d2u __attribute__((ext_vector_type(2), aligned(8)));
  const pqp_d2u t = *reinterpret_cast<const PQP_GLOBAL pqp_d2u*>(p);
  return DPair{ t.x, t.y };
#else
  return DPair{ p[0], p[1] };
#endif
}

// Columns col, col + 1 (col even, per lane) of a matrix row whose used columns are [clo, chi): a BUFFER load through a
// descriptor that ends at column chi -- elements at or beyond it come back as zeros without a memory access (raw buffer
// range checking, per dword); LOW: lanes wholly left of clo are sent out of range as well (what they would read are the
// stored zeros of a triangular factor).  No branch, no exec masking.
template<bool LOW>
__device__ __forceinline__ DPair
dw_load_row(cgptr base, int off8, int nrec8, int col, int clo)
{
#ifndef PQP_EMULATED_MFMA
  // one descriptor base per pass (the matrix); the row travels in the scalar offset, which the range check includes
  // (scripts/probe/buf_soffset.hip): extent = row offset + used bytes.  Both come out of per-pass vectors by v_readlane:
  // three instructions per row (two broadcasts and the load), no scalar arithmetic
  typedef unsigned pqp_u4 __attribute__((ext_vector_type(4)));
  typedef double pqp_d2v __attribute__((ext_vector_type(2)));
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, nrec8, 0x00020000);
  int off = col * 8;
  if (LOW)
    off = (col + 1 >= clo) ? off : 0x7ffffff0;
  const pqp_u4 raw = __builtin_amdgcn_raw_buffer_load_b128(rs, off, off8, 0);
  const pqp_d2v t = __builtin_bit_cast(pqp_d2v, raw);
  return DPair{ t.x, t.y };
#else
  // (the buffer's range check as the hardware applies it: the byte offset from the base -- row offset + column -- against
  // the extent; an item past the end of a pass has extent 0 and whatever row offset its stale descriptor holds: nothing is
  // read.  AddressSanitizer on the CPU build found the first form of this emulation reading in front of the matrix there.)
  DPair r{ 0.0, 0.0 };
  auto in_range = [&](int c) {
    const long o = (long)off8 + 8L * c;
    return o >= 0 && o + 8 <= (long)nrec8;
  };
  if (!LOW || col + 1 >= clo) {
    if (in_range(col))
      r.x = base[((long)off8 + 8L * col) / 8];
    if (in_range(col + 1))
      r.y = base[((long)off8 + 8L * (col + 1)) / 8];
  }
  return r;
#endif
}

// nothing is scheduled across this point: the loads of a batch stay in front of their uses (the machine scheduler
// otherwise interleaves them to save registers and five loads are in flight instead of sixteen)
__device__ __forceinline__ void
dw_sched_fence()
{
#ifndef PQP_EMULATED_MFMA
  __builtin_amdgcn_sched_barrier(0);
#endif
}

// value of lane `src` (uniform) of an int in every lane
__device__ __forceinline__ int
wave_bcast_i(int v, int src)
{
#ifndef PQP_EMULATED_MFMA
  return __builtin_amdgcn_readlane(v, src);
#else
  return __shfl(v, src);
#endif
}

#define DW_S(s) _Pragma("unroll") for (int s = 0; s < 2; ++s)
#define DW_B(b) _Pragma("unroll") for (int b = 0; b < 2; ++b)

struct DWave
{
  const Batch& batch;
  const long q;
  const Dims d;
  const QpRef P;
  const pqp_settings& st;
  UInfo info;
  const int lane;
  const int n, ne, ni, nd;
  PQP_LDS int* slot_of; // 128: inequality id -> slot among the inequality slots (-1: none)
  PQP_LDS int* actl;    // 128: inequality slot -> inequality id (a hole keeps a valid id)
  PQP_LDS int* rowid;   // 256: dual slot a -> row of Zr / G (equalities: a; inequality slot s: n_eq + id)
  PQP_LDS int* sid;     // 256: scratch (slot -> row of G or -1 for the gather of a factorisation; change lists' ranks)
  PQP_LDS int* chg;     // 2 * INCR_MAX
  lptr lds_v;           // DLV_COUNT x 128
  lptr scr;             // DW_SCR
  PQP_LDS long long* lds_stat;
  UD ruiz_c, dual_feasibility_rhs_2;
  int n_c, n_slots, r;
  bool schur_dirty, schur_incremental, aty_fresh, iterate_zero, nonfinite;
  // n-class
  double x[2], dx[2], dres[2], Hdx[2], ATdy[2], CTdz[2], CTzin[2], rx[2], ex[2], dF[2];
  // n_eq-class
  double y[2], dy[2], se[2], Adx[2];
  // n_in-class
  double z[2], dz[2], Cdx[2], si[2], rup[2];
  int fl[2]; // bit 0 active_set_up, bit 1 active_set_low, bit 2 wanted active, bit 3 in the factor (has a slot)
  // slot-class (dual block, n_eq + inequality slots)
  double sd[2][2], rd[2][2], ed[2][2], dS[2][2];

  __device__ __forceinline__ DWave(const Batch& b, long q_, lptr lds)
    : batch(b)
    , q(uni(q_))
    , d(b.d)
    , P(b, uni(q_))
    , st(b.settings[uni(q_)])
    , lane((int)(threadIdx.x & (WAVE - 1)))
    , n(uni(b.d.n))
    , ne(uni(b.d.n_eq))
    , ni(uni(b.d.n_in))
    , nd(uni(b.d.nd))
  {
    PQP_LDS int* li = (PQP_LDS int*)lds;
    slot_of = li;
    actl = li + 128;
    rowid = li + 256;
    sid = li + 512;
    chg = li + 768;
    lds_v = (lptr)(li + DW_INTS);
    scr = lds_v + DLV_COUNT * 128;
    lds_stat = (PQP_LDS long long*)(scr + DW_SCR);
    n_c = 0;
    n_slots = 0;
    r = ne;
    schur_dirty = true;
    schur_incremental = false;
    aty_fresh = false;
    iterate_zero = false;
    nonfinite = false;
  }
  __device__ __forceinline__ int idx(int s) const { return 2 * lane + s; }
  __device__ __forceinline__ int didx(int b, int s) const { return 128 * b + 2 * lane + s; }
  __device__ __forceinline__ bool active(int s) const { return (fl[s] & 8) != 0; }
  __device__ __forceinline__ double lv(int V, int s) const { return lds_v[V * 128 + idx(s)]; }
  __device__ __forceinline__ void lv_set(int V, int s, double v) { lds_v[V * 128 + idx(s)] = v; }

  // ---- statistics (instrumented build only; see Solver::tic / toc)
  __device__ __forceinline__ void tic()
  {
#ifdef PQP_STATS
    if (lane == 0)
      lds_stat[ST_COUNT] = clock64();
#endif
  }
  __device__ __forceinline__ void toc(int which)
  {
#ifdef PQP_STATS
    if (lane == 0) {
      long long t = clock64();
      lds_stat[which] += t - lds_stat[ST_COUNT];
      lds_stat[ST_COUNT] = t;
    }
#else
    (void)which;
#endif
  }
  __device__ __forceinline__ void sub_tic(int which)
  {
#ifdef PQP_STATS
    if (lane == 0)
      lds_stat[which] -= clock64();
#else
    (void)which;
#endif
  }
  __device__ __forceinline__ void sub_toc(int which)
  {
#ifdef PQP_STATS
    if (lane == 0)
      lds_stat[which] += clock64();
#else
    (void)which;
#endif
  }
  __device__ __forceinline__ void count(int which, long long v = 1)
  {
#ifdef PQP_STATS
    if (lane == 0)
      lds_stat[which] += v;
#else
    (void)which;
    (void)v;
#endif
  }
  __device__ __forceinline__ void bytes(long long b) { count(ST_BYTES_ENGINE, b); }
  __device__ __forceinline__ void trace_line(double kind, double i, double a, double b, double c, double d4, double e)
  {
    if (batch.trace == nullptr || lane != 0)
      return;
    const int slot = batch.trace_slot[q];
    if (slot < 0)
      return;
    gptr t = (gptr)(batch.trace + (long)slot * batch.trace_cap * 8);
    const int k = (int)t[0];
    if (k + 1 >= batch.trace_cap) {
      t[1] += 1.0;
      return;
    }
    gptr rr = t + (long)(k + 1) * 8;
    rr[0] = kind;
    rr[1] = i;
    rr[2] = a;
    rr[3] = b;
    rr[4] = c;
    rr[5] = d4;
    rr[6] = e;
    t[0] = double(k + 1);
  }

  // ---- vectors of one register block (len <= 128)
  __device__ __forceinline__ void vload(double (&v)[2], cgptr src, int len, double fill = 0.0)
  {
    DW_S(s) v[s] = (idx(s) < len) ? src[idx(s)] : fill;
  }
  __device__ __forceinline__ void vstore(gptr dst, const double (&v)[2], int len)
  {
    DW_S(s) if (idx(s) < len) dst[idx(s)] = v[s];
  }
  __device__ __forceinline__ void vzero(double (&v)[2]) { DW_S(s) v[s] = 0.0; }
  __device__ __forceinline__ void vcopy(double (&a)[2], const double (&b)[2]) { DW_S(s) a[s] = b[s]; }
  __device__ __forceinline__ void dzero(double (&v)[2][2])
  {
    DW_B(b) DW_S(s) v[b][s] = 0.0;
  }
  __device__ __forceinline__ void lv_load(int V, cgptr src, int len, double fill = 0.0)
  {
    DW_S(s) lv_set(V, s, (idx(s) < len) ? src[idx(s)] : fill);
  }
  // a block-0 vector by its own index -> LDS (linear), and back
  __device__ __forceinline__ void to_lds(lptr dst, const double (&v)[2]) const
  {
    DW_S(s) dst[idx(s)] = v[s];
  }
  __device__ __forceinline__ void to_lds2(lptr dst, const double (&v)[2][2]) const
  {
    DW_B(b) DW_S(s) dst[didx(b, s)] = v[b][s];
  }

  // -------------------------------------------------------------------------------------------------------------
  // One pass over rows of a row-major matrix.  Item t in [t0, t1) is row `rowptr(t)` (wave-uniform pointer) of which the
  // columns [lo, hi) = colrange(t) are used (the rest are structural zeros: not loaded).
  //   COLS: cacc[cb][s] += sum_t cvec_t * row_t[128 cb + 2 lane + s]      (M^T c; cvec in pair layout, block KBLK)
  //   ROWS: rout[t]      = sum_k row_t[k] * xop_k                          (M x; xop in pair layout; rout in LDS)
  // NCB = column blocks of 128 the rows span (1: rows of up to 128 doubles, 2: up to 256).
  // Item t lives in lane (t % 128) / 2, register t % 2 of coefficient block t / 128: the loop walks lane pairs, the
  // register index is static.  Sixteen items form one group: their loads are issued in two batches of eight, the
  // sixteen per-lane partial dots of a group are reduced across the wavefront on the FP64 matrix core --
  //   T[i][j] += sum_k p_j(16 k + i) * [j == jsel]     one v_mfma_f64_16x16x4 per row (A = the partials, B = a unit column)
  //   rowsum_j = sum_i T[i][j]                          in-lane adds of the four result registers + one more MFMA against ones
  // -------------------------------------------------------------------------------------------------------------
  // The rows of a pass are described by two int vectors in pair layout (Rows1 / Rows2 below): the byte offset of item t's
  // row from the matrix base and the byte extent its descriptor ends at (offset + used bytes; 0 for an item past the end
  // of the pass: nothing is fetched) -- contiguous rows, triangular rows and listed rows alike.
  // Loads are BUFFER loads: lanes beyond the extent get zeros without a memory access and without a branch (dw_load_row).
  // LOW: row t's first used column is t (an upper triangular factor): lanes wholly left of it are pushed out of range too.
  // cvec must be zero beyond the length of the pass.
  template<bool COLS, bool ROWS, int NCB, bool LOW>
  __device__ __forceinline__ void mat_pass_block(cgptr base, int kb, int t1, const int (&offv)[2], const int (&nrecv)[2],
                                                 const double (&cvec)[2], const double (&xop)[NCB][2], double (&cacc)[NCB][2],
                                                 lptr rout)
  {
    const int lo_t = 128 * kb;
    const int hi_t = (t1 < 128 * kb + 128) ? t1 : 128 * kb + 128;
    if (lo_t >= hi_t)
      return;
    const int lr = lane & 15, lk = lane >> 4;
    double acc2[NCB][2]; // second accumulator set (odd items): two independent FMA chains per column
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      acc2[cb][0] = 0.0;
      acc2[cb][1] = 0.0;
    }
    // eight rows: their loads (an item past the end: the descriptor's extent is zero -- nothing is fetched)
    auto issue = [&](DPair(&v)[8][NCB], int tb) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = tb + u;
        const int so = wave_bcast_i(offv[u & 1], (t & 127) >> 1);
        const int nr = wave_bcast_i(nrecv[u & 1], (t & 127) >> 1);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
          v[u][cb] = dw_load_row<LOW>(base, so, nr, 128 * cb + 2 * lane, t);
      }
    };
    // eight rows: into the column accumulators / their partial dots
    auto consume = [&](const DPair(&v)[8][NCB], int tb, double* p8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (COLS) {
          const int t = tb + u;
          const double c = wave_bcast(cvec[u & 1], (t & 127) >> 1);
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) {
            if (u & 1) {
              acc2[cb][0] = fma(c, v[u][cb].x, acc2[cb][0]);
              acc2[cb][1] = fma(c, v[u][cb].y, acc2[cb][1]);
            } else {
              cacc[cb][0] = fma(c, v[u][cb].x, cacc[cb][0]);
              cacc[cb][1] = fma(c, v[u][cb].y, cacc[cb][1]);
            }
          }
        }
        if (ROWS) {
          double pv = 0.0;
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb)
            pv = fma(v[u][cb].x, xop[cb][0], fma(v[u][cb].y, xop[cb][1], pv));
          p8[u] = pv;
        }
      }
    };
    auto reduce16 = [&](const double* p, int g) {
      pqp_d4 T;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        T[k] = 0.0;
#pragma unroll
      for (int j = 0; j < 16; ++j)
        T = mfma_f64_16x16x4(p[j], (lr == j) ? 1.0 : 0.0, T);
      const double qv = (T[0] + T[1]) + (T[2] + T[3]);
      pqp_d4 T2;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        T2[k] = 0.0;
      T2 = mfma_f64_16x16x4(qv, 1.0, T2);
      if (lr == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int t = g + lk + 4 * k;
          if (t < hi_t)
            rout[t] = T2[k];
        }
      }
    };
    // Sixteen rows (NCB = 1; eight with two column blocks) are loaded together and then consumed: what bounds a pass is
    // the number of rows in flight per wavefront (4 VGPRs each), one memory latency per batch.  (A software pipeline over
    // half batches -- the next eight rows issued while eight are consumed -- keeps FEWER rows in flight on average and was
    // 20 - 29 % slower: 9.5 - 10.2 ms against 7.9 ms per 2048 C2 QPs, profiles/r06_ab_dwave.txt.)
    // (Reducing a batch's row sums BEHIND the loads of the next batch -- the 17 matrix-core instructions of a group are 0.5 us
    // of the 1.7 us a lone wavefront spends per batch on an idle device -- keeps 32 more VGPRs live across the loads: 428
    // more spilled registers, 8.73 against 7.69 ms per 2048 C2 QPs and 4.09 against 3.69 ms per 256, profiles/r06_ab_dwave.txt.)
    constexpr int HB = (NCB == 1) ? PQP_DW_HB : 1; // half batches of eight rows issued together
    constexpr int GR = (HB >= 2) ? 8 * HB : 16;    // rows per trip of the loop (a multiple of the 16-row reduction groups)
    for (int g = lo_t; g < hi_t; g += GR) {
      double p[GR];
#pragma unroll
      for (int h0 = 0; h0 < GR; h0 += 8 * HB) {
        DPair v[HB][8][NCB];
#pragma unroll
        for (int hb = 0; hb < HB; ++hb)
          issue(v[hb], g + h0 + 8 * hb);
        dw_sched_fence(); // every load of the batch is issued before the first use of one
#pragma unroll
        for (int hb = 0; hb < HB; ++hb)
          consume(v[hb], g + h0 + 8 * hb, p + h0 + 8 * hb);
      }
      if (ROWS) {
#pragma unroll
        for (int k = 0; k < GR; k += 16)
          if (g + k < hi_t)
            reduce16(p + k, g + k);
      }
    }
    if (COLS) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        cacc[cb][0] += acc2[cb][0];
        cacc[cb][1] += acc2[cb][1];
      }
    }
  }
  struct Rows1
  {
    int off[2], nrec[2];
  };
  struct Rows2
  {
    int off[2][2], nrec[2][2];
  };
  // R contiguous rows of `stride` doubles, columns [0, len)
  __device__ __forceinline__ Rows1 rows_full(int stride, int len, int R) const
  {
    Rows1 w;
    DW_S(s)
    {
      w.off[s] = idx(s) * stride * 8;
      w.nrec[s] = (idx(s) < R) ? w.off[s] + len * 8 : 0;
    }
    return w;
  }
  // ... columns [0, t + 1) of row t (a lower triangular factor)
  __device__ __forceinline__ Rows1 rows_tril(int stride, int R) const
  {
    Rows1 w;
    DW_S(s)
    {
      w.off[s] = idx(s) * stride * 8;
      w.nrec[s] = (idx(s) < R) ? w.off[s] + (idx(s) + 1) * 8 : 0;
    }
    return w;
  }
  __device__ __forceinline__ Rows2 rows2_tril(int stride, int R) const
  {
    Rows2 w;
    DW_B(b) DW_S(s)
    {
      w.off[b][s] = didx(b, s) * stride * 8;
      w.nrec[b][s] = (didx(b, s) < R) ? w.off[b][s] + (didx(b, s) + 1) * 8 : 0;
    }
    return w;
  }
  // listed rows: item t is row sel_t
  __device__ __forceinline__ Rows1 rows_list(const int (&sel)[2], int stride, int len, int R) const
  {
    Rows1 w;
    DW_S(s)
    {
      w.off[s] = sel[s] * stride * 8;
      w.nrec[s] = (idx(s) < R) ? w.off[s] + len * 8 : 0;
    }
    return w;
  }
  __device__ __forceinline__ Rows2 rows2_list(const int (&sel)[2][2], int stride, int len, int R) const
  {
    Rows2 w;
    DW_B(b) DW_S(s)
    {
      w.off[b][s] = sel[b][s] * stride * 8;
      w.nrec[b][s] = (didx(b, s) < R) ? w.off[b][s] + len * 8 : 0;
    }
    return w;
  }
  // coefficient vector of up to 256 items (two register blocks)
  template<bool COLS, bool ROWS, int NCB>
  __device__ __forceinline__ void mat_pass2(cgptr base, int t1, const Rows2& w, const double (&cvec)[2][2],
                                            const double (&xop)[NCB][2], double (&cacc)[NCB][2], lptr rout)
  {
    mat_pass_block<COLS, ROWS, NCB, false>(base, 0, t1, w.off[0], w.nrec[0], cvec[0], xop, cacc, rout);
    if (t1 > 128)
      mat_pass_block<COLS, ROWS, NCB, false>(base, 1, t1, w.off[1], w.nrec[1], cvec[1], xop, cacc, rout);
  }
  // coefficient vector of up to 128 items
  template<bool COLS, bool ROWS, int NCB, bool LOW = false>
  __device__ __forceinline__ void mat_pass1(cgptr base, int t1, const Rows1& w, const double (&cvec)[2],
                                            const double (&xop)[NCB][2], double (&cacc)[NCB][2], lptr rout)
  {
    mat_pass_block<COLS, ROWS, NCB, LOW>(base, 0, t1, w.off, w.nrec, cvec, xop, cacc, rout);
  }

  // out = H_s v from the LOWER TRIANGLE of the symmetric H_s alone, one pass: the column sums of the rows' used parts
  // (j <= i) are the contributions of the lower triangle to out_j, their row sums those of its transpose to out_i, and
  // the diagonal is in both.  Half the bytes of a pass over H_s (the kernel is bound by them).
  __device__ __forceinline__ void hess_mv(const double (&v)[2], double (&out)[2])
  {
    cgptr Hs = P.Hs();
    const int nn = n;
    double dg[2];
    DW_S(s) dg[s] = (idx(s) < nn) ? Hs[(unsigned)(idx(s) * (nn + 1))] : 0.0;
    double acc[1][2] = { { 0.0, 0.0 } };
    const double xop[1][2] = { { v[0], v[1] } };
    mat_pass1<true, true, 1>(Hs, nn, rows_tril(nn, nn), v, xop, acc, scr);
    __syncthreads();
    DW_S(s) out[s] = (idx(s) < nn) ? (acc[0][s] + scr[idx(s)]) - dg[s] * v[s] : 0.0;
    __syncthreads();
  }
  // rowout (LDS -> registers, len R) = M x ; colout += M^T c   for a contiguous row-major R x n matrix, one read
  template<bool COLS, bool ROWS>
  __device__ __forceinline__ void dual_pass(cgptr M, int R, const double (&c)[2], const double (&xv)[2], double (&colout)[2],
                                            double (&rowout)[2])
  {
    const int nn = n;
    double acc[1][2] = { { 0.0, 0.0 } };
    const double xop[1][2] = { { xv[0], xv[1] } };
    mat_pass1<COLS, ROWS, 1>(M, R, rows_full(nn, nn, R), c, xop, acc, scr);
    if (COLS) {
      DW_S(s) colout[s] = (idx(s) < nn) ? acc[0][s] : 0.0;
    }
    if (ROWS) {
      __syncthreads();
      DW_S(s) rowout[s] = (idx(s) < R) ? scr[idx(s)] : 0.0;
      __syncthreads();
    }
  }

  // ---- dual Schur block (Solver, section "dual Schur block"): S_J = M_J + G_JJ = L_S D_S L_S^T kept as (W_S = L_S^{-1}, D_S)
  __device__ __forceinline__ bool slot_live(int a) const
  {
    const int k = a - ne;
    if (k < 0)
      return true;
    return slot_of[actl[k]] == k;
  }
  __device__ __forceinline__ void zero_dead(double (&v)[2][2]) const
  {
    DW_B(b) DW_S(s)
    {
      const int a = didx(b, s);
      const bool live = (a < r) && slot_live(a < r ? a : 0);
      if (!live)
        v[b][s] = 0.0;
    }
  }
  // rowid[a] for the current slots
  __device__ __forceinline__ void build_rowid()
  {
    DW_B(b) DW_S(s)
    {
      const int a = didx(b, s);
      if (a < r)
        rowid[a] = (a < ne) ? a : ne + actl[a - ne];
    }
    __syncthreads();
  }

  // LDL^T of the 16 x 16 diagonal tile in `tile` (LDS, row-major; rows / columns beyond the matrix are identity padding):
  // one row per lane (lanes 0..15), pivot rows through scalar registers; D into dSl[k0 ..], 1 / D into dinv, the strict lower
  // part N of L_kk back into `tile`; returns inv(L_kk) = (I - N)(I + N^2)(I + N^4)(I + N^8) in both operand layouts
  __device__ __forceinline__ TilePair diag_tile_factor(int k0, int rr, lptr dSl, lptr tile, lptr dinv)
  {
    const int lr = lane & 15, lk = lane >> 4;
    TilePair Pm;
    const int nb = (rr - k0 < 16) ? (rr - k0) : 16;
    const int rw = lane & 15;
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const double v = tile[rw * 16 + c];
      a[c] = (rw < nb && c < nb) ? v : ((rw == c) ? 1.0 : 0.0); // identity padding
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const double dc = lane_bcast(a[c], c);
      const double l = a[c] / dc;
#pragma unroll
      for (int cp = c + 1; cp < 16; ++cp) {
        const double mcp = lane_bcast(a[c], cp); // A[cp][c] before scaling
        if (rw >= cp)
          a[cp] = fma(-l, mcp, a[cp]);
      }
      if (rw > c)
        a[c] = l;
    }
    __syncthreads();
    if (lane < 16) {
#pragma unroll
      for (int c = 0; c < 16; ++c)
        tile[rw * 16 + c] = (c < rw) ? a[c] : 0.0; // strict lower N of L_kk
      double dr = 1.0;
#pragma unroll
      for (int c = 0; c < 16; ++c)
        if (c == rw)
          dr = a[c];
      dinv[rw] = 1.0 / dr;
      if (rw < nb)
        dSl[k0 + rw] = dr;
    }
    __syncthreads();
    TilePair N;
    bool dg[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = lk + 4 * q, col = lr;
      N.x[q] = (row > col) ? tile[row * 16 + col] : 0.0;
      N.xt[q] = (row < col) ? tile[col * 16 + row] : 0.0;
      dg[q] = (row == col);
      Pm.x[q] = (dg[q] ? 1.0 : 0.0) - N.x[q];
      Pm.xt[q] = (dg[q] ? 1.0 : 0.0) - N.xt[q];
    }
    TilePair Sq = tile_mul(N, N);
#pragma unroll
    for (int rep = 0; rep < 3; ++rep) {
      TilePair T = Sq;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (dg[q]) {
          T.x[q] += 1.0;
          T.xt[q] += 1.0;
        }
      Pm = tile_mul(Pm, T);
      if (rep < 2)
        Sq = tile_mul(Sq, Sq);
    }
    return Pm;
  }

  // Full factorisation of the current slots for r <= 128, ONE pass: the blocked left-looking LDL^T of S = M_J + G_JJ on the
  // matrix cores with the gather of S, the factorisation of the diagonal tile and the block row of W_S = L_S^{-1} fused
  // into the panel step, and every load of a step issued together.  (ldlt_factor_mfma + tri_inverse_mfma_rows of
  // pqp_block.hpp, which spread tiles over the wavefronts of a workgroup, leave a lone wavefront ~150 dependent memory
  // round trips per factorisation: 25 % of this kernel's cycles at C2.)  Panel kb (16 slots), tiles in the MFMA result
  // layout (register q of lane (lr, lk) = element [lk + 4 q][lr]):
  //   U(kb, x) = S(kb, x) - sum_{p < kb} U(p, kb)^T D_p U(p, x)      x = kb .. : S gathered from the Gram cache into the
  //                                                                  accumulators, the history read back from LS
  //   diagonal tile -> LDS, one row per lane, LDL^T by scalar broadcasts; inv(L_kk) by the Neumann product (registers)
  //   U(kb, x) <- D_k^{-1} inv(L_kk) U(kb, x), stored to LS          x > kb
  //   W(kb, j) = -inv(L_kk) sum_{p = j}^{kb - 1} L(kb, p) W(p, j)    j < kb : L(kb, p) = U(p, kb)^T, W(p, j) read back from W_S
  // Leaves W_S (lower, unit diagonal) in HBM, D_S in registers; LS holds the upper factor (scratch).
  __device__ __forceinline__ void factor_schur_fused()
  {
    const int rr = r, ld = nd;
    const int nbk = (rr + 15) >> 4; // <= 8
    const int lr = lane & 15, lk = lane >> 4;
    cgptr G = P.G();
    gptr U = P.LS();
    gptr Wg = P.WS();
    lptr dSl = scr, tile = scr + 256, dinv = scr + 512;
    const double mu_eq = info.mu_eq, mu_in = info.mu_in;
    for (int kb = 0; kb < nbk; ++kb) {
      const int k0 = kb * 16;
      const int kcol = k0 + lr;
      const int kcc = (kcol < rr) ? kcol : rr - 1;
      pqp_d4 acc[8];
      {
        int rs[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = k0 + lk + 4 * q;
          rs[q] = sid[(row < rr) ? row : rr - 1];
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int x = kb + c;
          if (x < nbk) { // uniform
            const int col = 16 * x + lr;
            const int cs = sid[(col < rr) ? col : rr - 1];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int row = k0 + lk + 4 * q;
              const bool live = (rs[q] | cs) >= 0;
              double v = G[(long)(live ? rs[q] : 0) * ld + (live ? cs : 0)];
              if (row == col)
                v = live ? v + ((row < ne) ? mu_eq : mu_in) : 1.0;
              else
                v = live ? v : 0.0;
              acc[c][q] = (row < rr && col < rr) ? v : 0.0;
            }
          }
        }
      }
      for (int p = 0; p < kb; ++p) {
        const int p0 = 16 * p;
        double ap[4], bp[8][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          ap[q] = U[(long)(p0 + 4 * q + lk) * ld + kcc];
#pragma unroll
        for (int c = 1; c < 8; ++c) {
          const int x = kb + c;
          if (x < nbk) {
            const int col = 16 * x + lr;
            const int cc = (col < rr) ? col : rr - 1;
#pragma unroll
            for (int q = 0; q < 4; ++q)
              bp[c][q] = U[(long)(p0 + 4 * q + lk) * ld + cc];
          }
        }
        double an[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          an[q] = -ap[q] * dSl[p0 + 4 * q + lk];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          acc[0] = mfma_f64_16x16x4(an[q], ap[q], acc[0]);
#pragma unroll
        for (int c = 1; c < 8; ++c) {
          if (kb + c < nbk) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              acc[c] = mfma_f64_16x16x4(an[q], bp[c][q], acc[c]);
          }
        }
      }
      // ---- diagonal tile: one row per lane (lanes 0..15), pivot rows through scalar registers
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q)
        tile[(lk + 4 * q) * 16 + lr] = acc[0][q];
      __syncthreads();
      TilePair Pm = diag_tile_factor(k0, rr, dSl, tile, dinv);
      const double dcol = dinv[lr];
      // W(kb, kb) = inv(L_kk)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = lk + 4 * q, col = lr;
        const int gr = k0 + row, gc = k0 + col;
        if (gr < rr && gc < rr && row >= col)
          Wg[(long)gr * ld + gc] = (row > col) ? Pm.x[q] : 1.0;
      }
      // ---- the panel to the right of the diagonal tile: U(kb, x) <- D_k^{-1} inv(L_kk) U(kb, x)
#pragma unroll
      for (int c = 1; c < 8; ++c) {
        const int x = kb + c;
        if (x < nbk) {
          const int col = 16 * x + lr;
          pqp_d4 res;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            res[q] = 0.0;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            res = mfma_f64_16x16x4(Pm.xt[q] * dcol, (col < rr) ? acc[c][q] : 0.0, res);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (col < rr)
              U[(long)(k0 + lk + 4 * q) * ld + col] = res[q]; // (block kb is full: a block to its right exists)
        }
      }
      // ---- block row kb of W = L^{-1}, left of the diagonal
      if (kb > 0) {
        pqp_d4 T[7];
#pragma unroll
        for (int c = 0; c < 7; ++c)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            T[c][q] = 0.0;
        for (int p = 0; p < kb; ++p) {
          const int p0 = 16 * p;
          double ap[4], wb[7][4];
#pragma unroll
          for (int q = 0; q < 4; ++q)
            ap[q] = U[(long)(p0 + 4 * q + lk) * ld + kcc]; // L[k0 + lr][p0 + 4 q + lk]
#pragma unroll
          for (int c = 0; c < 7; ++c) {
            if (c <= p) { // W(p, j) exists for j <= p (uniform)
#pragma unroll
              for (int q = 0; q < 4; ++q)
                wb[c][q] = Wg[(long)(p0 + 4 * q + lk) * ld + 16 * c + lr];
            }
          }
#pragma unroll
          for (int c = 0; c < 7; ++c) {
            if (c <= p) {
#pragma unroll
              for (int q = 0; q < 4; ++q)
                T[c] = mfma_f64_16x16x4(ap[q], wb[c][q], T[c]);
            }
          }
        }
#pragma unroll
        for (int c = 0; c < 7; ++c) {
          if (c < kb) {
            pqp_d4 Wij;
#pragma unroll
            for (int q = 0; q < 4; ++q)
              Wij[q] = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q)
              Wij = mfma_f64_16x16x4(Pm.xt[q], T[c][q], Wij); // inv(L_kk) * T
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int row = k0 + lk + 4 * q;
              if (row < rr)
                Wg[(long)row * ld + 16 * c + lr] = -Wij[q];
            }
          }
        }
      }
      __syncthreads();
    }
    DW_B(b) DW_S(s)
    {
      const int a = didx(b, s);
      dS[b][s] = (a < rr) ? dSl[a] : 1.0;
    }
    __syncthreads();
    bytes((long)rr * rr * 8 * 2);
  }

  // tiles (i, j >= i) of S = M_J + G_JJ gathered from the Gram cache (sid: slot -> row of G, -1 at a hole)
  template<int NBR, int i>
  __device__ __forceinline__ void schur_reg_gather_row(pqp_d4 (&Ut)[NBR][NBR], int nbk, int rr, int ld, cgptr G, double mu_eq, double mu_in)
  {
    if (i >= NBR || i >= nbk) // uniform
      return;
    constexpr int ii = (i < NBR) ? i : 0;
    const int lr = lane & 15, lk = lane >> 4;
    int rs[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 16 * ii + lk + 4 * q;
      rs[q] = sid[(row < rr) ? row : rr - 1];
    }
#pragma unroll
    for (int j = ii; j < NBR; ++j) {
      if (j < nbk) {
        const int col = 16 * j + lr;
        const int cs = sid[(col < rr) ? col : rr - 1];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = 16 * ii + lk + 4 * q;
          const bool live = (rs[q] | cs) >= 0;
          double v = G[(unsigned)((live ? rs[q] : 0) * ld + (live ? cs : 0))];
          if (row == col)
            v = live ? v + ((row < ne) ? mu_eq : mu_in) : 1.0;
          else
            v = live ? v : 0.0;
          Ut[ii][j][q] = (row < rr && col < rr) ? v : 0.0;
        }
      }
    }
  }

  // one step of factor_schur_reg (the block index is a template parameter: every tile index is static)
  template<int NBR, int kb>
  __device__ __forceinline__ void schur_reg_step(pqp_d4 (&Ut)[NBR][NBR], pqp_d4 (&Wt)[NBR][NBR], int nbk, int rr, int ld, gptr Wg,
                                                 lptr dSl, lptr tile, lptr dinv, cgptr G, double mu_eq, double mu_in)
  {
    if (kb >= nbk) // uniform
      return;
    const int lr = lane & 15, lk = lane >> 4;
    const int k0 = kb * 16;
    if (PQP_DW_SCHUR_LAZY) {
      // LEFT-LOOKING with the gather one block row ahead: the loads of row kb + 1 go out before this step computes, and row
      // kb takes the updates of every earlier step here, in the order the right-looking form applies them (same bits).
      // S tiles of later rows are not live yet: 11 tiles instead of 21 at the first step, and no burst of 84 gathers that
      // serialise behind the spills they cause.
      schur_reg_gather_row<NBR, kb + 1>(Ut, nbk, rr, ld, G, mu_eq, mu_in);
#pragma unroll
      for (int p = 0; p < kb; ++p) {
        double an[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          an[q] = -Ut[p][kb][q] * dSl[16 * p + 4 * q + lk];
#pragma unroll
        for (int j = kb; j < NBR; ++j) {
          if (j < nbk) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              Ut[kb][j] = mfma_f64_16x16x4(an[q], Ut[p][j][q], Ut[kb][j]);
          }
        }
      }
    }
    // ---- diagonal tile: one row per lane (lanes 0..15), pivot rows through scalar registers
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q)
      tile[(lk + 4 * q) * 16 + lr] = Ut[kb][kb][q];
    __syncthreads();
    toc(ST_CYC_F_WRITEBACK); // (sub-phases of the factorisation, instrumented build: the diagonal tile has arrived)
    TilePair Pm = diag_tile_factor(k0, rr, dSl, tile, dinv);
    toc(ST_CYC_S_GATHER); // (... and is factorised)
    const double dcol = dinv[lr];
    // W(kb, kb) = inv(L_kk)
    Wt[kb][kb] = Pm.x;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = lk + 4 * q, col = lr;
      const int gr = k0 + row, gc = k0 + col;
      if (gr < rr && gc < rr && row >= col)
        Wg[(unsigned)(gr * ld + gc)] = (row > col) ? Pm.x[q] : 1.0;
    }
    // ---- the panel to the right of the diagonal tile
#pragma unroll
    for (int c = kb + 1; c < NBR; ++c) {
      if (c < nbk) {
        const int col = 16 * c + lr;
        pqp_d4 res;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          res[q] = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          res = mfma_f64_16x16x4(Pm.xt[q] * dcol, (col < rr) ? Ut[kb][c][q] : 0.0, res);
        Ut[kb][c] = res;
      }
    }
    // ---- block row kb of W = L^{-1}, left of the diagonal
#pragma unroll
    for (int c = 0; c < kb; ++c) {
      pqp_d4 T;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        T[q] = 0.0;
#pragma unroll
      for (int p = c; p < kb; ++p) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          T = mfma_f64_16x16x4(Ut[p][kb][q], Wt[p][c][q], T);
      }
      pqp_d4 Wij;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        Wij[q] = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        Wij = mfma_f64_16x16x4(Pm.xt[q], T[q], Wij); // inv(L_kk) * T
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        Wij[q] = -Wij[q];
        const int row = k0 + lk + 4 * q;
        if (row < rr)
          Wg[(unsigned)(row * ld + 16 * c + lr)] = Wij[q];
      }
      Wt[kb][c] = Wij;
    }
    // ---- trailing tiles (right-looking form)
#pragma unroll
    for (int i = kb + 1; i < NBR; ++i) {
      if (!PQP_DW_SCHUR_LAZY && i < nbk) {
        double an[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          an[q] = -Ut[kb][i][q] * dSl[k0 + 4 * q + lk];
#pragma unroll
        for (int j = i; j < NBR; ++j) {
          if (j < nbk) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              Ut[i][j] = mfma_f64_16x16x4(an[q], Ut[kb][j][q], Ut[i][j]);
          }
        }
      }
    }
  }
  template<int NBR, int... KB>
  __device__ __forceinline__ void schur_reg_steps(pqp_d4 (&Ut)[NBR][NBR], pqp_d4 (&Wt)[NBR][NBR], int nbk, int rr, int ld, gptr Wg,
                                                  lptr dSl, lptr tile, lptr dinv, cgptr G, double mu_eq, double mu_in,
                                                  std::integer_sequence<int, KB...>)
  {
    (schur_reg_step<NBR, KB>(Ut, Wt, nbk, rr, ld, Wg, dSl, tile, dinv, G, mu_eq, mu_in), ...);
  }
  template<int NBR, int... I>
  __device__ __forceinline__ void schur_reg_gather_all(pqp_d4 (&Ut)[NBR][NBR], int nbk, int rr, int ld, cgptr G, double mu_eq,
                                                       double mu_in, std::integer_sequence<int, I...>)
  {
    (schur_reg_gather_row<NBR, I>(Ut, nbk, rr, ld, G, mu_eq, mu_in), ...);
  }

  // The same factorisation with every tile in REGISTERS (r <= 16 NBR): S is gathered once (all loads of the upper block
  // triangle in flight together: one memory round trip), the factor never travels through HBM, W_S is written as it is
  // formed.  A lone wavefront on a saturated device pays ~5 us per DEPENDENT memory round trip, and the form above has
  // 1 + 2 kb of them per panel (gather, history of U, history of W: 25 at r = 80 plus the stores they wait for): 34 % of
  // this kernel's cycles for 5 % of its bytes (profiles/r06_dwave_phases.txt).  Right-looking over statically indexed
  // tiles in the MFMA result layout; tile (i, j) takes its updates in ascending p with the operands of
  // factor_schur_fused, so both forms give the same bits:
  //   step kb:  diagonal tile -> LDS -> L_kk, D_k, inv(L_kk) = Pm                      (as above)
  //             U(kb, c) <- D_k^{-1} inv(L_kk) U(kb, c)                                c > kb
  //             U(i, j) -= U(kb, i)^T D_k U(kb, j)                                     kb < i <= j
  //             W(kb, c) = -inv(L_kk) sum_{p = c}^{kb - 1} U(p, kb)^T W(p, c)           c < kb;  W(kb, kb) = inv(L_kk)
  // Live tiles: rows <= kb of U right of column kb, the trailing triangle, rows <= kb of W: nbk (nbk + 1) / 2 at every step
  // (21 tiles = 84 VGPRs at NBR = 6).
  template<int NBR>
  __device__ __forceinline__ void factor_schur_reg()
  {
    const int rr = r, ld = nd;
    const int nbk = (rr + 15) >> 4; // <= NBR
    const int lr = lane & 15, lk = lane >> 4;
    cgptr G = P.G();
    gptr Wg = P.WS();
    lptr dSl = scr, tile = scr + 256, dinv = scr + 512;
    const double mu_eq = info.mu_eq, mu_in = info.mu_in;
    pqp_d4 Ut[NBR][NBR]; // i <= j
    pqp_d4 Wt[NBR][NBR]; // j <= i
    if (PQP_DW_SCHUR_LAZY)
      schur_reg_gather_row<NBR, 0>(Ut, nbk, rr, ld, G, mu_eq, mu_in); // (row kb + 1 goes out at the top of step kb)
    else
      schur_reg_gather_all<NBR>(Ut, nbk, rr, ld, G, mu_eq, mu_in, std::make_integer_sequence<int, NBR>{});
    toc(ST_CYC_F_LOAD);
    schur_reg_steps<NBR>(Ut, Wt, nbk, rr, ld, Wg, dSl, tile, dinv, G, mu_eq, mu_in, std::make_integer_sequence<int, NBR>{});
    __syncthreads();
    DW_B(b) DW_S(s)
    {
      const int a = didx(b, s);
      dS[b][s] = (a < rr) ? dSl[a] : 1.0;
    }
    __syncthreads();
    bytes((long)rr * rr * 8);
  }

  // full factorisation of the current slots (holes stay identity rows): blocked LDL^T + row-wise inverse on the matrix
  // cores (pqp_block.hpp, one wavefront); leaves W_S in HBM, D_S in registers
  __device__ __forceinline__ void factor_schur()
  {
    const int rr = r;
    DW_B(b) DW_S(s)
    {
      const int a = didx(b, s);
      if (a < rr)
        sid[a] = slot_live(a) ? rowid[a] : -1;
    }
    __syncthreads();
    if (rr <= 128) {
      if (rr <= 16 * DW_SCHUR_REG_BLOCKS)
        factor_schur_reg<DW_SCHUR_REG_BLOCKS>();
      else
        factor_schur_fused();
      toc(ST_CYC_F_UPDATE);
      schur_dirty = false;
      schur_incremental = false;
      count(ST_N_SCHUR_FACT);
      count(ST_FLOPS_FACT, (long long)rr * rr * rr / 3);
      return;
    }
    lptr dSl = scr, top = scr + 256;
    schur_gather_blocked<WAVE>(P.G(), P.LS(), nd, rr, ne, info.mu_eq, info.mu_in, sid);
    toc(ST_CYC_F_LOAD);
    ldlt_factor_mfma<WAVE, true>(P.LS(), nd, rr, dSl, top);
    toc(ST_CYC_F_UPDATE);
    tri_inverse_mfma_rows<WAVE, false>(P.LS(), nd, rr, P.WS(), P.WS());
    toc(ST_CYC_F_WRITEBACK);
    __syncthreads();
    DW_B(b) DW_S(s)
    {
      const int a = didx(b, s);
      dS[b][s] = (a < rr) ? dSl[a] : 1.0;
    }
    __syncthreads();
    bytes((long)rr * rr * 8 * 4);
    schur_dirty = false;
    schur_incremental = false;
    count(ST_N_SCHUR_FACT);
    count(ST_N_SCHUR_BLOCKED);
    count(ST_FLOPS_FACT, (long long)rr * rr * rr / 3);
  }

  // v <- S_J^{-1} v = W^T D^{-1} W v over the r slots (zero at the holes, and it stays zero there)
  __device__ __forceinline__ void schur_apply(double (&v)[2][2])
  {
    const int rr = r, ld = nd;
    cgptr W = P.WS();
    double none[2][2] = { { 0.0, 0.0 }, { 0.0, 0.0 } };
    // t = W v (row sums)
    if (rr > 128)
      mat_pass2<false, true, 2>(W, rr, rows2_tril(ld, rr), none, v, none, scr);
    else {
      const double xop[1][2] = { { v[0][0], v[0][1] } };
      double na[1][2] = { { 0.0, 0.0 } };
      mat_pass1<false, true, 1>(W, rr, rows_tril(ld, rr), none[0], xop, na, scr);
    }
    __syncthreads();
    double t[2][2];
    DW_B(b) DW_S(s)
    {
      const int a = didx(b, s);
      t[b][s] = (a < rr) ? scr[a] / dS[b][s] : 0.0;
    }
    __syncthreads();
    // v = W^T (t / D) (column sums)
    if (rr > 128) {
      double acc[2][2] = { { 0.0, 0.0 }, { 0.0, 0.0 } };
      mat_pass2<true, false, 2>(W, rr, rows2_tril(ld, rr), t, none, acc, scr);
      DW_B(b) DW_S(s) v[b][s] = (didx(b, s) < rr) ? acc[b][s] : 0.0;
    } else {
      double acc[1][2] = { { 0.0, 0.0 } };
      const double xn[1][2] = { { 0.0, 0.0 } };
      mat_pass1<true, false, 1>(W, rr, rows_tril(ld, rr), t[0], xn, acc, scr);
      DW_S(s) v[0][s] = (didx(0, s) < rr) ? acc[0][s] : 0.0;
      DW_S(s) v[1][s] = 0.0;
    }
    bytes((long)rr * (rr + 1) * 8);
  }

  // inclusive prefix sum over the slots in slot order (pair layout, two blocks) of one value per slot
  __device__ __forceinline__ void slot_scan(const double (&e)[2][2], double (&incl)[2][2])
  {
    double base = 0.0;
    DW_B(b)
    {
      const double mine = e[b][0] + e[b][1];
      double sc = mine;
#pragma unroll
      for (int o = 1; o < WAVE; o <<= 1) {
        const double up = __shfl_up(sc, o);
        if (lane >= o)
          sc += up;
      }
      const double excl = base + (sc - mine);
      incl[b][0] = excl + e[b][0];
      incl[b][1] = incl[b][0] + e[b][1];
      base += wave_bcast(sc, WAVE - 1);
    }
  }

  // delete the row / column of inequality slot `s` (uniform) from the factorisation (Solver::schur_delete)
  __device__ __forceinline__ void schur_delete(int sl)
  {
    const int ld = nd, rr = r;
    const int p = ne + sl;
    gptr W = P.WS();
    double wp[2][2], pv[2][2], beta[2][2], e[2][2], incl[2][2];
    DW_B(b) DW_S(s)
    {
      const int c = didx(b, s);
      wp[b][s] = (c < p) ? W[(long)p * ld + c] : 0.0;
      const bool own = (c > p) && (c < rr);
      pv[b][s] = own ? -W[(long)(own ? c : p) * ld + p] : 0.0;
      e[b][s] = own ? pv[b][s] * pv[b][s] / dS[b][s] : 0.0;
    }
    slot_scan(e, incl);
    const int pb = p >> 7, pl = (p & 127) >> 1, ps = p & 1;
    double dp = 0.0;
    DW_B(b) DW_S(s) if (b == pb && s == ps) dp = wave_bcast(dS[b][s], pl);
    const double inv_a0 = 1.0 / dp;
    DW_B(b) DW_S(s)
    {
      const int c = didx(b, s);
      const bool own = (c > p) && (c < rr);
      const double c_i = inv_a0 + incl[b][s], c_im1 = c_i - e[b][s];
      beta[b][s] = own ? pv[b][s] / (dS[b][s] * c_i) : 0.0;
      if (own)
        dS[b][s] = dS[b][s] * (c_i / c_im1);
    }
    // W' = Ltilde^{-1} (W + p w_p^T on the columns left of p):  x_i = y_i - p_i s,  s += beta_i x_i  down each column
    double sacc[2][2] = { { 0.0, 0.0 }, { 0.0, 0.0 } };
    const int nb = (rr > 128) ? 2 : 1;
    for (int i0 = p + 1; i0 < rr; i0 += 8) {
      DPair yv[8][2];
      double pi[8], bi[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = (i0 + u < rr) ? (i0 + u) : (rr - 1);
        const int ib = i >> 7, il = (i & 127) >> 1, is = i & 1;
        double pvv = 0.0, bvv = 0.0;
        DW_B(b) DW_S(s) if (b == ib && s == is)
        {
          pvv = wave_bcast(pv[b][s], il);
          bvv = wave_bcast(beta[b][s], il);
        }
        pi[u] = pvv;
        bi[u] = bvv;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const int col = 128 * cb + 2 * lane;
          yv[u][cb] = (cb < nb && col <= i) ? dw_load_pair(W + (long)i * ld + col) : DPair{ 0.0, 0.0 };
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u;
        if (i < rr) { // uniform
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            if (cb < nb) {
              const int col = 128 * cb + 2 * lane;
              const double x0 = fma(-pi[u], sacc[cb][0], fma(pi[u], wp[cb][0], yv[u][cb].x));
              const double x1 = fma(-pi[u], sacc[cb][1], fma(pi[u], wp[cb][1], yv[u][cb].y));
              sacc[cb][0] = fma(bi[u], x0, sacc[cb][0]);
              sacc[cb][1] = fma(bi[u], x1, sacc[cb][1]);
              if (col != p && col <= i)
                W[(long)i * ld + col] = x0;
              if (col + 1 != p && col + 1 <= i)
                W[(long)i * ld + col + 1] = x1;
            }
          }
        }
      }
    }
    __syncthreads();
    // the slot becomes a hole: identity row and column, unit pivot
    DW_B(b) DW_S(s)
    {
      const int c = didx(b, s);
      if (c < p)
        W[(long)p * ld + c] = 0.0;
      if (c > p && c < rr)
        W[(long)c * ld + p] = 0.0;
      if (c == p)
        dS[b][s] = 1.0;
    }
    bytes((long)(rr - p) * rr * 16);
    count(ST_N_DELETE);
    __syncthreads();
  }

  // append inequality constraint `cid` (uniform) as the new last slot; false: the new pivot is not positive
  __device__ __forceinline__ bool schur_append(int cid)
  {
    const int ld = nd, rr = r;
    const long gid = ne + cid;
    cgptr G = P.G();
    gptr W = P.WS();
    double gv[2][2];
    DW_B(b) DW_S(s)
    {
      const int a = didx(b, s);
      const bool live = (a < rr) && slot_live(a < rr ? a : 0);
      gv[b][s] = live ? G[gid * ld + rowid[a]] : 0.0;
    }
    const double scc = G[gid * ld + gid] + info.mu_in;
    double delta = scc;
    if (rr > 0) {
      double none[2][2] = { { 0.0, 0.0 }, { 0.0, 0.0 } };
      if (rr > 128)
        mat_pass2<false, true, 2>(W, rr, rows2_tril(ld, rr), none, gv, none, scr);
      else {
        const double xop[1][2] = { { gv[0][0], gv[0][1] } };
        double na[1][2] = { { 0.0, 0.0 } };
        mat_pass1<false, true, 1>(W, rr, rows_tril(ld, rr), none[0], xop, na, scr);
      }
      __syncthreads();
      double tv[2][2], acc = 0.0;
      DW_B(b) DW_S(s)
      {
        const int a = didx(b, s);
        const double t = (a < rr) ? scr[a] : 0.0;
        const double td = t / dS[b][s];
        acc = fma(t, td, acc);
        tv[b][s] = td;
      }
      __syncthreads();
      delta = scc - lane_sum(acc);
      double uacc[2][2] = { { 0.0, 0.0 }, { 0.0, 0.0 } };
      if (rr > 128)
        mat_pass2<true, false, 2>(W, rr, rows2_tril(ld, rr), tv, none, uacc, scr);
      else {
        double a1[1][2] = { { 0.0, 0.0 } };
        const double xn[1][2] = { { 0.0, 0.0 } };
        mat_pass1<true, false, 1>(W, rr, rows_tril(ld, rr), tv[0], xn, a1, scr);
        uacc[0][0] = a1[0][0];
        uacc[0][1] = a1[0][1];
      }
      DW_B(b) DW_S(s)
      {
        const int a = didx(b, s);
        if (a < rr)
          W[(long)rr * ld + a] = -uacc[b][s];
      }
    }
    DW_B(b) DW_S(s)
    {
      const int a = didx(b, s);
      if (a == rr) {
        W[(long)rr * ld + rr] = 1.0;
        dS[b][s] = delta;
        actl[rr - ne] = cid;
        slot_of[cid] = rr - ne;
        rowid[rr] = ne + cid;
      }
    }
    n_slots += 1;
    r += 1;
    bytes((long)rr * (rr + 1) * 8 + (long)rr * 16);
    count(ST_N_APPEND);
    __syncthreads();
    return delta > 0.0;
  }

  // rank of inequality (lane, s) among the flagged ones in ascending id order; total in `tot`
  __device__ __forceinline__ void ranks(const bool (&flag)[2], int (&rk)[2], int& tot)
  {
    const unsigned long long m0 = __ballot(flag[0] ? 1 : 0), m1 = __ballot(flag[1] ? 1 : 0);
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (WAVE - lane));
    const int before = __popcll(m0 & below) + __popcll(m1 & below);
    rk[0] = before;
    rk[1] = before + (flag[0] ? 1 : 0);
    tot = uni((int)(__popcll(m0) + __popcll(m1)));
  }

  // New active set from fl bit 2 (Solver::apply_active_set; reference linesearch.hpp:549-786)
  __device__ __forceinline__ void apply_active_set()
  {
    tic();
    bool fadd[2], frm[2];
    DW_S(s)
    {
      const bool want = (fl[s] & 4) != 0, had = (fl[s] & 8) != 0;
      fadd[s] = want && !had;
      frm[s] = !want && had;
    }
    int rka[2], rkr[2], na, nr;
    ranks(fadd, rka, na);
    ranks(frm, rkr, nr);
    if (na + nr == 0 && !schur_dirty) {
      toc(ST_CYC_ZG);
      return;
    }
    const int lim = incr_max(r);
    const bool incremental = !schur_dirty && (na + nr) <= lim && n_slots + na <= ni && (n_slots - n_c) + nr <= lim;
    if (INCR_MAX > 0 && incremental) {
      DW_S(s)
      {
        if (frm[s])
          chg[rkr[s]] = idx(s);
        if (fadd[s])
          chg[INCR_MAX + rka[s]] = idx(s);
      }
      __syncthreads();
      toc(ST_CYC_ZG);
      bool ok = true;
      for (int t = 0; t < nr; ++t) {
        const int i = uni(chg[t]);
        const int sl = uni(slot_of[i]);
        __syncthreads();
        if (lane == 0)
          slot_of[i] = -1;
        DW_S(s) if (idx(s) == i) fl[s] &= ~8;
        __syncthreads();
        schur_delete(sl);
        n_c -= 1;
      }
      for (int t = 0; t < na; ++t) {
        const int i = uni(chg[INCR_MAX + t]);
        ok = schur_append(i) && ok;
        DW_S(s) if (idx(s) == i) fl[s] |= 8;
        n_c += 1;
      }
      schur_incremental = true;
      toc(ST_CYC_SCHUR);
      if (PQP_LIKELY(ok))
        return;
      schur_dirty = true; // a non-positive pivot came out of an append: full factorisation of the set now installed
      tic();
    }
    // the slot map rebuilt in ascending constraint order
    bool want[2];
    int rk[2], total;
    DW_S(s) want[s] = (fl[s] & 4) != 0;
    ranks(want, rk, total);
    __syncthreads();
    DW_S(s)
    {
      if (idx(s) < ni) {
        slot_of[idx(s)] = want[s] ? rk[s] : -1;
        if (want[s])
          actl[rk[s]] = idx(s);
      }
      fl[s] = (fl[s] & 7) | (want[s] ? 8 : 0);
    }
    n_c = total;
    n_slots = total;
    r = ne + n_slots;
    schur_dirty = true;
    __syncthreads();
    build_rowid();
    toc(ST_CYC_ZG);
    if (r > 0)
      factor_schur();
    else
      schur_dirty = false;
    toc(ST_CYC_SCHUR);
  }

  // Solve K [sx; sd] = [bx; bd] in place (Solver::kkt_solve_in_place; reference solver.hpp:320-335)
  __device__ __forceinline__ void kkt_solve_in_place(double (&bx)[2], double (&bd)[2][2])
  {
    const int nn = n, rr = r;
    cgptr WU = P.WU(), WL = P.WL(), Zr = P.Zr();
    double t[2], t2[2];
    {
      // t = L^{-1} bx = sum_k bx_k W[:, k] (row k of WU, columns k .. n-1)
      double acc[1][2] = { { 0.0, 0.0 } };
      const double xn[1][2] = { { 0.0, 0.0 } };
      mat_pass1<true, false, 1, true>(WU, nn, rows_full(nn, nn, nn), bx, xn, acc, scr);
      DW_S(s)
      {
        t[s] = (idx(s) < nn) ? acc[0][s] : 0.0;
        t2[s] = t[s] / dF[s];
      }
    }
    double t1[2];
    if (rr > 0) {
      // s_a = z_a . (t / D) - bd_a over the slots (rows rowid[a] of Zr)
      int rid[2][2];
      DW_B(b) DW_S(s) rid[b][s] = rowid[didx(b, s)];
      const Rows2 zrows = rows2_list(rid, nn, nn, rr);
      {
        const double xop[1][2] = { { t2[0], t2[1] } };
        double na[1][2] = { { 0.0, 0.0 } };
        double none[2][2] = { { 0.0, 0.0 }, { 0.0, 0.0 } };
        mat_pass2<false, true, 1>(Zr, rr, zrows, none, xop, na, scr);
      }
      __syncthreads();
      DW_B(b) DW_S(s)
      {
        const int a = didx(b, s);
        const bool live = (a < rr) && slot_live(a < rr ? a : 0);
        bd[b][s] = live ? scr[a] - bd[b][s] : 0.0;
      }
      __syncthreads();
      toc(ST_CYC_KKT_SOLVE);
      schur_apply(bd);
      toc(ST_CYC_SOLVE_LDLT);
      // t1 = (t - Z_J^T dvec) / D
      double acc[1][2] = { { 0.0, 0.0 } };
      const double xn[1][2] = { { 0.0, 0.0 } };
      mat_pass2<true, false, 1>(Zr, rr, zrows, bd, xn, acc, scr);
      DW_S(s) t1[s] = (idx(s) < nn) ? (t[s] - acc[0][s]) / dF[s] : 0.0;
    } else {
      DW_S(s) t1[s] = t2[s];
    }
    {
      // x = L^{-T} t1 = sum_j t1_j W[j][:] (row j of WL, columns 0 .. j)
      double acc[1][2] = { { 0.0, 0.0 } };
      const double xn[1][2] = { { 0.0, 0.0 } };
      mat_pass1<true, false, 1>(WL, nn, rows_tril(nn, nn), t1, xn, acc, scr);
      DW_S(s) bx[s] = (idx(s) < nn) ? acc[0][s] : 0.0;
    }
    bytes((long)nn * (nn + 1) * 8 + (long)rr * nn * 16);
    count(ST_N_KKT_SOLVES);
  }

  // right-hand side of the step (kept in rx / rd registers)
  // err = rhs - K sol for sol = (dx, sd); by-products Hdx, Adx, ATdy, CTdz (active part), Cdx (Solver::kkt_residual)
  __device__ __forceinline__ double kkt_residual()
  {
    const int nn = n;
    const double rho = info.rho;
    hess_mv(dx, Hdx);
    if (ne > 0) {
      const double c[2] = { sd[0][0], sd[0][1] }; // (equalities are the first slots: n_eq <= 128 = block 0)
      double cm[2];
      DW_S(s) cm[s] = (idx(s) < ne) ? c[s] : 0.0;
      dual_pass<true, true>(P.As(), ne, cm, dx, ATdy, Adx);
    } else {
      vzero(ATdy);
      vzero(Adx);
    }
    if (ni > 0) {
      // C dx over all rows; C^T dz over the rows that have a slot, in the same pass: the coefficient of an
      // inactive row is zero (its multiplier's part is CTzin)
      __syncthreads();
      to_lds2(scr + 512, sd);
      __syncthreads();
      double zf[2];
      DW_S(s) zf[s] = (idx(s) < ni && active(s)) ? scr[512 + ne + slot_of[idx(s)]] : 0.0;
      __syncthreads();
      dual_pass<true, true>(P.Cs(), ni, zf, dx, CTdz, Cdx);
    } else {
      vzero(CTdz);
      vzero(Cdx);
    }
    double m = 0;
    DW_S(s)
    {
      const double e = (idx(s) < nn) ? rx[s] - rho * dx[s] - Hdx[s] - ATdy[s] - CTdz[s] : 0.0;
      ex[s] = e;
      m = vmax_abs(m, e);
    }
    // dual part by slot: equalities directly, inequalities through LDS
    __syncthreads();
    DW_S(s)
    {
      if (idx(s) < ne)
        scr[idx(s)] = Adx[s];
      if (idx(s) < ni && active(s))
        scr[ne + slot_of[idx(s)]] = Cdx[s];
    }
    __syncthreads();
    DW_B(b) DW_S(s)
    {
      const int a = didx(b, s);
      const bool live = (a < r) && slot_live(a < r ? a : 0);
      double e = 0.0;
      if (live)
        e = (a < ne) ? rd[b][s] - scr[a] + sd[b][s] * info.mu_eq : rd[b][s] - (scr[a] - sd[b][s] * info.mu_in);
      ed[b][s] = e;
      m = vmax_abs(m, e);
    }
    __syncthreads();
    bytes(((long)nn * (nn + 1) / 2 + (long)ne * nn + (long)ni * nn) * 8);
    return lane_max0(m);
  }

  // reference solver.hpp:406-541 (Solver::iterative_solve).  In: rx, rd.  Out: dx, sd.
  __device__ __forceinline__ bool iterative_solve(double eps)
  {
    vzero(dx);
    dzero(sd);
    vcopy(ex, rx);
    DW_B(b) DW_S(s) ed[b][s] = rd[b][s];
    long it = 0, it_stability = 0;
    UD preverr = 0, cur = 0;
    while (true) {
      tic();
      kkt_solve_in_place(ex, ed);
      DW_S(s) dx[s] += ex[s];
      DW_B(b) DW_S(s) sd[b][s] += ed[b][s];
      toc(ST_CYC_KKT_SOLVE);
      cur = kkt_residual();
      toc(ST_CYC_RESIDUAL);
      ++it;
      if (it > 1) {
        if (cur > preverr)
          it_stability += 1;
        else
          it_stability = 0;
        if (it_stability == 2)
          break;
      }
      preverr = cur;
      if (!(cur >= eps))
        break;
      if (it >= st.nb_iterative_refinement)
        break;
    }
    info.iterative_residual = cur;
    return (cur >= fmax(eps, st.eps_refact)) && schur_incremental && r > 0;
  }

  // mode 0: semismooth Newton step (reference solver.hpp:754-869); 1: equality-constrained initial guess
  // (helpers.hpp:199-228); 2: install the active set in fl only (solver.hpp:1231-1240)   (Solver::linear_step)
  __device__ __forceinline__ bool linear_step(int mode, double eps)
  {
    const double zfac = (st.merit_function_type == PQP_MERIT_GPDAL) ? st.alpha_gpdal : 1.0;
    if (mode == 0) {
      DW_S(s) if (idx(s) < ni)
      {
        const int up = rup[s] >= 0 ? 1 : 0;
        const int lo = si[s] <= 0 ? 2 : 0;
        fl[s] = (fl[s] & 8) | up | lo | ((up | lo) ? 4 : 0);
      }
    }
    apply_active_set();
    if (mode == 2)
      return false;
    tic();
    if (mode == 0) {
      // C^T z over the INACTIVE rows that carry a multiplier (typically a handful): rows compacted into a list
      double zin[2];
      bool f[2];
      DW_S(s)
      {
        zin[s] = (idx(s) < ni && !active(s)) ? z[s] : 0.0;
        f[s] = zin[s] != 0.0;
      }
      int rk[2], listed;
      ranks(f, rk, listed);
      vzero(CTzin);
      if (listed > 0) {
        __syncthreads();
        DW_S(s) if (f[s])
        {
          sid[rk[s]] = idx(s);
          scr[rk[s]] = zin[s];
        }
        __syncthreads();
        double cl[2];
        DW_S(s) cl[s] = (idx(s) < listed) ? scr[idx(s)] : 0.0;
        cgptr Cs = P.Cs();
        const int nn = n;
        double acc[1][2] = { { 0.0, 0.0 } };
        const double xn[1][2] = { { 0.0, 0.0 } };
        int rs[2];
        DW_S(s) rs[s] = (idx(s) < listed) ? sid[idx(s)] : 0;
        mat_pass1<true, false, 1>(Cs, listed, rows_list(rs, nn, nn, listed), cl, xn, acc, scr + 512);
        DW_S(s) CTzin[s] = (idx(s) < nn) ? acc[0][s] : 0.0;
        __syncthreads();
        bytes((long)listed * nn * 8);
      }
      DW_S(s) rx[s] = -dres[s] + CTzin[s];
      // dual right-hand side by slot
      __syncthreads();
      DW_S(s)
      {
        if (idx(s) < ne)
          scr[idx(s)] = -se[s];
        if (idx(s) < ni && active(s)) {
          double v = 0;
          if (fl[s] & 1)
            v = -rup[s] + z[s] * info.mu_in * zfac;
          else if (fl[s] & 2)
            v = -si[s] + z[s] * info.mu_in * zfac;
          scr[ne + slot_of[idx(s)]] = v;
        }
      }
      __syncthreads();
      DW_B(b) DW_S(s)
      {
        const int a = didx(b, s);
        const bool live = (a < r) && slot_live(a < r ? a : 0);
        rd[b][s] = live ? scr[a] : 0.0;
      }
      __syncthreads();
    } else {
      DW_S(s) rx[s] = -lv(DLV_GS, s);
      DW_B(b) DW_S(s)
      {
        const int a = didx(b, s);
        rd[b][s] = (b == 0 && a < ne) ? lds_v[DLV_BS * 128 + a] : 0.0;
      }
    }
    toc(ST_CYC_NEWTON_MISC);
    const bool missed = iterative_solve(eps);
    if (mode == 1) {
      vcopy(x, dx);
      DW_S(s) y[s] = (idx(s) < ne) ? sd[0][s] : 0.0;
      return false;
    }
    if (PQP_UNLIKELY(missed))
      return true;
    // un-permute: dz_i = solution of its slot, or -z_i when inactive (:860-868)
    DW_S(s) dy[s] = (idx(s) < ne) ? sd[0][s] : 0.0;
    __syncthreads();
    to_lds2(scr, sd);
    __syncthreads();
    DW_S(s)
    {
      if (idx(s) < ni)
        dz[s] = active(s) ? scr[ne + slot_of[idx(s)]] : -z[s];
      else
        dz[s] = 0.0;
      CTdz[s] -= CTzin[s];
    }
    __syncthreads();
    return false;
  }

  // ---- exact line search (reference linesearch.hpp:320-538), as in DiagSolver: every lane sums the terms of its own
  // constraints, one wavefront reduction per sum
  // MAG: also the sum of the MAGNITUDES of the terms of the b-sum at that step length (they cancel; the a-sum is a sum of
  // squares): what the sign of phi' is trusted against in the bracket.  (An alpha-independent bound, as in the workgroup
  // kernels' ls_bracket, is useless with one-sided constraints: a bound of -1e20 puts 1e20 |C dx| into it although its
  // term only enters the sum beyond a breakpoint of that size.)
  template<int NP, bool MAG = false>
  __device__ __forceinline__ void ls_terms(const double (&al)[NP], double (&a_in)[NP], double (&b_in)[NP], double* mag_in = nullptr)
  {
    const bool gpdal = st.merit_function_type == PQP_MERIT_GPDAL;
    double sa[NP], sb[NP], sa2[NP], sb2[NP], sm[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p)
      sa[p] = sb[p] = sa2[p] = sb2[p] = sm[p] = 0.0;
    DW_S(c)
    {
      const double cdx = Cdx[c], up0 = rup[c], lo0 = si[c];
      const double dzi = dz[c] * info.mu_in, zi = z[c] * info.mu_in;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const bool up = (up0 + cdx * al[p]) > 0.;
        const bool lw = (lo0 + cdx * al[p]) < 0.;
        const double e = (up || lw) ? cdx : 0.0;
        const double apz = (up ? up0 : 0.0) + (lw ? lo0 : 0.0);
        sa[p] = fma(e, e, sa[p]);
        sb[p] = fma(apz, e, sb[p]);
        if (MAG)
          sm[p] = fma(fabs(apz), fabs(e), sm[p]);
        if (!gpdal) {
          const double e2 = e - dzi, apz2 = apz - zi;
          sa2[p] = fma(e2, e2, sa2[p]);
          sb2[p] = fma(e2, apz2, sb2[p]);
          if (MAG)
            sm[p] = fma(double(info.nu) * fabs(e2), fabs(apz2), sm[p]);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      sa[p] = lane_sum(sa[p]);
      sb[p] = lane_sum(sb[p]);
      if (MAG)
        mag_in[p] = (gpdal ? info.mu_in_inv * lane_sum(sm[p]) / st.alpha_gpdal : info.mu_in_inv * lane_sum(sm[p]));
      if (gpdal) {
        a_in[p] = info.mu_in_inv * sa[p] / st.alpha_gpdal;
        b_in[p] = info.mu_in_inv * sb[p] / st.alpha_gpdal;
      } else {
        sa2[p] = lane_sum(sa2[p]);
        sb2[p] = lane_sum(sb2[p]);
        a_in[p] = info.mu_in_inv * sa[p] + info.nu * info.mu_in_inv * sa2[p];
        b_in[p] = info.mu_in_inv * sb[p] + info.nu * info.mu_in_inv * sb2[p];
      }
    }
  }
  __device__ __forceinline__ double ls_grad(double al, double a0, double b0)
  {
    const double a1[1] = { al };
    double ai[1], bi[1];
    ls_terms<1>(a1, ai, bi);
    return (a0 + ai[0]) * al + (b0 + bi[0]);
  }

  static constexpr int NBP = 4; // breakpoints a lane owns: two per constraint

  // (DiagSolver::ls_select)
  __device__ __forceinline__ bool ls_select(const double (&mine)[NBP], const bool (&take)[NBP], double a0, double b0, double pred,
                                            bool all_negative, double amax, bool everything, double& result,
                                            double succ = __builtin_inf())
  {
    const double INF = __builtin_inf();
    double gr[NBP];
#pragma unroll
    for (int k = 0; k < NBP; ++k)
      gr[k] = 0.0;
    int nev = 0;
#pragma unroll
    for (int k = 0; k < NBP; ++k) {
      unsigned long long m = __ballot(take[k] ? 1 : 0);
      while (m != 0ull) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1ull;
        const double al = wave_bcast(mine[k], src);
        const double g = ls_grad(al, a0, b0);
        if (lane == src)
          gr[k] = g;
        ++nev;
      }
    }
    count(ST_N_LS_BREAKPOINTS, nev);
    double afp = INF;
#pragma unroll
    for (int k = 0; k < NBP; ++k)
      if (take[k] && !(gr[k] < 0) && mine[k] < afp)
        afp = mine[k];
    afp = wave_min(afp);
    double g_succ = -INF;
    if (!all_negative && !(afp < INF) && succ < INF) {
      g_succ = ls_grad(succ, a0, b0);
      count(ST_N_LS_BREAKPOINTS, 1);
      if (!(g_succ < 0))
        afp = succ;
    }
    if (all_negative) {
      if (afp < INF)
        return false;
      double a1[1] = { 2 * amax + 1 }, ai[1], bi[1];
      ls_terms<1>(a1, ai, bi);
      result = -(b0 + bi[0]) / (a0 + ai[0]);
      return true;
    }
    double aln = 0.0;
    if (!(afp < INF)) {
      if (!everything)
        return false;
#pragma unroll
      for (int k = 0; k < NBP; ++k)
        if (take[k])
          aln = vmax(aln, mine[k]);
      aln = lane_max0(aln);
      double a1[1] = { 2 * aln + 1 }, ai[1], bi[1];
      ls_terms<1>(a1, ai, bi);
      result = -(b0 + bi[0]) / (a0 + ai[0]);
      return true;
    }
    if (pred > 0.0 && !(afp > pred))
      return false;
    double gfp = -INF;
#pragma unroll
    for (int k = 0; k < NBP; ++k)
      if (take[k]) {
        if (mine[k] == afp && !(gr[k] < 0))
          gfp = vmax(gfp, gr[k]);
        if (mine[k] < afp)
          aln = vmax(aln, mine[k]);
      }
    gfp = wave_max(gfp);
    if (afp == succ && !(g_succ < 0))
      gfp = vmax(gfp, g_succ);
    aln = lane_max0(aln);
    double gln = -INF;
#pragma unroll
    for (int k = 0; k < NBP; ++k)
      if (take[k] && mine[k] == aln)
        gln = vmax(gln, gr[k]);
    gln = wave_max(gln);
    if (aln == 0.0) {
      if (pred > 0.0)
        return false;
      double a1[1] = { 0.0 }, ai[1], bi[1];
      ls_terms<1>(a1, ai, bi);
      gln = b0 + bi[0];
    }
    result = fabs(aln - gln * (afp - aln) / (gfp - gln)); // linesearch.hpp:534-536
    return true;
  }

  __device__ __forceinline__ double primal_dual_ls(double& dw_max)
  {
    const bool gpdal = st.merit_function_type == PQP_MERIT_GPDAL;
    const double INF = __builtin_inf();
    double s_dxHdx = 0, s_adx2 = 0, s_dx2 = 0, s_e2 = 0, s_xHdx = 0, s_errdx = 0, s_adxres = 0, s_eres = 0, s_dz2 = 0, s_dzz = 0,
           dwm = 0;
    DW_S(c)
    {
      const double dxk = dx[c];
      dwm = vmax_abs(dwm, dxk);
      s_dxHdx += dxk * Hdx[c];
      s_dx2 += dxk * dxk;
      s_xHdx += x[c] * Hdx[c];
      s_errdx += (info.rho * (x[c] - lv(DLV_XP, c)) + lv(DLV_GS, c)) * dxk;
    }
    DW_S(c)
    {
      const double adx = Adx[c];
      const double e = adx - dy[c] * info.mu_eq;
      dwm = vmax_abs(dwm, dy[c]);
      s_adx2 += adx * adx;
      s_e2 += e * e;
      s_adxres += adx * (se[c] + y[c] * info.mu_eq);
      s_eres += e * se[c];
    }
    DW_S(c)
    {
      dwm = vmax_abs(dwm, dz[c]);
      s_dz2 += dz[c] * dz[c];
      s_dzz += dz[c] * z[c];
    }
    dw_max = lane_max0(dwm);
    s_dxHdx = lane_sum(s_dxHdx);
    s_dx2 = lane_sum(s_dx2);
    s_xHdx = lane_sum(s_xHdx);
    s_errdx = lane_sum(s_errdx);
    if (ne > 0) {
      s_adx2 = lane_sum(s_adx2);
      s_e2 = lane_sum(s_e2);
      s_adxres = lane_sum(s_adxres);
      s_eres = lane_sum(s_eres);
    }
    s_dz2 = lane_sum(s_dz2);
    s_dzz = lane_sum(s_dzz);
    const double nu = gpdal ? 1.0 : double(info.nu);
    double a0 = s_dxHdx + info.mu_eq_inv * s_adx2 + info.rho * s_dx2 + s_e2 * info.mu_eq_inv * nu;
    double b0 = s_xHdx + s_errdx + info.mu_eq_inv * s_adxres + nu * info.mu_eq_inv * s_eres;
    if (gpdal) {
      a0 += info.mu_in * (1. - st.alpha_gpdal) * s_dz2;
      b0 += info.mu_in * (1. - st.alpha_gpdal) * s_dzz;
    }
    sub_tic(ST_CYC_LS_EVAL);
    double mine[NBP];
    double amax = 0;
    int cnti = 0;
    DW_S(c)
    {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        double al = -1.0;
        if (idx(c) < ni && Cdx[c] != 0.) {
          const double num = h ? si[c] : rup[c];
          al = -num / (Cdx[c] + MACHINE_EPS);
        }
        const bool ok = al > MACHINE_EPS;
        mine[2 * c + h] = ok ? al : -1.0;
        cnti += __popcll(__ballot(ok ? 1 : 0));
        if (ok)
          amax = vmax(amax, al);
      }
    }
    const double cnt = (double)cnti;
    amax = lane_max0(amax);
    double result = 0;
    if (cnt == 0.0) { // linesearch.hpp:405-419
      double a1[1] = { 0.0 }, ai[1], bi[1];
      ls_terms<1>(a1, ai, bi);
      sub_toc(ST_CYC_LS_EVAL);
      return -(b0 + bi[0]) / (a0 + ai[0]);
    }
    constexpr int WCAP = 32;
    bool take[NBP];
    bool done = false;
    if (cnt > 8.0 && amax < INF) {
      // bracket (lo, hi] of the zero of the monotone phi' by semismooth Newton probes / quickselect on the breakpoints
      // (DiagSolver::primal_dual_ls); nothing of it is trusted: ls_select validates it on exact values
      const double SURE = 3.6e-15 * (double)(d.nc + d.n + d.n_eq);
      double lo = 0.0, hi = INF, inside = cnt;
      bool all_negative = false, give_up = false;
      double newt = -1.0;
      {
        const double al[1] = { 0.0 };
        double ai[1], bi[1], mi[1];
        ls_terms<1, true>(al, ai, bi, mi);
        const double g0 = b0 + bi[0], mag0 = fabs(b0) + mi[0];
        if (a0 + ai[0] > 0.)
          newt = -g0 / (a0 + ai[0]);
        if (!(g0 < -SURE * mag0))
          give_up = true;
      }
      if (!give_up && !all_negative) {
        bool newton_ok = true;
        for (int round = 0; round < 40 && (inside > 6.0 || !(hi < INF)); ++round) {
          double pv;
          const bool by_newton = newton_ok && newt > lo && newt < hi;
          if (!(hi < INF) && (!(inside > 6.0) || (by_newton && !(newt < amax)))) {
            pv = amax;
          } else if (by_newton) {
            pv = newt;
          } else {
            double cand = -1.0;
#pragma unroll
            for (int k = NBP - 1; k >= 0; --k)
              if (mine[k] > lo && mine[k] < hi)
                cand = mine[k];
            const unsigned long long m = __ballot(cand > 0 ? 1 : 0);
            if (m == 0ull)
              break;
            const int rank = __popcll(m & ((lane == 0) ? 0ull : (~0ull >> (WAVE - lane))));
            const unsigned long long pick = __ballot((cand > 0 && rank == (__popcll(m) >> 1)) ? 1 : 0);
            pv = wave_bcast(cand, __ffsll((long long)pick) - 1);
          }
          const double a1[1] = { pv };
          double ai[1], bi[1], mi[1];
          ls_terms<1, true>(a1, ai, bi, mi);
          const double slope = a0 + ai[0];
          const double g = slope * pv + (b0 + bi[0]), mag = fabs(slope * pv) + fabs(b0) + mi[0];
          int cbi = 0;
#pragma unroll
          for (int k = 0; k < NBP; ++k)
            cbi += __popcll(__ballot((mine[k] > lo && mine[k] <= pv) ? 1 : 0));
          const double cb = (double)cbi;
          const double before = inside;
          if (g > SURE * mag) {
            hi = pv;
            inside = cb;
          } else if (g < -SURE * mag) {
            lo = pv;
            inside -= cb;
          } else {
            double below = 0, above = -INF;
#pragma unroll
            for (int k = 0; k < NBP; ++k) {
              if (mine[k] < pv)
                below = vmax(below, mine[k]);
              if (mine[k] >= pv)
                above = vmax(above, -mine[k]);
            }
            lo = lane_max0(below);
            hi = -wave_max(above);
            inside = 1.0;
            break;
          }
          newt = slope > 0. ? -(b0 + bi[0]) / slope : -1.0;
          newton_ok = !by_newton || inside <= 0.5 * before;
          if (!(lo < amax)) {
            all_negative = true;
            break;
          }
        }
        if (inside > double(WCAP - 3))
          give_up = true;
      }
      if (!give_up) {
        double pred = 0.0, succ = INF;
        if (!all_negative) {
          double below = 0.0, above = -INF;
#pragma unroll
          for (int k = 0; k < NBP; ++k)
            if (mine[k] > 0) {
              if (mine[k] <= lo)
                below = vmax(below, mine[k]);
              if (mine[k] > hi)
                above = vmax(above, -mine[k]);
            }
          pred = lane_max0(below);
          succ = -wave_max(above);
        }
#pragma unroll
        for (int k = 0; k < NBP; ++k) {
          const double a = mine[k];
          take[k] = all_negative ? (a > 0 && a == amax) : (a > 0 && ((a > lo && a <= hi) || a == pred));
        }
        done = ls_select(mine, take, a0, b0, pred, all_negative, amax, false, result, succ);
      }
    }
    if (!done) {
#pragma unroll
      for (int k = 0; k < NBP; ++k)
        take[k] = mine[k] > 0;
      ls_select(mine, take, a0, b0, 0.0, false, amax, true, result);
    }
    sub_toc(ST_CYC_LS_EVAL);
    return result;
  }

  // both infeasibility certificates + the inner stopping criterion (Solver::saddle_point_and_certificates)
  __device__ __forceinline__ void saddle_point_and_certificates(bool do_cert, double& err_in, bool& primal_infeasible,
                                                                bool& dual_infeasible)
  {
    const double c = ruiz_c;
    const double NEG = -__builtin_inf();
    double lb1 = 0, gdx = 0, nrm_dy = 0, nrm_dz = 0, lb2 = 0, ndx = 0, nadx = 0, nhdx = 0, mviol = NEG, e1 = 0, e2 = 0, e3 = 0;
    {
      const double zf = (st.merit_function_type == PQP_MERIT_GPDAL) ? st.alpha_gpdal : 1.0;
      DW_S(k)
      {
        if (idx(k) < ni) {
          const double up = rup[k], lo = si[k];
          const double v = (up > 0 ? up : 0.0) + (lo < 0 ? lo : 0.0) - zf * z[k] * info.mu_in;
          e1 = vmax_abs(e1, v);
        }
        e2 = vmax_abs(e2, se[k]);
        e3 = vmax_abs(e3, dres[k]);
      }
    }
    if (do_cert) {
      cgptr sxg = P.dlt_x(), seg = P.dlt_eq(), sig = P.dlt_in();
      double sx[2], sq[2], sc_[2];
      vload(sx, sxg, n, 1.0);
      vload(sq, seg, ne, 1.0);
      vload(sc_, sig, ni, 1.0);
      DW_S(k) if (idx(k) < n)
      {
        const double sc = sx[k] * c;
        ATdy[k] /= sc;
        CTdz[k] /= sc;
        lb2 = vmax_abs(lb2, ATdy[k] + CTdz[k]);
        Hdx[k] /= sc;
        nhdx = vmax_abs(nhdx, Hdx[k]);
        gdx += dx[k] * lv(DLV_GS, k);
        dx[k] *= sx[k];
        ndx = vmax_abs(ndx, dx[k]);
      }
      DW_S(k) if (idx(k) < ne)
      {
        lb1 += dy[k] * lv(DLV_BS, k);
        dy[k] = dy[k] * sq[k] / c;
        nrm_dy = vmax_abs(nrm_dy, dy[k]);
        Adx[k] /= sq[k];
        nadx = vmax_abs(nadx, Adx[k]);
      }
      DW_S(k) if (idx(k) < ni)
      {
        const double v = dz[k];
        const double ubk = lv(DLV_US, k), lbk = lv(DLV_LS, k);
        lb1 += (v > 0 ? v : 0.0) * ubk;
        lb1 -= (v < 0 ? v : 0.0) * lbk;
        dz[k] = v * sc_[k] / c;
        nrm_dz = vmax_abs(nrm_dz, dz[k]);
        const double w = Cdx[k] / sc_[k];
        Cdx[k] = w;
        const double val = (ubk <= 1.E20 && lbk >= -1.E20) ? fabs(w) : ((ubk > 1.E20) ? -w : w);
        mviol = vmax(mviol, val);
      }
    }
    lb1 = lane_sum(lb1);
    gdx = lane_sum(gdx);
    nrm_dy = lane_max0(nrm_dy);
    nrm_dz = lane_max0(nrm_dz);
    lb2 = lane_max0(lb2);
    ndx = lane_max0(ndx);
    nadx = lane_max0(nadx);
    nhdx = lane_max0(nhdx);
    mviol = wave_max(mviol);
    e1 = lane_max0(e1);
    e2 = lane_max0(e2);
    e3 = lane_max0(e3);
    err_in = fmax(e1, fmax(e2, e3));
    primal_infeasible = false;
    dual_infeasible = false;
    if (!do_cert)
      return;
    {
      const double upper_bound = st.eps_primal_inf * fmax(nrm_dy, nrm_dz);
      primal_infeasible = (nrm_dy != 0 || nrm_dz != 0) && lb2 <= upper_bound && lb1 <= -upper_bound;
    }
    {
      double bound = ndx * st.eps_dual_inf;
      const bool first_cond = (nadx <= bound) && !(mviol > bound);
      bound *= c;
      const bool second_cond_alt1 = nhdx <= bound && gdx <= -bound;
      dual_infeasible = first_cond && second_cond_alt1 && ndx != 0;
    }
  }

  // reference solver.hpp:882-1077 (Solver::newton_semi_smooth)
  // mode != 0: the one linear step in front of the first outer iteration (equality-constrained initial guess, or the active
  // set of a warm start installed) through the SAME call site of linear_step -- the step's code exists once in the kernel
  __device__ __forceinline__ void newton_semi_smooth(double eps_int, int mode)
  {
    bool refactorized = false;
    for (long iter = 0; iter <= st.max_iter_in; ++iter) {
      if (mode == 0 && iter == st.max_iter_in) {
        info.iter += st.max_iter_in + 1;
        break;
      }
      if (mode == 0)
        count(ST_N_NEWTON);
      const bool missed = linear_step(mode, eps_int);
      if (mode != 0)
        break;
      if (PQP_UNLIKELY(missed) && !refactorized) {
        schur_dirty = true; // refinement fallback (solver.hpp:474-532): factor rebuilt, solve + refinement repeated once
        refactorized = true;
        count(ST_N_REFACTORIZE);
        --iter;
        continue;
      }
      refactorized = false;
      tic();
      if (st.merit_function_type == PQP_MERIT_GPDAL) {
        DW_S(c) Cdx[c] += (st.alpha_gpdal - 1.) * info.mu_in * dz[c];
      }
      UD alpha = 1.0;
      double dw_max = 0;
      if (ni > 0) {
        alpha = primal_dual_ls(dw_max);
      } else {
        DW_S(c)
        {
          dw_max = vmax_abs(dw_max, dx[c]);
          dw_max = vmax_abs(dw_max, dy[c]);
        }
        dw_max = lane_max0(dw_max);
      }
      toc(ST_CYC_LINESEARCH);
      sub_tic(ST_CYC_UPDATE);
      if (fabs(alpha) * dw_max < 1.E-11 && iter > 0) {
        info.iter += iter + 1;
        sub_toc(ST_CYC_UPDATE);
        break;
      }
      DW_S(c)
      {
        x[c] += alpha * dx[c];
        dres[c] += alpha * (info.rho * dx[c] + Hdx[c] + ATdy[c] + CTdz[c]);
        rup[c] += alpha * Cdx[c];
        si[c] += alpha * Cdx[c];
        z[c] += alpha * dz[c];
        se[c] += alpha * (Adx[c] - info.mu_eq * dy[c]);
        y[c] += alpha * dy[c];
      }
      sub_toc(ST_CYC_UPDATE);
      bool stop = false;
      UD err_in = 0.0;
      {
        sub_tic(ST_CYC_CERT);
        const bool do_cert = iter % st.frequence_infeasibility_check == 0 || st.primal_infeasibility_solving;
        bool is_primal_infeasible, is_dual_infeasible;
        double e;
        saddle_point_and_certificates(do_cert, e, is_primal_infeasible, is_dual_infeasible);
        err_in = e;
        if (PQP_UNLIKELY(st.verbose != 0))
          trace_line(2.0, double(iter + 1), e, alpha, 0.0, 0.0, 0.0);
        sub_toc(ST_CYC_CERT);
        if (PQP_UNLIKELY(is_primal_infeasible)) {
          info.status = PQP_PRIMAL_INFEASIBLE;
          if (!st.primal_infeasibility_solving) {
            info.iter += iter + 1;
            stop = true;
          }
        } else if (PQP_UNLIKELY(is_dual_infeasible)) {
          info.status = PQP_DUAL_INFEASIBLE;
          info.iter += iter + 1;
          stop = true;
        }
      }
      toc(ST_CYC_NEWTON_MISC);
      if (stop)
        break;
      if (err_in <= eps_int) {
        info.iter += iter + 1;
        break;
      }
      if (PQP_UNLIKELY(!(err_in == err_in))) {
        info.iter += iter + 1;
        nonfinite = true;
        break;
      }
    }
  }

  // out = sum_t c_t M[t][:] for the UNSCALED model matrices (closest-feasible mode, objective)
  __device__ __forceinline__ void unscaled_cols(cgptr M, int R, const double (&c)[2], double (&out)[2])
  {
    const int nn = n;
    double acc[1][2] = { { 0.0, 0.0 } };
    const double xn[1][2] = { { 0.0, 0.0 } };
    mat_pass1<true, false, 1>(M, R, rows_full(nn, nn, R), c, xn, acc, scr);
    DW_S(s) out[s] = (idx(s) < nn) ? acc[0][s] : 0.0;
  }

  // reference utils.hpp:164-252 (Solver::global_primal_residual)
  __device__ __forceinline__ void global_primal_residual(UD& lhs, UD& eq_rhs_0, UD& in_rhs_0, UD& eq_lhs, UD& in_lhs)
  {
    double m_eq0 = 0, m_in0 = 0, m_eql = 0, m_inl = 0;
    if (iterate_zero) {
      vzero(se);
      vzero(rup);
      vzero(ATdy);
      vzero(CTdz);
    } else {
      // one pass over A_s and C_s: row sums A x / C x, column sums A^T y / C^T z (kept for global_dual_residual: aty_fresh)
      if (ne > 0)
        dual_pass<true, true>(P.As(), ne, y, x, ATdy, se);
      else {
        vzero(ATdy);
        vzero(se);
      }
      if (ni > 0)
        dual_pass<true, true>(P.Cs(), ni, z, x, CTdz, rup);
      else {
        vzero(CTdz);
        vzero(rup);
      }
      bytes(((long)ne * n + (long)ni * n) * 8);
    }
    aty_fresh = true;
    {
      cgptr de = P.dlt_eq(), bb = P.bvec(), di = P.dlt_in(), uu = P.u(), ll = P.l();
      double dev[2], bv[2], div_[2], uv[2], lwv[2];
      vload(dev, de, ne, 1.0);
      vload(bv, bb, ne);
      vload(div_, di, ni, 1.0);
      vload(uv, uu, ni);
      vload(lwv, ll, ni);
      DW_S(k) if (idx(k) < ne)
      {
        double v = se[k] / dev[k];
        m_eq0 = vmax_abs(m_eq0, v);
        v -= bv[k];
        m_eql = vmax_abs(m_eql, v);
        se[k] = v;
      }
      DW_S(k) if (idx(k) < ni)
      {
        const double v = rup[k] / div_[k];
        rup[k] = v;
        m_in0 = vmax_abs(m_in0, v);
        const double pu = v - uv[k], pl = v - lwv[k];
        const double sv = (pu > 0 ? pu : 0.0) + (pl < 0 ? pl : 0.0);
        si[k] = sv;
        m_inl = vmax_abs(m_inl, sv);
      }
      eq_rhs_0 = lane_max0(m_eq0);
      in_rhs_0 = lane_max0(m_in0);
      eq_lhs = lane_max0(m_eql);
      in_lhs = lane_max0(m_inl);
      lhs = fmax(eq_lhs, in_lhs);
      if (PQP_UNLIKELY(st.primal_infeasibility_solving && info.status == PQP_PRIMAL_INFEASIBLE)) {
        // utils.hpp:241-248 : || A^T se + C^T si ||_inf on the unscaled model
        double t1[2], t2[2];
        vzero(t1);
        vzero(t2);
        if (ne > 0)
          unscaled_cols(P.A(), ne, se, t1);
        if (ni > 0)
          unscaled_cols(P.C(), ni, si, t2);
        double m = 0;
        DW_S(k) if (idx(k) < n) m = vmax_abs(m, t1[k] + t2[k]);
        lhs = lane_max0(m);
      }
      DW_S(k) if (idx(k) < ne) se[k] *= dev[k];
    }
  }

  // reference utils.hpp:437-587 (Solver::global_dual_residual)
  __device__ __forceinline__ void global_dual_residual(UD& lhs, UD& rhs_0, UD& rhs_1, UD& rhs_3, UD& rhs_duality_gap,
                                                       UD& duality_gap)
  {
    const double c = ruiz_c;
    double m0 = 0, m1 = 0, m3 = 0, ml = 0, xHx = 0, gx = 0, by = 0, zu = 0, zl = 0;
    const double ib = 1.3407807929942596e+154; // sqrt(DBL_MAX), helpers/common.hpp:17-25
    double hx[2];
    if (iterate_zero)
      vzero(hx);
    else
      hess_mv(x, hx);
    const bool have_products = aty_fresh;
    bytes(((iterate_zero ? 0L : (long)n * (n + 1) / 2) + (have_products ? 0L : (long)ne * n + (long)ni * n)) * 8);
    double aty[2], ctz[2];
    if (have_products) {
      vcopy(aty, ATdy);
      vcopy(ctz, CTdz);
    } else {
      double dummy[2];
      if (ne > 0)
        dual_pass<true, false>(P.As(), ne, y, x, aty, dummy);
      else
        vzero(aty);
      if (ni > 0)
        dual_pass<true, false>(P.Cs(), ni, z, x, ctz, dummy);
      else
        vzero(ctz);
    }
    {
      double sx[2], gv[2];
      vload(sx, P.dlt_x(), n, 1.0);
      vload(gv, P.g(), n);
      DW_S(k) if (idx(k) < n)
      {
        const double sc = sx[k] * c;
        const double v = hx[k] / sc; // unscaled H x (utils.hpp:469-471)
        m0 = vmax_abs(m0, v);
        const double xu = x[k] * sx[k];
        xHx += v * xu;
        gx += gv[k] * xu;
        m1 = vmax_abs(m1, aty[k] / sc);
        m3 = vmax_abs(m3, ctz[k] / sc);
        const double dr = lv(DLV_GS, k) + hx[k] + aty[k] + ctz[k];
        dres[k] = dr;
        ml = vmax_abs(ml, dr / sc);
      }
    }
    {
      double dev[2], bv[2], div_[2], uv[2], lwv[2];
      vload(dev, P.dlt_eq(), ne, 1.0);
      vload(bv, P.bvec(), ne);
      vload(div_, P.dlt_in(), ni, 1.0);
      vload(uv, P.u(), ni);
      vload(lwv, P.l(), ni);
      DW_S(k) if (idx(k) < ne) by += bv[k] * (y[k] * dev[k] / c);
      DW_S(k) if (idx(k) < ni)
      {
        const double zi = z[k] * div_[k] / c;
        const double uk = uv[k] < ib ? uv[k] : ib;
        const double lk = lwv[k] > -ib ? lwv[k] : -ib;
        if (fl[k] & 1)
          zu += zi * uk;
        if (fl[k] & 2)
          zl += zi * lk;
      }
    }
    gx = lane_sum(gx);
    xHx = lane_sum(xHx);
    by = lane_sum(by);
    zu = lane_sum(zu);
    zl = lane_sum(zl);
    m0 = lane_max0(m0);
    m1 = lane_max0(m1);
    m3 = lane_max0(m3);
    ml = lane_max0(ml);
    rhs_0 = m0;
    rhs_1 = m1;
    rhs_3 = m3;
    lhs = ml;
    duality_gap = gx;
    rhs_duality_gap = fabs(gx);
    duality_gap += xHx;
    rhs_duality_gap = fmax(rhs_duality_gap, fabs(xHx));
    rhs_duality_gap = fmax(rhs_duality_gap, fabs(by));
    duality_gap += by;
    rhs_duality_gap = fmax(rhs_duality_gap, fabs(zu));
    duality_gap += zu;
    rhs_duality_gap = fmax(rhs_duality_gap, fabs(zl));
    duality_gap += zl;
  }

  // ---- reference solver.hpp:1088-1843 (Solver::solve, behind Solver::prologue)
  __device__ __forceinline__ void solve()
  {
    State W = *P.state();
    info.load(*P.info());
    ruiz_c = W.ruiz_c;
    dual_feasibility_rhs_2 = W.dual_feasibility_rhs_2;
#ifdef PQP_STATS
    {
      // (the prologue kernel left its own counters in the QP's slot)
      const PQP_GLOBAL long long* gs0 = P.stats();
      for (int k = lane; k < ST_COUNT + 2; k += WAVE)
        lds_stat[k] = (k < ST_COUNT) ? gs0[k] : 0;
      __syncthreads();
    }
#endif
    const long long cyc0 = clock64();
    const long long wall0 = wall_clock64();
    DW_S(c)
    {
      dx[c] = dres[c] = Hdx[c] = ATdy[c] = CTdz[c] = CTzin[c] = rx[c] = ex[c] = 0.0;
      dy[c] = se[c] = Adx[c] = 0.0;
      dz[c] = Cdx[c] = si[c] = rup[c] = 0.0;
      lv_set(DLV_XP, c, 0.0);
      lv_set(DLV_YP, c, 0.0);
      lv_set(DLV_ZP, c, 0.0);
    }
    dzero(sd);
    dzero(rd);
    dzero(ed);
    DW_B(b) DW_S(s) dS[b][s] = 1.0;
    vload(x, P.x(), n);
    vload(y, P.y(), ne);
    vload(z, P.z(), ni);
    {
      const PQP_GLOBAL int* ga = P.act();
      DW_S(c)
      {
        fl[c] = (idx(c) < ni) ? act_flags(ga[idx(c) < ni ? idx(c) : 0]) : 0;
        slot_of[idx(c)] = -1;
        actl[idx(c)] = 0;
      }
    }
    __syncthreads();
    tic();
    const int ig = st.initial_guess;
    const bool wswpr = (ig == PQP_WARM_START_WITH_PREVIOUS_RESULT);
    const bool dirty = W.dirty != 0;
    bool do_factor, do_scale_ws, do_aset_from_z, do_eq_guess = false, do_restore = false;
    if (dirty) {
      if (ig == PQP_EQUALITY_CONSTRAINED_INITIAL_GUESS || ig == PQP_NO_INITIAL_GUESS) {
        vzero(x); // results.cleanup
        vzero(y);
        vzero(z);
        cold_start(info, st);
      } else if (wswpr) {
        cleanup_statistics(info);
      } else {
        cold_start(info, st);
      }
    }
    if (ig == PQP_EQUALITY_CONSTRAINED_INITIAL_GUESS) {
      do_factor = true;
      do_scale_ws = false;
      do_aset_from_z = false;
      do_eq_guess = true;
    } else if (ig == PQP_NO_INITIAL_GUESS) {
      do_factor = true;
      do_scale_ws = false;
      do_aset_from_z = false;
    } else if (ig == PQP_COLD_START_WITH_PREVIOUS_RESULT || ig == PQP_WARM_START) {
      do_factor = true;
      do_scale_ws = true;
      do_aset_from_z = true;
    } else { // WARM_START_WITH_PREVIOUS_RESULT
      do_scale_ws = true;
      if (!dirty && W.refactorize) {
        do_factor = true;
        do_aset_from_z = true;
      } else if (!W.factor_valid) {
        do_factor = true;
        do_aset_from_z = true;
      } else {
        do_factor = false;
        do_aset_from_z = false;
        do_restore = true;
      }
    }
    // (the re-applied equilibration and the factorisation of the primal block, Z and G: Solver::prologue, the kernel in
    // front of this one, under the same decisions)
    lv_load(DLV_GS, P.gs(), n);
    lv_load(DLV_BS, P.bs(), ne);
    lv_load(DLV_US, P.us(), ni);
    lv_load(DLV_LS, P.ls(), ni);
    if (do_scale_ws) {
      // solver.hpp:1137-1146: the warm start into the equilibrated space
      double sx[2], sq[2], sc_[2];
      vload(sx, P.dlt_x(), n, 1.0);
      vload(sq, P.dlt_eq(), ne, 1.0);
      vload(sc_, P.dlt_in(), ni, 1.0);
      DW_S(k)
      {
        x[k] /= sx[k];
        y[k] = y[k] / sq[k] * ruiz_c;
        z[k] = z[k] / sc_[k] * ruiz_c;
      }
    }
    vload(dF, P.dF(), n, 1.0);
    if (do_factor) {
      n_c = 0;
      n_slots = 0;
      r = ne;
      schur_dirty = true;
    }
    if (do_restore) {
      // WARM_START_WITH_PREVIOUS_RESULT on an unchanged model: the block factorisation the previous solve left in HBM
      n_c = W.n_c;
      n_slots = W.n_slots;
      r = ne + n_slots;
      {
        cgptr dSg = P.dS();
        DW_B(b) DW_S(s)
        {
          const int a = didx(b, s);
          dS[b][s] = (a < r) ? dSg[a] : 1.0;
        }
        const PQP_GLOBAL int* ga = P.act();
        DW_S(c)
        {
          const int j = idx(c);
          if (j < n_slots) {
            const int i = act_cid(ga[j]);
            actl[j] = (i >= 0) ? i : 0;
            if (i >= 0)
              slot_of[i] = j;
          }
        }
        __syncthreads();
        DW_S(c) if (idx(c) < ni && slot_of[idx(c)] >= 0) fl[c] |= 8;
      }
      build_rowid();
      schur_dirty = !(W.ls_valid && W.mu_eq_fact == info.mu_eq && W.mu_in_fact == info.mu_in);
      schur_incremental = W.ls_edited != 0;
    }
    bool pend = false; // a Newton loop (or the linear step in front of the first outer iteration) is due: ONE call site below
    int pend_mode = 0;
    double pend_eps = 1.0;
    if (do_aset_from_z || do_eq_guess) {
      if (do_aset_from_z) {
        DW_S(c) fl[c] = (fl[c] & 11) | ((idx(c) < ni && z[c] != 0) ? 4 : 0);
      } else {
        DW_S(c) fl[c] = (fl[c] & 11); // (the equality-constrained guess works on the empty active set)
      }
      pend = true;
      pend_mode = do_eq_guess ? 1 : 2;
    }

    // BCL state (solver.hpp:1378-1395)
    const UD bcl_eta_ext_init = pow(0.1, st.alpha_bcl);
    UD bcl_eta_ext = bcl_eta_ext_init;
    UD bcl_eta_in = 1;
    const UD eps_in_min = fmin(st.eps_abs, 1.E-9);
    UD primal_feasibility_eq_rhs_0 = 0, primal_feasibility_in_rhs_0 = 0;
    UD dual_feasibility_rhs_0 = 0, dual_feasibility_rhs_1 = 0, dual_feasibility_rhs_3 = 0;
    UD primal_feasibility_lhs = 0, primal_feasibility_eq_lhs = 0, primal_feasibility_in_lhs = 0;
    UD dual_feasibility_lhs = 0;
    UD duality_gap = 0, rhs_duality_gap = 0;
    UD scaled_eps = st.eps_abs;
    UD primal_feasibility_lhs_new = 0, dual_feasibility_lhs_new = 0;
    UD new_bcl_mu_in = 0, new_bcl_mu_eq = 0, new_bcl_mu_in_inv = 0, new_bcl_mu_eq_inv = 0;
    bool is_primal_feasible = false, is_dual_feasible = false;
    long iter = 0;
    int stage = 0;
    bool done = (st.max_iter <= 0);
    bool gpr_fresh = false, gdr_fresh = false;
    aty_fresh = false;
    {
      double mz = 0;
      DW_S(k)
      {
        mz = vmax_abs(mz, x[k]);
        mz = vmax_abs(mz, y[k]);
        mz = vmax_abs(mz, z[k]);
      }
      iterate_zero = lane_max0(mz) == 0.0;
    }
    UD pl_cache = 0, dl_cache = 0;
    toc(ST_CYC_F_PANEL);
    for (;;) {
      if (pend) {
        newton_semi_smooth(pend_eps, pend_mode);
        pend = false;
        tic();
        if (pend_mode != 0) {
          // (the iterate the first residual evaluations see)
          double mz = 0;
          DW_S(k)
          {
            mz = vmax_abs(mz, x[k]);
            mz = vmax_abs(mz, y[k]);
            mz = vmax_abs(mz, z[k]);
          }
          iterate_zero = lane_max0(mz) == 0.0;
        } else {
          iterate_zero = false;
          gpr_fresh = false;
          gdr_fresh = false;
          aty_fresh = false;
          if (PQP_UNLIKELY(nonfinite)) {
            info.status = PQP_MAX_ITER_REACHED;
            break;
          }
          if ((info.status == PQP_PRIMAL_INFEASIBLE && !st.primal_infeasibility_solving) || info.status == PQP_DUAL_INFEASIBLE) {
            vcopy(x, dx); // certificates (solver.hpp:1572-1580)
            vcopy(y, dy);
            vcopy(z, dz);
            break;
          }
          if (PQP_UNLIKELY(scaled_eps == st.eps_abs && st.primal_infeasibility_solving && info.status == PQP_PRIMAL_INFEASIBLE)) {
            // solver.hpp:1581-1595 : || A^T 1 + C^T 1 ||_inf * eps_abs
            double one_e[2], one_i[2], t1[2], t2[2];
            DW_S(k)
            {
              one_e[k] = (idx(k) < ne) ? 1.0 : 0.0;
              one_i[k] = (idx(k) < ni) ? 1.0 : 0.0;
            }
            vzero(t1);
            vzero(t2);
            if (ne > 0)
              unscaled_cols(P.A(), ne, one_e, t1);
            if (ni > 0)
              unscaled_cols(P.C(), ni, one_i, t2);
            double m = 0;
            DW_S(k) if (idx(k) < n) m = vmax_abs(m, t1[k] + t2[k]);
            scaled_eps = lane_max0(m) * st.eps_abs;
          }
          stage = 1;
        }
      }
      if (done)
        break;
      tic();
      if (st.primal_infeasibility_solving)
        gpr_fresh = false;
      UD pl = pl_cache, dl = dl_cache;
      const bool want_primal = (stage != 2);
      const bool want_dual_pre = (stage != 1);
      if (want_primal && !gpr_fresh) {
        global_primal_residual(pl, primal_feasibility_eq_rhs_0, primal_feasibility_in_rhs_0, primal_feasibility_eq_lhs,
                               primal_feasibility_in_lhs);
        pl_cache = pl;
        gpr_fresh = true;
      }
      bool want_dual = want_dual_pre;
      if (stage == 1) {
        primal_feasibility_lhs_new = pl;
        is_primal_feasible = primal_feasibility_lhs_new <=
                             (scaled_eps + st.eps_rel * fmax(primal_feasibility_eq_rhs_0, primal_feasibility_in_rhs_0));
        info.pri_res = primal_feasibility_lhs_new;
        want_dual = is_primal_feasible;
      }
      if (want_dual && !gdr_fresh) {
        global_dual_residual(dl, dual_feasibility_rhs_0, dual_feasibility_rhs_1, dual_feasibility_rhs_3, rhs_duality_gap,
                             duality_gap);
        dl_cache = dl;
        gdr_fresh = true;
      }
      toc(ST_CYC_GLOBAL_RES);
      const UD rhs_dua_rel = st.eps_rel * fmax(fmax(dual_feasibility_rhs_3, dual_feasibility_rhs_0),
                                               fmax(dual_feasibility_rhs_1, dual_feasibility_rhs_2));
      if (stage == 0) {
        primal_feasibility_lhs = pl;
        dual_feasibility_lhs = dl;
        info.pri_res = primal_feasibility_lhs;
        info.dua_res = dual_feasibility_lhs;
        info.duality_gap = duality_gap;
        new_bcl_mu_in = info.mu_in;
        new_bcl_mu_eq = info.mu_eq;
        new_bcl_mu_in_inv = info.mu_in_inv;
        new_bcl_mu_eq_inv = info.mu_eq_inv;
        UD rhs_pri = scaled_eps;
        if (st.eps_rel != 0)
          rhs_pri += st.eps_rel * fmax(primal_feasibility_eq_rhs_0, primal_feasibility_in_rhs_0);
        is_primal_feasible = primal_feasibility_lhs <= rhs_pri;
        UD rhs_dua = st.eps_abs;
        if (st.eps_rel != 0)
          rhs_dua += rhs_dua_rel;
        is_dual_feasible = dual_feasibility_lhs <= rhs_dua;
        if (PQP_UNLIKELY(st.verbose != 0)) {
          // solver.hpp:1469-1510: the reference's report block unscales x, y, z and scales them back
          trace_line(1.0, double(info.iter_ext + 1), info.pri_res, info.dua_res, info.duality_gap, info.mu_in, info.rho);
          double sx[2], sq[2], sc_[2];
          vload(sx, P.dlt_x(), n, 1.0);
          vload(sq, P.dlt_eq(), ne, 1.0);
          vload(sc_, P.dlt_in(), ni, 1.0);
          DW_S(k)
          {
            x[k] = (x[k] * sx[k]) / sx[k];
            y[k] = (y[k] * sq[k] / ruiz_c) / sq[k] * ruiz_c;
            z[k] = (z[k] * sc_[k] / ruiz_c) / sc_[k] * ruiz_c;
          }
        }
        if (is_primal_feasible && is_dual_feasible) {
          if (st.check_duality_gap) {
            if (fabs(info.duality_gap) <= st.eps_duality_gap_abs + st.eps_duality_gap_rel * rhs_duality_gap) {
              info.status = (st.primal_infeasibility_solving && info.status == PQP_PRIMAL_INFEASIBLE)
                              ? PQP_SOLVED_CLOSEST_PRIMAL_FEASIBLE
                              : PQP_SOLVED;
              break;
            }
          } else {
            info.status = PQP_SOLVED;
            break;
          }
        }
        info.iter_ext += 1;
        DW_S(c)
        {
          lv_set(DLV_XP, c, x[c]);
          lv_set(DLV_YP, c, y[c]);
          lv_set(DLV_ZP, c, z[c]);
        }
        // shifted inequality residuals (solver.hpp:1523-1559)
        {
          double sc_[2];
          vload(sc_, P.dlt_in(), ni, 1.0);
          DW_S(i) if (idx(i) < ni)
          {
            double v = rup[i] * sc_[i] + z[i] * info.mu_in;
            if (st.merit_function_type == PQP_MERIT_GPDAL)
              v += (st.alpha_gpdal - 1.) * info.mu_in * z[i];
            rup[i] = v - lv(DLV_US, i);
            si[i] = v - lv(DLV_LS, i);
          }
        }
        toc(ST_CYC_F_PANEL);
        pend = true;
        pend_mode = 0;
        pend_eps = bcl_eta_in;
        continue;
      }
      if (stage == 1) {
        if (is_primal_feasible) {
          dual_feasibility_lhs_new = dl;
          info.dua_res = dual_feasibility_lhs_new;
          info.duality_gap = duality_gap;
          is_dual_feasible = dual_feasibility_lhs_new <= (st.eps_abs + rhs_dua_rel);
          if (is_dual_feasible) {
            bool gap_ok = true;
            if (st.check_duality_gap)
              gap_ok = fabs(info.duality_gap) <= st.eps_duality_gap_abs + st.eps_duality_gap_rel * rhs_duality_gap;
            if (gap_ok)
              info.status = (st.primal_infeasibility_solving && info.status == PQP_PRIMAL_INFEASIBLE)
                              ? PQP_SOLVED_CLOSEST_PRIMAL_FEASIBLE
                              : PQP_SOLVED;
          }
        }
        if (st.bcl_update) { // solver.hpp:564-614
          if (primal_feasibility_lhs_new <= bcl_eta_ext || info.iter > st.safe_guard) {
            bcl_eta_ext *= pow(info.mu_in, st.beta_bcl);
            bcl_eta_in = fmax(bcl_eta_in * info.mu_in, eps_in_min);
          } else {
            DW_S(c)
            {
              y[c] = lv(DLV_YP, c);
              z[c] = lv(DLV_ZP, c);
            }
            gdr_fresh = false; // y, z were reset
            aty_fresh = false;
            new_bcl_mu_in = fmax(info.mu_in * st.mu_update_factor, st.mu_min_in);
            new_bcl_mu_eq = fmax(info.mu_eq * st.mu_update_factor, st.mu_min_eq);
            new_bcl_mu_in_inv = fmin(info.mu_in_inv * st.mu_update_inv_factor, st.mu_max_in_inv);
            new_bcl_mu_eq_inv = fmin(info.mu_eq_inv * st.mu_update_inv_factor, st.mu_max_eq_inv);
            bcl_eta_ext = bcl_eta_ext_init * pow(new_bcl_mu_in, st.alpha_bcl);
            bcl_eta_in = fmax(new_bcl_mu_in, eps_in_min);
          }
        } else { // Martinez, solver.hpp:637-677
          bcl_eta_in = fmax(bcl_eta_in * 0.1, eps_in_min);
          if (!(primal_feasibility_lhs_new <= 0.95 * primal_feasibility_lhs)) {
            new_bcl_mu_in = fmax(info.mu_in * st.mu_update_factor, st.mu_min_in);
            new_bcl_mu_eq = fmax(info.mu_eq * st.mu_update_factor, st.mu_min_eq);
            new_bcl_mu_in_inv = fmin(info.mu_in_inv * st.mu_update_inv_factor, st.mu_max_in_inv);
            new_bcl_mu_eq_inv = fmin(info.mu_eq_inv * st.mu_update_inv_factor, st.mu_max_eq_inv);
          }
        }
        stage = 2;
        continue;
      }
      // stage 2 (solver.hpp:1693-1746)
      dual_feasibility_lhs_new = dl;
      info.dua_res = dual_feasibility_lhs_new;
      info.duality_gap = duality_gap;
      if (primal_feasibility_lhs_new >= primal_feasibility_lhs && dual_feasibility_lhs_new >= dual_feasibility_lhs &&
          info.mu_in <= 1e-5) {
        new_bcl_mu_in = st.cold_reset_mu_in; // cold restart
        new_bcl_mu_eq = st.cold_reset_mu_eq;
        new_bcl_mu_in_inv = st.cold_reset_mu_in_inv;
        new_bcl_mu_eq_inv = st.cold_reset_mu_eq_inv;
      }
      if (info.mu_in != new_bcl_mu_in || info.mu_eq != new_bcl_mu_eq) {
        ++info.mu_updates;
        if (ne + n_c > 0)
          schur_dirty = true; // mu_update (solver.hpp:128-232): a diagonal shift of the Schur block
      }
      info.mu_eq = new_bcl_mu_eq;
      info.mu_in = new_bcl_mu_in;
      info.mu_eq_inv = new_bcl_mu_eq_inv;
      info.mu_in_inv = new_bcl_mu_in_inv;
      stage = 0;
      ++iter;
      if (iter >= st.max_iter)
        done = true;
    }

    // unscale the solution (solver.hpp:1749-1767)
    tic();
    {
      double sx[2], sq[2], sc_[2];
      vload(sx, P.dlt_x(), n, 1.0);
      vload(sq, P.dlt_eq(), ne, 1.0);
      vload(sc_, P.dlt_in(), ni, 1.0);
      DW_S(k)
      {
        x[k] *= sx[k];
        y[k] = y[k] * sq[k] / ruiz_c;
        z[k] = z[k] * sc_[k] / ruiz_c;
      }
      if (st.primal_infeasibility_solving && info.status == PQP_PRIMAL_INFEASIBLE) {
        DW_S(k)
        {
          se[k] /= sq[k];
          si[k] /= sc_[k];
        }
      }
    }
    // objective on the unscaled model (solver.hpp:1771-1780)
    {
      double hx[2], gv[2];
      unscaled_cols(P.H(), n, x, hx);
      bytes((long)n * n * 8);
      vload(gv, P.g(), n);
      double obj = 0;
      DW_S(k) if (idx(k) < n) obj += 0.5 * hx[k] * x[k] + gv[k] * x[k];
      info.objValue = lane_sum(obj);
    }
    // write back
    vstore(P.x(), x, n);
    vstore(P.y(), y, ne);
    vstore(P.z(), z, ni);
    vstore(P.se(), se, ne);
    vstore(P.si(), si, ni);
    {
      gptr dSg = P.dS();
      DW_B(b) DW_S(s)
      {
        const int a = didx(b, s);
        if (a < nd)
          dSg[a] = dS[b][s];
      }
    }
    if (batch.hx) { // host-mapped mirrors (see Batch)
      vstore((gptr)(batch.hx + P.lq() * n), x, n);
      vstore((gptr)(batch.hy + P.lq() * ne), y, ne);
      vstore((gptr)(batch.hz + P.lq() * ni), z, ni);
      vstore((gptr)(batch.hse + P.lq() * ne), se, ne);
      vstore((gptr)(batch.hsi + P.lq() * ni), si, ni);
    }
    {
      PQP_GLOBAL int* ga = P.act();
      DW_S(c)
      {
        const int i = idx(c);
        if (i < ni)
          ga[i] = act_pack((i < n_slots && slot_live(ne + i)) ? actl[i] : -1, fl[c] & 3);
      }
    }
    toc(ST_CYC_F_TINV);
    if (lane == 0) {
      if (st.compute_timings) {
        info.solve_time = (double)(wall_clock64() - wall0) * batch.wall_us_per_tick;
        info.run_time = info.solve_time + info.setup_time;
      }
      info.store(*P.info());
      if (batch.hinfo)
        info.store(batch.hinfo[q]);
      W.dirty = 1;
      W.is_initialized = 1;
      W.n_c = n_c;
      W.n_slots = n_slots;
      W.factor_valid = 1;
      W.ls_valid = schur_dirty ? 0 : 1;
      W.ls_edited = schur_incremental ? 1 : 0;
      W.mu_eq_fact = info.mu_eq;
      W.mu_in_fact = info.mu_in;
      W.rho_fact = info.rho;
      *P.state() = W;
      PQP_GLOBAL long long* gs_ = P.stats();
#ifdef PQP_STATS
      for (int k = 0; k < ST_COUNT; ++k)
        gs_[k] = lds_stat[k];
#else
      for (int k = 0; k < ST_COUNT; ++k)
        gs_[k] = 0;
#endif
      gs_[ST_N_ACTIVE_FINAL] = n_c;
      gs_[ST_CYC_TOTAL] = clock64() - cyc0;
      gs_[ST_WALL_TICKS] = wall_clock64() - wall0;
    }
  }
};

#undef DW_S
#undef DW_B

__device__ __forceinline__ void
dwave_solve_body(const Batch& batch, long q, lptr lds)
{
  DWave S(batch, q, lds);
  S.solve();
}

} // namespace pqp

#endif
