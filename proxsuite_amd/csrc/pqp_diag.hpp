// The diagonal-structure solver as ONE WAVEFRONT PER QP with every per-QP vector in REGISTERS
// (BASELINE.json configs[4]: "4096 box-constrained QPs with diagonal Hessian ... bandwidth-bound, no MFMA";
// north_star: "one wavefront owns one QP instance").
//
// Diagonal structure (Solver::dm, pqp_solver.hpp): H diagonal or zero, no equality, and every inequality row on ONE
// variable -- box constraints, or a C without off-diagonal entries and n_in == dim -- so constraint k sits on variable
// k, L = I, the Gram matrix and the dual Schur block are diagonal, and the whole of qp_solve (reference
// dense/solver.hpp:1088-1843) is element-wise work on vectors of length dim plus reductions and the exact line search.
// The workgroup kernel pqp_solve_kernel<256, ., 2> spends that on 256 threads meeting at ~45 barriers per Newton step
// with the vectors in 62.8 KB of LDS (two QPs per CU; 49 us per Newton step on 200-element vectors: VERDICT r4).  Here:
//   * element k of every vector lives in lane k % 64, register slot k / 64 (E slots: dim <= 64 E); constraint k and
//     variable k share a lane, so the KKT solve, the residuals, the active-set bookkeeping and the iterate update
//     never leave the lane;
//   * reductions are DPP wavefront reductions (pqp_block.hpp's scan without its identity moves) -- no LDS round trip, no
//     barrier anywhere in the kernel (a workgroup IS one wavefront); counts are ballots + population counts on the scalar unit;
//   * the exact line search (reference linesearch.hpp:320-538) brackets the zero of phi' by quickselect on the breakpoints
//     themselves (a pivot of arbitrary rank a round, phi' there with every lane summing its own constraints' terms + one
//     wavefront reduction per sum), evaluates the <= 8 breakpoints that remain the same way, and applies the reference's
//     selections to them: no lane ever walks all constraints in a serial chain;
//   * LDS: 19 KB per QP (the nine vectors that are read at most a few times per Newton step, the slot list of the persistent
//     state) against 62.8 KB: eight QPs per CU instead of two.
// Bound by vector-ALU issue (61 k instructions per QP, 0.58 of the peak: profiles/r05_pmc_c5.json), not by memory or latency.
// Same algorithm, same decisions, same HBM state as the workgroup kernels (a QP may be solved by either, in any order:
// warm starts, the QPLayer backward and pqp_batch_get_schur_factor read what this kernel leaves).  Sums are taken in
// a different order (lane-serial over the E slots, then the wavefront tree), so results agree with the other kernels
// and with the oracle to rounding, not bit for bit: tests/test_gpu_parity.py gates every C5 / C5box QP against the
// oracle (1e-10, equal Info).
#ifndef PQP_DIAG_HPP
#define PQP_DIAG_HPP

#include "pqp_solver.hpp"

namespace pqp {

// bytes of LDS one QP (= one wavefront = one workgroup) of the kernel needs
__host__ __device__ inline size_t
diag_lds_bytes(int E)
{
  // slot list (ints), the nine vectors that are read at most a few times per Newton step (previous iterate x, z of the
  // proximal terms, the two bounds, g_s, diag(H_s), the Ruiz scalings of the variables and of the constraint rows, the
  // unscaled g), statistics.  19.3 KB for E = 4: eight QPs per CU.
  return (size_t)(64 * E) * sizeof(int) + (size_t)(64 * E) * sizeof(double) * 9 + (ST_COUNT + 2) * sizeof(long long);
}

// Sum over the wavefront for THIS kernel (bound by vector-ALU issue: 65 % of the peak, profiles/r05_pmc_c5.json): the DPP
// scan of pqp_block.hpp without the two identity moves per step -- the four steps inside a row read zero where no source
// lane exists (bound_ctrl) instead of keeping a pre-loaded identity; the two cross-row steps keep it.  (ds_swizzle for
// the data movement -- a third of the vector instructions -- was measured: 0.81 -> 0.90 ms, the LDS crossbar's latency
// costs more than the issue slots it frees, even where eight independent reductions run together: 0.84 ms.)
// max(a, b) and max(a, |x|) of values in vector registers as ONE v_max_f64: fmax() canonicalises an operand the compiler
// cannot prove quiet (a DPP move, a loaded value) with a v_max_f64 x, x, x of its own first -- 561 of the 999 v_max_f64 of
// the E = 4 kernel were those.  Same results: the hardware instruction returns the other operand for a NaN, as fmax does.
#ifndef PQP_EMULATED_MFMA
__device__ __forceinline__ double
vmax(double a, double b)
{
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double
vmax_abs(double a, double x)
{
  double r;
  asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(x));
  return r;
}
#else
__device__ __forceinline__ double
vmax(double a, double b)
{
  return fmax(a, b);
}
__device__ __forceinline__ double
vmax_abs(double a, double x)
{
  return fmax(a, fabs(x));
}
#endif
#ifndef PQP_EMULATED_MFMA
template<int CTRL>
__device__ __forceinline__ double
dpp_shift_zero(double v)
{
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double
lane_sum(double v)
{
  v += dpp_shift_zero<0x111>(v); // row_shr:1
  v += dpp_shift_zero<0x112>(v); // row_shr:2
  v += dpp_shift_zero<0x114>(v); // row_shr:4
  v += dpp_shift_zero<0x118>(v); // row_shr:8
  v += dpp_move<0x142, 0xa>(0.0, v); // row_bcast:15 -> rows 1, 3
  v += dpp_move<0x143, 0xc>(0.0, v); // row_bcast:31 -> rows 2, 3
  return readlane_f64(v, 63);
}
// the same for the maximum of NON-NEGATIVE values (norms, counts, step lengths: zero is their identity)
__device__ __forceinline__ double
lane_max0(double v)
{
  v = vmax(v, dpp_shift_zero<0x111>(v));
  v = vmax(v, dpp_shift_zero<0x112>(v));
  v = vmax(v, dpp_shift_zero<0x114>(v));
  v = vmax(v, dpp_shift_zero<0x118>(v));
  v = vmax(v, dpp_move<0x142, 0xa>(0.0, v));
  v = vmax(v, dpp_move<0x143, 0xc>(0.0, v));
  return readlane_f64(v, 63);
}
#else
__device__ __forceinline__ double
lane_sum(double v)
{
  return wave_sum(v);
}
__device__ __forceinline__ double
lane_max0(double v)
{
  return wave_max(v);
}
#endif

// value of lane `src` (uniform) in every lane
__device__ __forceinline__ double
wave_bcast(double v, int src)
{
#ifndef PQP_EMULATED_MFMA
  return readlane_f64(v, src);
#else
  return __shfl(v, src);
#endif
}

#define PQP_E(c) _Pragma("unroll") for (int c = 0; c < E; ++c)
// keeps a value in a vector register across this point (see rhs_d)
#ifndef PQP_EMULATED_MFMA
#define PQP_KEEP(v) asm volatile("" : "+v"(v))
#else
#define PQP_KEEP(v)
#endif

template<int E>
struct DiagSolver
{
  const Batch& batch;
  const long q;
  const Dims d;
  const QpRef P;
  const pqp_settings& st;
  UInfo info;
  const int lane;
  const int n;      // dim
  const bool cform; // bounds as C (n_in == dim, C diagonal)
  const bool boxf;  // bounds as box constraints (n_in == 0)
  const bool hasc;  // nc > 0
  PQP_LDS int* lds_i;          // 64 E ints: slot lists
  PQP_LDS double* lds_d;       // 64 E doubles of scratch
  PQP_LDS double* lds_v;       // LV_COUNT vectors of 64 E doubles (element k of a vector is only ever touched by its own lane)
  enum
  {
    LV_XP = 0,
    LV_ZP,
    LV_UB,
    LV_LB,
    LV_GS,
    LV_HD,
    LV_SX, // delta_x (1 beyond dim)
    LV_SC, // delta_in / delta_box (1 beyond dim)
    LV_GU, // g of the unscaled model
    LV_COUNT
  };
  PQP_LDS long long* lds_stat; // ST_COUNT + 2 (statistics, instrumented build)
  UD ruiz_c, dual_feasibility_rhs_2;
  int n_c;
  bool schur_dirty, aty_fresh, nonfinite;
  // persistent vectors (slot c of lane l = element 64 c + l; zeros beyond dim, dF = dS = 1 there)
  // (xp, zp, the bounds, g_s and diag(H_s) live in LDS: lv(); 27 vectors in registers are 100 more registers than a
  // wavefront sharing its SIMD with another one may hold)
  double x[E], z[E], zd[E], dF[E], dS[E], dres[E], rup[E], si[E];
  // Newton step
  double dx[E], dz[E], Hdx[E], Cdx[E], CTdz[E], ex[E], ed[E], sd[E];
  int fl[E]; // bit 0 active_set_up, bit 1 active_set_low, bit 2 wanted active, bit 3 in the factor (has a slot)

  __device__ __forceinline__ DiagSolver(const Batch& b, long q_, lptr lds)
    : batch(b)
    , q(uni(q_))
    , d(b.d)
    , P(b, uni(q_))
    , st(b.settings[uni(q_)])
    , lane((int)(threadIdx.x & (WAVE - 1)))
    , n(uni(b.d.n))
    , cform(b.d.n_in > 0)
    , boxf(b.d.box != 0)
    , hasc(b.d.nc > 0)
  {
    lds_i = (PQP_LDS int*)lds;
    lds_v = (PQP_LDS double*)(lds + 32 * E);
    lds_d = lds_v; // (scratch of the restore path of the prologue: the LV_XP slot, written for the first time after it)
    lds_stat = (PQP_LDS long long*)(lds + 32 * E + 64 * E * LV_COUNT);
    n_c = 0;
    schur_dirty = true;
    aty_fresh = false;
    nonfinite = false;
  }
  __device__ __forceinline__ int hess() const { return d.hessian == PQP_HESSIAN_ZERO ? (int)PQP_HESSIAN_ZERO : (int)PQP_HESSIAN_DIAGONAL; }
  __device__ __forceinline__ int idx(int c) const { return c * WAVE + lane; }
  __device__ __forceinline__ bool in(int c) const { return idx(c) < n; }
  __device__ __forceinline__ bool active(int c) const { return (fl[c] & 8) != 0; }
  // element (c, lane) of LDS vector V
  __device__ __forceinline__ double lv(int V, int c) const { return lds_v[V * (64 * E) + idx(c)]; }
  __device__ __forceinline__ void lv_set(int V, int c, double v) { lds_v[V * (64 * E) + idx(c)] = v; }
  __device__ __forceinline__ void lv_load(int V, cgptr src, double fill = 0.0)
  {
    PQP_E(c) lv_set(V, c, in(c) ? src[idx(c)] : fill);
  }
  // right-hand side of the linear step in progress (mode 0: Newton step, reference solver.hpp:787-847; 1: equality-constrained
  // initial guess), recomputed from the iterate where it is used instead of held in registers
  __device__ __forceinline__ double ctz_inactive(int c) const { return hasc ? zd[c] * (active(c) ? 0.0 : z[c]) : 0.0; }
  __device__ __forceinline__ double rhs_x(int c, int mode) const
  {
    return mode == 0 ? -dres[c] + ctz_inactive(c) : -lv(LV_GS, c);
  }
  __device__ __forceinline__ double rhs_d(int c, int mode) const
  {
    double v = 0;
    if (mode == 0 && active(c)) {
      const double zfac = (st.merit_function_type == PQP_MERIT_GPDAL) ? st.alpha_gpdal : 1.0;
      // (both candidates are read into registers first: a select between the ADDRESSES of rup[c] and si[c] would put
      // the whole solver object into scratch memory)
      double ru = rup[c], sv = si[c];
      PQP_KEEP(ru);
      PQP_KEEP(sv);
      const double shift = z[c] * info.mu_in * zfac;
      if (fl[c] & 1)
        v = -ru + shift;
      else if (fl[c] & 2)
        v = -sv + shift;
    }
    return v;
  }

  // ---- statistics (see Solver::tic / toc): the instrumented build only; the total cycles of a solve (dispatch order)
  // and the wall ticks (Info timings) are always recorded
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
    gptr r = t + (long)(k + 1) * 8;
    r[0] = kind;
    r[1] = i;
    r[2] = a;
    r[3] = b;
    r[4] = c;
    r[5] = d4;
    r[6] = e;
    t[0] = double(k + 1);
  }

  // ---- loads / stores of one vector of length dim
  __device__ __forceinline__ void vload(double (&v)[E], cgptr src, double fill = 0.0)
  {
    PQP_E(c) v[c] = in(c) ? src[idx(c)] : fill;
  }
  __device__ __forceinline__ void vstore(gptr dst, const double (&v)[E])
  {
    PQP_E(c) if (in(c)) dst[idx(c)] = v[c];
  }
  __device__ __forceinline__ void vzero(double (&v)[E]) { PQP_E(c) v[c] = 0.0; }
  __device__ __forceinline__ void vcopy(double (&a)[E], const double (&b)[E]) { PQP_E(c) a[c] = b[c]; }
  // Ruiz scaling of the constraint rows: delta_in (C form) / delta_box (box form)
  __device__ __forceinline__ cgptr dlt_c() const { return cform ? P.dlt_in() : P.dlt_box(); }

  // rank of element (c, lane) among the elements with `flag`, in ascending element order; total in `tot`
  __device__ __forceinline__ void ranks(const bool (&flag)[E], int (&rk)[E], int& tot)
  {
    int before = 0;
    PQP_E(c)
    {
      const unsigned long long m = __ballot(flag[c] ? 1 : 0);
      rk[c] = before + __popcll(m & ((lane == 0) ? 0ull : (~0ull >> (WAVE - lane))));
      before += __popcll(m);
    }
    tot = uni(before);
  }

  // ---- the equilibrated vectors of a dirty re-solve (Solver::solve -> write_scaled(..., diag_only, matrices) with the
  // stored scaling: reference solver.hpp:1192-1214; u, l unclamped, the box bounds clamped, helpers.hpp:638-649)
  __device__ __forceinline__ void rescale(bool matrices)
  {
    cgptr S = P.delta();
    const double c = ruiz_c;
    double Sx[E], Sc[E];
    vload(Sx, S);
    if (hasc)
      vload(Sc, S + n);
    if (matrices) {
      cgptr H = P.H();
      gptr Hs = P.Hs(), hdg = P.F();
      PQP_E(k) if (in(k))
      {
        const long o = (long)idx(k) * n + idx(k);
        const double h = H[o];
        const double v = (d.hessian == PQP_HESSIAN_DIAGONAL) ? h * Sx[k] * Sx[k] * c : h * c;
        Hs[o] = v;
        hdg[idx(k)] = v;
      }
      if (cform) {
        cgptr C = P.C();
        gptr Cs = P.Cs(), cd = P.CTs();
        PQP_E(k) if (in(k))
        {
          const long o = (long)idx(k) * n + idx(k);
          const double v = Sc[k] * C[o] * Sx[k];
          Cs[o] = v;
          cd[idx(k)] = v;
        }
      }
      bytes(((long)n * 3 + (long)d.n_in * 3) * 8);
    }
    {
      cgptr g = P.g();
      gptr gsg = P.gs();
      PQP_E(k) if (in(k)) gsg[idx(k)] = g[idx(k)] * Sx[k] * c;
    }
    if (cform) {
      cgptr u = P.u(), l = P.l();
      gptr us = P.us(), ls = P.ls();
      PQP_E(k) if (in(k))
      {
        us[idx(k)] = u[idx(k)] * Sc[k];
        ls[idx(k)] = l[idx(k)] * Sc[k];
      }
    }
    if (boxf) {
      cgptr u = P.u_box(), l = P.l_box();
      gptr us = P.ubs(), ls = P.lbs(), is = P.is();
      PQP_E(k) if (in(k))
      {
        double uu = u[idx(k)], ll = l[idx(k)];
        uu = (uu <= 1.E20) ? uu : 1.E20;
        ll = (ll >= -1.E20) ? ll : -1.E20;
        us[idx(k)] = uu * Sc[k];
        ls[idx(k)] = ll * Sc[k];
        is[idx(k)] = Sx[k] * Sc[k];
      }
    }
    // (delta is rewritten with the values it holds by the workgroup kernels: nothing to do)
    __syncthreads();
  }

  // ---- primal block: L = I, D = diag(H_s) + rho; one entry of Z and of the Gram matrix per constraint
  // (Solver::factor_primal_block + build_ZG, their dm() branches)
  __device__ __forceinline__ void factor_primal_block()
  {
    const double rho = info.rho;
    PQP_E(c) dF[c] = in(c) ? ((hess() == PQP_HESSIAN_DIAGONAL) ? lv(LV_HD, c) : 0.0) + rho : 1.0;
    vstore(P.dF(), dF);
    if (hasc) {
      gptr Zr = P.Zr(), gdg = P.G();
      PQP_E(c) if (in(c))
      {
        Zr[idx(c)] = zd[c];
        gdg[idx(c)] = zd[c] * zd[c] / dF[c];
      }
      bytes((long)d.nd * 8 * 3);
    }
  }

  // ---- active set from fl bit 2 (Solver::apply_active_set, dm() path: never incremental; the slot order is the
  // ascending constraint order, which only the persistent slot list and dS in HBM ever see)
  __device__ __forceinline__ void apply_active_set()
  {
    tic();
    // (counts on the scalar unit: a ballot and a population count per slot)
    int changed = 0;
    if (hasc) {
      PQP_E(c)
      {
        const bool want = (fl[c] & 4) != 0, had = (fl[c] & 8) != 0;
        changed += __popcll(__ballot((want != had) ? 1 : 0));
      }
    }
    if (changed == 0 && !schur_dirty) {
      toc(ST_CYC_ZG);
      return;
    }
    int tot = 0;
    PQP_E(c)
    {
      const bool want = (fl[c] & 4) != 0;
      fl[c] = (fl[c] & 7) | (want ? 8 : 0);
      tot += __popcll(__ballot(want ? 1 : 0));
    }
    n_c = uni(tot);
    schur_dirty = true;
    toc(ST_CYC_ZG);
    if (n_c > 0) {
      // diagonal Schur block: D_S = mu_in + gd over the active constraints (Solver::factor_schur)
      const double mu_in = info.mu_in;
      PQP_E(c) dS[c] = active(c) ? mu_in + zd[c] * zd[c] / dF[c] : 1.0;
      bytes((long)n_c * 8);
      count(ST_N_SCHUR_FACT);
    }
    schur_dirty = false;
    toc(ST_CYC_SCHUR);
  }

  // ---- K = [[D, Z_J^T], [Z_J, -mu I]] solved element-wise (Solver::kkt_solve_in_place, dm() branch)
  __device__ __forceinline__ void kkt_solve_in_place(double (&bx)[E], double (&bd)[E])
  {
    PQP_E(c)
    {
      double t = bx[c];
      if (active(c)) {
        const double s = zd[c] * (bx[c] / dF[c]) - bd[c];
        bd[c] = s / dS[c];
        t -= zd[c] * bd[c];
      }
      bx[c] = t / dF[c];
    }
    bytes((long)n_c * 16);
    count(ST_N_KKT_SOLVES);
  }

  // err = rhs - K sol; by-products Hdx, Cdx, CTdz (Solver::kkt_residual)
  __device__ __forceinline__ double kkt_residual(int mode)
  {
    const double rho = info.rho, mu_in = info.mu_in;
    double m = 0;
    PQP_E(c)
    {
      Hdx[c] = (hess() == PQP_HESSIAN_DIAGONAL) ? lv(LV_HD, c) * dx[c] : 0.0;
      if (hasc) {
        Cdx[c] = zd[c] * dx[c];
        CTdz[c] = zd[c] * (active(c) ? sd[c] : 0.0); // (the dual solution by constraint, zero where inactive)
      } else {
        CTdz[c] = 0.0;
      }
      const double e = rhs_x(c, mode) - rho * dx[c] - Hdx[c] - CTdz[c];
      ex[c] = e;
      m = vmax_abs(m, e);
      if (active(c)) {
        const double e2 = rhs_d(c, mode) - (Cdx[c] - sd[c] * mu_in);
        ed[c] = e2;
        m = vmax_abs(m, e2);
      }
    }
    bytes(((long)n + (long)d.n_in) * 8);
    return lane_max0(m);
  }

  // reference solver.hpp:406-541 (Solver::iterative_solve): solve + refinement on the unfactorised operator
  __device__ __forceinline__ void iterative_solve(double eps, int mode)
  {
    PQP_E(c)
    {
      dx[c] = 0.0;
      sd[c] = 0.0;
      ex[c] = rhs_x(c, mode);
      ed[c] = rhs_d(c, mode);
    }
    long it = 0, it_stability = 0;
    UD preverr = 0, cur = 0;
    while (true) {
      tic();
      kkt_solve_in_place(ex, ed);
      PQP_E(c)
      {
        dx[c] += ex[c];
        if (active(c))
          sd[c] += ed[c];
      }
      toc(ST_CYC_KKT_SOLVE);
      cur = kkt_residual(mode);
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
  }

  // mode 0: semismooth Newton step (reference solver.hpp:754-869); 1: equality-constrained initial guess
  // (helpers.hpp:199-228); 2: install the active set in fl only (solver.hpp:1231-1240)   (Solver::linear_step)
  __device__ __forceinline__ void linear_step(int mode, double eps)
  {
    if (mode == 0 && hasc) {
      PQP_E(c) if (in(c))
      {
        const int up = rup[c] >= 0 ? 1 : 0;
        const int lo = si[c] <= 0 ? 2 : 0;
        fl[c] = (fl[c] & 8) | up | lo | ((up | lo) ? 4 : 0);
      }
    }
    apply_active_set();
    if (mode == 2)
      return;
    iterative_solve(eps, mode);
    if (mode == 1) {
      vcopy(x, dx);
      return;
    }
    PQP_E(c)
    {
      dz[c] = active(c) ? sd[c] : -z[c];
      CTdz[c] -= ctz_inactive(c);
    }
  }

  // ---- exact line search (reference linesearch.hpp:320-538; Solver::primal_dual_ls / ls_bracket).
  // Inequality part of (a, b) of phi' at NP step lengths: every lane sums the terms of its own constraints, one wavefront
  // reduction per sum.
  template<int NP>
  __device__ __forceinline__ void ls_terms(const double (&al)[NP], double (&a_in)[NP], double (&b_in)[NP])
  {
    const bool gpdal = st.merit_function_type == PQP_MERIT_GPDAL;
    double sa[NP], sb[NP], sa2[NP], sb2[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p)
      sa[p] = sb[p] = sa2[p] = sb2[p] = 0.0;
    PQP_E(c)
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
        if (!gpdal) {
          const double e2 = e - dzi, apz2 = apz - zi;
          sa2[p] = fma(e2, e2, sa2[p]);
          sb2[p] = fma(e2, apz2, sb2[p]);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      sa[p] = lane_sum(sa[p]);
      sb[p] = lane_sum(sb[p]);
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

  static constexpr int NBP = 2 * E; // breakpoints a lane owns: two per constraint

  // The breakpoints flagged in `take` evaluated exactly, then the selections of linesearch.hpp:427-536.  `pred`: the
  // largest breakpoint at or below the bracket (0: none).  false: the caller must evaluate every breakpoint.
  // `succ`: the first breakpoint above the bracket (INF: none) -- evaluated only when no breakpoint of `take` has phi' >= 0.
  __device__ __forceinline__ bool ls_select(const double (&mine)[NBP], const bool (&take)[NBP], double a0, double b0, double pred,
                                            bool all_negative, double amax, bool everything, double& result,
                                            double succ = __builtin_inf())
  {
    const double INF = __builtin_inf();
    double gr[NBP];
#pragma unroll
    for (int r = 0; r < NBP; ++r)
      gr[r] = 0.0;
    int nev = 0;
#pragma unroll
    for (int r = 0; r < NBP; ++r) {
      unsigned long long m = __ballot(take[r] ? 1 : 0);
      while (m != 0ull) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1ull;
        const double al = wave_bcast(mine[r], src);
        const double g = ls_grad(al, a0, b0);
        if (lane == src)
          gr[r] = g;
        ++nev;
      }
    }
    count(ST_N_LS_BREAKPOINTS, nev);
    double afp = INF;
#pragma unroll
    for (int r = 0; r < NBP; ++r)
      if (take[r] && !(gr[r] < 0) && mine[r] < afp)
        afp = mine[r];
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
        return false; // the exact value at the largest breakpoint is not negative after all
      double a1[1] = { 2 * amax + 1 }, ai[1], bi[1];
      ls_terms<1>(a1, ai, bi);
      result = -(b0 + bi[0]) / (a0 + ai[0]);
      return true;
    }
    double aln = 0.0;
    if (!(afp < INF)) {
      if (!everything)
        return false;
      // no breakpoint with phi' >= 0 (linesearch.hpp:496-526): aln = the largest one
#pragma unroll
      for (int r = 0; r < NBP; ++r)
        if (take[r])
          aln = vmax(aln, mine[r]);
      aln = lane_max0(aln);
      double a1[1] = { 2 * aln + 1 }, ai[1], bi[1];
      ls_terms<1>(a1, ai, bi);
      result = -(b0 + bi[0]) / (a0 + ai[0]);
      return true;
    }
    if (pred > 0.0 && !(afp > pred))
      return false; // phi'(pred) >= 0 exactly: the zero lies further left than the bracket said
    double gfp = -INF;
#pragma unroll
    for (int r = 0; r < NBP; ++r)
      if (take[r]) {
        if (mine[r] == afp && !(gr[r] < 0))
          gfp = vmax(gfp, gr[r]);
        if (mine[r] < afp)
          aln = vmax(aln, mine[r]);
      }
    gfp = wave_max(gfp);
    if (afp == succ && !(g_succ < 0))
      gfp = vmax(gfp, g_succ);
    aln = lane_max0(aln);
    double gln = -INF;
#pragma unroll
    for (int r = 0; r < NBP; ++r)
      if (take[r] && mine[r] == aln)
        gln = vmax(gln, gr[r]);
    gln = wave_max(gln);
    if (aln == 0.0) { // no breakpoint before afp: linesearch.hpp:477-495
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
    double s_dxHdx = 0, s_dx2 = 0, s_xHdx = 0, s_errdx = 0, s_dz2 = 0, s_dzz = 0, s_bmag = 0, dwm = 0;
    PQP_E(c)
    {
      const double dxk = dx[c];
      dwm = vmax_abs(dwm, dxk);
      s_dxHdx += dxk * Hdx[c];
      s_dx2 += dxk * dxk;
      s_xHdx += x[c] * Hdx[c];
      s_errdx += (info.rho * (x[c] - lv(LV_XP, c)) + lv(LV_GS, c)) * dxk;
    }
    PQP_E(c)
    {
      dwm = vmax_abs(dwm, dz[c]);
      s_dz2 += dz[c] * dz[c];
      s_dzz += dz[c] * z[c];
      const double ac = fabs(Cdx[c]), ar = fabs(rup[c]) + fabs(si[c]);
      s_bmag = fma(ar, ac, s_bmag);
      if (!gpdal)
        s_bmag = fma(double(info.nu) * (ar + fabs(z[c]) * info.mu_in), ac + fabs(dz[c]) * info.mu_in, s_bmag);
    }
    dw_max = lane_max0(dwm);
    s_dxHdx = lane_sum(s_dxHdx);
    s_dx2 = lane_sum(s_dx2);
    s_xHdx = lane_sum(s_xHdx);
    s_errdx = lane_sum(s_errdx);
    s_dz2 = lane_sum(s_dz2);
    s_dzz = lane_sum(s_dzz);
    s_bmag = lane_sum(s_bmag);
    const double nu = gpdal ? 1.0 : double(info.nu);
    double a0 = s_dxHdx + info.mu_eq_inv * 0.0 + info.rho * s_dx2 + 0.0 * info.mu_eq_inv * nu;
    double b0 = s_xHdx + s_errdx + info.mu_eq_inv * 0.0 + nu * info.mu_eq_inv * 0.0;
    if (gpdal) {
      a0 += info.mu_in * (1. - st.alpha_gpdal) * s_dz2;
      b0 += info.mu_in * (1. - st.alpha_gpdal) * s_dzz;
    }
    const double bmag = gpdal ? info.mu_in_inv * s_bmag / st.alpha_gpdal : info.mu_in_inv * s_bmag;
    sub_tic(ST_CYC_LS_EVAL);
    // this lane's breakpoints (linesearch.hpp:378-391): two per constraint
    double mine[NBP];
    double amax = 0;
    int cnti = 0;
    PQP_E(c)
    {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        double al = -1.0;
        if (in(c) && Cdx[c] != 0.) {
          const double num = h ? si[c] : rup[c];
          al = -num / (Cdx[c] + MACHINE_EPS);
        }
        const bool ok = al > MACHINE_EPS;
        mine[2 * c + h] = ok ? al : -1.0;
        cnti += __popcll(__ballot(ok ? 1 : 0));
        if (ok) {
          amax = vmax(amax, al);
        }
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
    constexpr int WCAP = 32; // breakpoints evaluated exactly at most before the bracket is given up
    bool take[NBP];
    bool done = false;
    if (cnt > 8.0 && amax < INF) {
      // Bracket (lo, hi] of the zero of the monotone phi' (the job of Solver::ls_bracket).  The probes are BREAKPOINTS: a
      // round takes one that lies strictly inside the bracket (every lane offers the first of its own, the lane in the
      // middle of those that offer one is taken: a pivot of arbitrary rank, as in quickselect), evaluates phi' there --
      // each lane its own constraints' terms, two wavefront reductions -- and keeps the side the zero is on: the number of
      // breakpoints left shrinks by a third per round on average whatever their distribution over the decades (the
      // geometric quarter-steps of the workgroup kernel cost three evaluations, a division and two square roots a round).
      // Nothing here is trusted: the exact evaluation of the few breakpoints that remain validates the bracket (ls_select).
      const double SURE = 3.6e-15 * (double)(d.nc + d.n + d.n_eq);
      double lo = 0.0, hi = INF, inside = cnt; // (hi = INF: no point with phi' > 0 seen yet)
      bool all_negative = false, give_up = false;
      double newt = -1.0; // zero of the linear piece of phi' at the last probe: the semismooth Newton step on phi'
      {
        const double al[1] = { 0.0 };
        double ai[1], bi[1];
        ls_terms<1>(al, ai, bi);
        const double g0 = b0 + bi[0], mag0 = fabs(b0) + bmag;
        if (a0 + ai[0] > 0.)
          newt = -g0 / (a0 + ai[0]);
        if (!(g0 < -SURE * mag0))
          give_up = true;
      }
      if (!give_up && !all_negative) {
        bool newton_ok = true;
        for (int round = 0; round < 40 && (inside > 6.0 || !(hi < INF)); ++round) {
          // the probe: the Newton point when it lies strictly inside the bracket (phi' is piecewise linear: from a probe on
          // the zero's own piece the step lands on the zero, and a handful of steps get there from anywhere) -- else, or
          // when the last Newton probe kept more than half of the breakpoints, a breakpoint of middle rank
          // While no point with phi' > 0 is known, a Newton point beyond the last breakpoint -- and the last probe of a
          // bracket that has come down to a few breakpoints -- is the last breakpoint itself: phi' < 0 there is the case
          // of linesearch.hpp:496-526 (the zero lies beyond every breakpoint).
          double pv;
          const bool by_newton = newton_ok && newt > lo && newt < hi;
          if (!(hi < INF) && (!(inside > 6.0) || (by_newton && !(newt < amax)))) {
            pv = amax;
          } else if (by_newton) {
            pv = newt;
          } else {
            double cand = -1.0;
#pragma unroll
            for (int r = NBP - 1; r >= 0; --r)
              if (mine[r] > lo && mine[r] < hi)
                cand = mine[r];
            const unsigned long long m = __ballot(cand > 0 ? 1 : 0);
            if (m == 0ull)
              break; // (what is left are ties with hi)
            const int rank = __popcll(m & ((lane == 0) ? 0ull : (~0ull >> (WAVE - lane))));
            const unsigned long long pick = __ballot((cand > 0 && rank == (__popcll(m) >> 1)) ? 1 : 0);
            pv = wave_bcast(cand, __ffsll((long long)pick) - 1);
          }
          const double a1[1] = { pv };
          double ai[1], bi[1];
          ls_terms<1>(a1, ai, bi);
          const double slope = a0 + ai[0];
          const double g = slope * pv + (b0 + bi[0]), mag = fabs(slope * pv) + fabs(b0) + bmag;
          // breakpoints in (lo, pv]: counted on the scalar unit (a ballot and a population count per slot) -- no vector
          // accumulation, no wavefront reduction
          int cbi = 0;
#pragma unroll
          for (int r = 0; r < NBP; ++r)
            cbi += __popcll(__ballot((mine[r] > lo && mine[r] <= pv) ? 1 : 0));
          const double cb = (double)cbi;
          const double before = inside;
          if (g > SURE * mag) {
            hi = pv;
            inside = cb;
          } else if (g < -SURE * mag) {
            lo = pv;
            inside -= cb;
          } else {
            // too close to the zero to trust the sign: the zero is at this point or right beside it -- between the
            // breakpoint below it and the first one at or above it
            double below = 0, above = -INF;
#pragma unroll
            for (int r = 0; r < NBP; ++r) {
              if (mine[r] < pv)
                below = vmax(below, mine[r]);
              if (mine[r] >= pv)
                above = vmax(above, -mine[r]);
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
          for (int r = 0; r < NBP; ++r)
            if (mine[r] > 0) {
              if (mine[r] <= lo)
                below = vmax(below, mine[r]);
              if (mine[r] > hi)
                above = vmax(above, -mine[r]);
            }
          pred = lane_max0(below);
          succ = -wave_max(above);
        }
#pragma unroll
        for (int r = 0; r < NBP; ++r) {
          const double a = mine[r];
          take[r] = all_negative ? (a > 0 && a == amax) : (a > 0 && ((a > lo && a <= hi) || a == pred));
        }
        done = ls_select(mine, take, a0, b0, pred, all_negative, amax, false, result, succ);
      }
    }
    if (!done) {
      // every breakpoint (few of them, or a bracket that could not be trusted): the reference's own evaluation
#pragma unroll
      for (int r = 0; r < NBP; ++r)
        take[r] = mine[r] > 0;
      ls_select(mine, take, a0, b0, 0.0, false, amax, true, result);
    }
    sub_toc(ST_CYC_LS_EVAL);
    return result;
  }

  // both infeasibility certificates + the inner stopping criterion (Solver::saddle_point_and_certificates;
  // reference utils.hpp:269-324, :343-419, solver.hpp:687-743)
  __device__ __forceinline__ void saddle_point_and_certificates(bool do_cert, double& err_in, bool& primal_infeasible,
                                                                bool& dual_infeasible)
  {
    const double c = ruiz_c;
    const double NEG = -__builtin_inf();
    double lb1 = 0, gdx = 0, nrm_dz = 0, lb2 = 0, ndx = 0, nhdx = 0, mviol = NEG, e1 = 0, e3 = 0;
    {
      const double zf = (st.merit_function_type == PQP_MERIT_GPDAL) ? st.alpha_gpdal : 1.0;
      PQP_E(k)
      {
        if (hasc && in(k)) {
          const double up = rup[k], lo = si[k];
          const double v = (up > 0 ? up : 0.0) + (lo < 0 ? lo : 0.0) - zf * z[k] * info.mu_in;
          e1 = vmax_abs(e1, v);
        }
        e3 = vmax_abs(e3, dres[k]);
      }
    }
    if (do_cert) {
      double sx[E], sc_[E];
      PQP_E(k)
      {
        sx[k] = lv(LV_SX, k);
        sc_[k] = lv(LV_SC, k);
      }
      PQP_E(k) if (in(k))
      {
        const double sc = sx[k] * c;
        CTdz[k] /= sc;
        lb2 = vmax_abs(lb2, 0.0 + CTdz[k]);
        Hdx[k] /= sc;
        nhdx = vmax_abs(nhdx, Hdx[k]);
        gdx += dx[k] * lv(LV_GS, k);
        dx[k] *= sx[k];
        ndx = vmax_abs(ndx, dx[k]);
      }
      if (hasc) {
        PQP_E(k) if (in(k))
        {
          const double v = dz[k];
          const double ubk = lv(LV_UB, k), lbk = lv(LV_LB, k);
          lb1 += (v > 0 ? v : 0.0) * ubk;
          lb1 -= (v < 0 ? v : 0.0) * lbk;
          dz[k] = cform ? v * sc_[k] / c : sc_[k] * v / c;
          nrm_dz = vmax_abs(nrm_dz, dz[k]);
          Cdx[k] /= sc_[k];
          // utils.hpp:381-398: two-sided bound -> |w| <= bound; no upper bound -> -w <= bound; no lower -> w <= bound
          const double w = cform ? Cdx[k] : dx[k]; // (box form: the unscaled dx itself)
          const double val = (ubk <= 1.E20 && lbk >= -1.E20) ? fabs(w) : ((ubk > 1.E20) ? -w : w);
          mviol = vmax(mviol, val);
        }
      }
    }
    lb1 = lane_sum(lb1);
    gdx = lane_sum(gdx);
    nrm_dz = lane_max0(nrm_dz);
    lb2 = lane_max0(lb2);
    ndx = lane_max0(ndx);
    nhdx = lane_max0(nhdx);
    mviol = wave_max(mviol);
    e1 = lane_max0(e1);
    e3 = lane_max0(e3);
    err_in = fmax(e1, fmax(0.0, e3));
    primal_infeasible = false;
    dual_infeasible = false;
    if (!do_cert)
      return;
    {
      const double upper_bound = st.eps_primal_inf * fmax(0.0, nrm_dz);
      primal_infeasible = (nrm_dz != 0) && lb2 <= upper_bound && lb1 <= -upper_bound;
    }
    {
      double bound = ndx * st.eps_dual_inf;
      const bool first_cond = (0.0 <= bound) && !(mviol > bound);
      bound *= c;
      const bool second_cond_alt1 = nhdx <= bound && gdx <= -bound;
      dual_infeasible = first_cond && second_cond_alt1 && ndx != 0;
    }
  }

  // reference solver.hpp:882-1077 (Solver::newton_semi_smooth)
  __device__ __forceinline__ void newton_semi_smooth(double eps_int)
  {
    for (long iter = 0; iter <= st.max_iter_in; ++iter) {
      if (iter == st.max_iter_in) {
        info.iter += st.max_iter_in + 1;
        break;
      }
      count(ST_N_NEWTON);
      linear_step(0, eps_int);
      tic();
      if (st.merit_function_type == PQP_MERIT_GPDAL && hasc) {
        PQP_E(c) Cdx[c] += (st.alpha_gpdal - 1.) * info.mu_in * dz[c];
      }
      UD alpha = 1.0;
      double dw_max = 0;
      if (hasc) {
        alpha = primal_dual_ls(dw_max);
      } else {
        PQP_E(c) dw_max = vmax_abs(dw_max, dx[c]);
        dw_max = lane_max0(dw_max);
      }
      toc(ST_CYC_LINESEARCH);
      sub_tic(ST_CYC_UPDATE);
      if (fabs(alpha) * dw_max < 1.E-11 && iter > 0) {
        info.iter += iter + 1;
        sub_toc(ST_CYC_UPDATE);
        break;
      }
      PQP_E(c)
      {
        x[c] += alpha * dx[c];
        dres[c] += alpha * (info.rho * dx[c] + Hdx[c] + 0.0 + CTdz[c]);
        if (hasc) {
          rup[c] += alpha * Cdx[c];
          si[c] += alpha * Cdx[c];
          z[c] += alpha * dz[c];
        }
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

  // reference utils.hpp:164-252 (Solver::global_primal_residual, dm() branch)
  __device__ __forceinline__ void global_primal_residual(UD& lhs, UD& eq_rhs_0, UD& in_rhs_0, UD& eq_lhs, UD& in_lhs)
  {
    double m_in0 = 0, m_inl = 0;
    if (cform) {
      cgptr uu = P.u(), ll = P.l();
      double dv[E], uv[E], lwv[E];
      PQP_E(k) dv[k] = lv(LV_SC, k);
      vload(uv, uu);
      vload(lwv, ll);
      PQP_E(k) if (in(k))
      {
        CTdz[k] = zd[k] * z[k];
        const double v = (zd[k] * x[k]) / dv[k]; // unscaled C x
        rup[k] = v;
        m_in0 = vmax_abs(m_in0, v);
        const double pu = v - uv[k], pl = v - lwv[k];
        const double sv = (pu > 0 ? pu : 0.0) + (pl < 0 ? pl : 0.0);
        si[k] = sv;
        m_inl = vmax_abs(m_inl, sv);
      }
    } else {
      vzero(CTdz);
    }
    if (boxf) {
      cgptr ubx = P.u_box(), lbx = P.l_box();
      double dv[E], uv[E], lwv[E];
      PQP_E(k) dv[k] = lv(LV_SX, k);
      vload(uv, ubx);
      vload(lwv, lbx);
      PQP_E(k) if (in(k))
      {
        const double v = x[k] * dv[k]; // unscaled x
        rup[k] = v;
        const double pu = v - uv[k], pl = v - lwv[k];
        const double sv = (pu > 0 ? pu : 0.0) + (pl < 0 ? pl : 0.0);
        si[k] = sv;
        m_inl = vmax_abs(m_inl, sv);
        m_in0 = vmax_abs(m_in0, x[k] - sv); // utils.hpp:225-229 (as written)
        m_in0 = vmax_abs(m_in0, x[k]);      // utils.hpp:230-231
      }
    }
    aty_fresh = true;
    bytes((long)d.n_in * 8);
    eq_rhs_0 = 0.0;
    in_rhs_0 = lane_max0(m_in0);
    eq_lhs = 0.0;
    in_lhs = lane_max0(m_inl);
    lhs = fmax(eq_lhs, in_lhs);
    if (PQP_UNLIKELY(st.primal_infeasibility_solving && info.status == PQP_PRIMAL_INFEASIBLE)) {
      // utils.hpp:241-248 : || A^T se + C^T si ||_inf on the unscaled model (C is diagonal here)
      double m = 0;
      if (cform) {
        cgptr C = P.C();
        PQP_E(k) if (in(k)) m = vmax_abs(m, 0.0 + C[(long)idx(k) * n + idx(k)] * si[k]);
      }
      lhs = lane_max0(m);
    }
  }

  // reference utils.hpp:437-587 (Solver::global_dual_residual)
  __device__ __forceinline__ void global_dual_residual(UD& lhs, UD& rhs_0, UD& rhs_1, UD& rhs_3, UD& rhs_duality_gap,
                                                       UD& duality_gap)
  {
    const double c = ruiz_c;
    double m0 = 0, m3 = 0, ml = 0, xHx = 0, gx = 0, zu = 0, zl = 0;
    const double ib = 1.3407807929942596e+154; // sqrt(DBL_MAX), helpers/common.hpp:17-25
    double sx[E], gv[E], sc_[E], uv[E], lwv[E];
    if (hasc) {
      vload(uv, cform ? P.u() : P.u_box());
      vload(lwv, cform ? P.l() : P.l_box());
    }
    const bool have_products = aty_fresh;
    bytes(((hess() == PQP_HESSIAN_ZERO ? 0L : (long)n) + (have_products ? 0L : (long)d.n_in)) * 8);
    PQP_E(k)
    {
      sx[k] = lv(LV_SX, k);
      gv[k] = lv(LV_GU, k);
      sc_[k] = lv(LV_SC, k);
    }
    PQP_E(k) if (in(k))
    {
      const double sc = sx[k] * c;
      const double hx = (hess() == PQP_HESSIAN_DIAGONAL) ? lv(LV_HD, k) * x[k] : 0.0;
      double ctz = cform ? (have_products ? CTdz[k] : zd[k] * z[k]) : 0.0;
      const double v = hx / sc; // unscaled H x (utils.hpp:469-471)
      m0 = vmax_abs(m0, v);
      const double xu = x[k] * sx[k];
      xHx += v * xu;
      gx += gv[k] * xu;
      double m3k = fabs(ctz / sc);
      if (boxf) {
        const double zb = z[k] * zd[k];
        ctz += zb;
        m3k = vmax_abs(m3k, zb / sc);
      }
      m3 = vmax(m3, m3k);
      const double dr = lv(LV_GS, k) + hx + 0.0 + ctz;
      dres[k] = dr;
      ml = vmax_abs(ml, dr / sc);
      if (hasc) {
        // duality gap terms (utils.hpp:482-586)
        const double zi = cform ? z[k] * sc_[k] / c : sc_[k] * z[k] / c;
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
    zu = lane_sum(zu);
    zl = lane_sum(zl);
    m0 = lane_max0(m0);
    m3 = lane_max0(m3);
    ml = lane_max0(ml);
    rhs_0 = (hess() == PQP_HESSIAN_ZERO) ? 0.0 : m0;
    rhs_1 = 0.0;
    rhs_3 = m3;
    lhs = ml;
    duality_gap = gx;
    rhs_duality_gap = fabs(gx);
    if (hess() != PQP_HESSIAN_ZERO) {
      duality_gap += xHx;
      rhs_duality_gap = fmax(rhs_duality_gap, fabs(xHx));
    }
    // (no equality: by = 0)
    rhs_duality_gap = fmax(rhs_duality_gap, 0.0);
    duality_gap += 0.0;
    if (cform) {
      rhs_duality_gap = fmax(rhs_duality_gap, fabs(zu));
      duality_gap += zu;
      rhs_duality_gap = fmax(rhs_duality_gap, fabs(zl));
      duality_gap += zl;
    } else {
      // (the general-inequality sums are empty; the box sums follow them in the reference's order)
      rhs_duality_gap = fmax(rhs_duality_gap, 0.0);
      duality_gap += 0.0;
      rhs_duality_gap = fmax(rhs_duality_gap, 0.0);
      duality_gap += 0.0;
      if (boxf) {
        rhs_duality_gap = fmax(rhs_duality_gap, fabs(zu));
        duality_gap += zu;
        rhs_duality_gap = fmax(rhs_duality_gap, fabs(zl));
        duality_gap += zl;
      }
    }
  }

  // ---- reference solver.hpp:1088-1843 (Solver::solve)
  __device__ __forceinline__ void solve()
  {
    State W = *P.state();
    info.load(*P.info());
    ruiz_c = W.ruiz_c;
    dual_feasibility_rhs_2 = W.dual_feasibility_rhs_2;
#ifdef PQP_STATS
    for (int k = lane; k < ST_COUNT + 2; k += WAVE)
      lds_stat[k] = 0;
    __syncthreads();
#endif
    const long long cyc0 = clock64();
    const long long wall0 = wall_clock64();
    const int nc = d.nc;
    // every register slot beyond dim holds a benign value from here on (0; 1 for the divisors dF, dS): the element-wise
    // code runs over all E slots of all lanes, and the reductions take whatever those slots hold
    PQP_E(c)
    {
      dres[c] = rup[c] = si[c] = 0.0;
      dx[c] = dz[c] = Hdx[c] = Cdx[c] = CTdz[c] = ex[c] = ed[c] = sd[c] = 0.0;
      lv_set(LV_XP, c, 0.0);
      lv_set(LV_ZP, c, 0.0);
    }
    // results -> registers (the warm-start modes read them)
    vload(x, P.x());
    if (hasc)
      vload(z, P.z());
    else
      vzero(z);
    {
      // the persistent active_set_up / active_set_low flags (bits 16-17 of act[i], see Solver::solve)
      const PQP_GLOBAL int* ga = P.act();
      PQP_E(c) fl[c] = (hasc && in(c)) ? act_flags(ga[idx(c)]) : 0;
    }
    tic();
    const int ig = st.initial_guess;
    const bool wswpr = (ig == PQP_WARM_START_WITH_PREVIOUS_RESULT);
    const bool dirty = W.dirty != 0;
    const bool do_rescale = dirty && !wswpr;
    bool do_factor, do_scale_ws, do_aset_from_z, do_eq_guess = false, do_restore = false;
    if (dirty) {
      if (ig == PQP_EQUALITY_CONSTRAINED_INITIAL_GUESS || ig == PQP_NO_INITIAL_GUESS) {
        vzero(x); // results.cleanup
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
    if (do_rescale) {
      toc(ST_CYC_F_LOAD); // (instrumented build: prologue, outer-loop logic and epilogue of this kernel are billed to the
                          // three counters the dense path uses for its blocked factorisation: f_load, f_update, f_writeback)
      rescale(W.scaled_valid == 0);
      toc(ST_CYC_SCALE);
    }
    // the equilibrated model: g_s, the bounds, the diagonals of H_s and of the constraint rows
    lv_load(LV_GS, P.gs());
    lv_load(LV_HD, P.F());
    lv_load(LV_SX, P.dlt_x(), 1.0);
    lv_load(LV_GU, P.g());
    if (hasc) {
      lv_load(LV_SC, dlt_c(), 1.0);
    } else {
      PQP_E(c) lv_set(LV_SC, c, 1.0);
    }
    if (cform) {
      lv_load(LV_UB, P.us());
      lv_load(LV_LB, P.ls());
      vload(zd, P.CTs());
    } else if (boxf) {
      lv_load(LV_UB, P.ubs());
      lv_load(LV_LB, P.lbs());
      vload(zd, P.is());
    } else {
      PQP_E(c)
      {
        lv_set(LV_UB, c, 0.0);
        lv_set(LV_LB, c, 0.0);
      }
      vzero(zd);
    }
    if (do_scale_ws) {
      // solver.hpp:1137-1146: the warm start into the equilibrated space
      PQP_E(k) x[k] /= lv(LV_SX, k);
      if (hasc) {
        PQP_E(k) z[k] = z[k] / lv(LV_SC, k) * ruiz_c;
      }
    }
    PQP_E(c)
    {
      dF[c] = 1.0;
      dS[c] = 1.0;
    }
    if (do_factor) {
      toc(ST_CYC_F_LOAD);
      factor_primal_block();
      toc(ST_CYC_FACTOR_H);
      n_c = 0;
      schur_dirty = true;
    }
    if (do_restore) {
      // WARM_START_WITH_PREVIOUS_RESULT on an unchanged model: the state the previous solve left in HBM
      // (solver.hpp:1173-1187, 1343-1375): D, the Gram entries, the slot list and D_S by slot
      vload(dF, P.dF(), 1.0);
      n_c = W.n_c;
      const int n_slots = W.n_slots;
      {
        const PQP_GLOBAL int* ga = P.act();
        cgptr dSg = P.dS();
        for (int j = lane; j < 64 * E; j += WAVE)
          lds_i[j] = -1;
        __syncthreads();
        for (int j = lane; j < n_slots; j += WAVE) {
          const int i = act_cid(ga[j]);
          if (i >= 0) {
            lds_i[i] = j;
            lds_d[i] = dSg[j];
          }
        }
        __syncthreads();
        PQP_E(c)
        {
          const int s = lds_i[idx(c)];
          if (in(c) && s >= 0) {
            fl[c] |= 8;
            dS[c] = lds_d[idx(c)];
          }
        }
        __syncthreads();
      }
      schur_dirty = !(W.ls_valid && W.mu_eq_fact == info.mu_eq && W.mu_in_fact == info.mu_in);
    }
    if (do_aset_from_z || do_eq_guess) {
      if (do_aset_from_z) {
        PQP_E(c) fl[c] = (fl[c] & 11) | ((z[c] != 0) ? 4 : 0); // only active_inequalities is rewritten (solver.hpp:1231-1238)
      } else {
        PQP_E(c) fl[c] = (fl[c] & 15); // (mode 1 keeps the wanted bits as they are: none set on this path)
      }
      linear_step(do_eq_guess ? 1 : 2, 1.0);
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
    int stage = 0; // 0: top of loop, 1: after the Newton loop, 2: before the mu update (see Solver::solve)
    bool done = (st.max_iter <= 0);
    bool gpr_fresh = false, gdr_fresh = false;
    aty_fresh = false;
    UD pl_cache = 0, dl_cache = 0;
    toc(ST_CYC_F_LOAD);
    while (!done) {
      toc(ST_CYC_F_UPDATE);
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
          PQP_E(k) x[k] = (x[k] * lv(LV_SX, k)) / lv(LV_SX, k);
          if (hasc) {
            if (cform) {
              PQP_E(k) z[k] = (z[k] * lv(LV_SC, k) / ruiz_c) / lv(LV_SC, k) * ruiz_c;
            } else {
              PQP_E(k) z[k] = (lv(LV_SC, k) * z[k] / ruiz_c) / lv(LV_SC, k) * ruiz_c;
            }
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
        PQP_E(c)
        {
          lv_set(LV_XP, c, x[c]);
          lv_set(LV_ZP, c, z[c]);
        }
        // shifted inequality residuals (solver.hpp:1523-1559)
        if (hasc) {
          PQP_E(i) if (in(i))
          {
            double v = rup[i] * lv(LV_SC, i) + z[i] * info.mu_in;
            if (st.merit_function_type == PQP_MERIT_GPDAL)
              v += (st.alpha_gpdal - 1.) * info.mu_in * z[i];
            rup[i] = v - lv(LV_UB, i);
            si[i] = v - lv(LV_LB, i);
          }
        }
        toc(ST_CYC_F_UPDATE);
        newton_semi_smooth(bcl_eta_in);
        tic();
        gpr_fresh = false;
        gdr_fresh = false;
        aty_fresh = false;
        if (PQP_UNLIKELY(nonfinite)) {
          info.status = PQP_MAX_ITER_REACHED;
          break;
        }
        if ((info.status == PQP_PRIMAL_INFEASIBLE && !st.primal_infeasibility_solving) || info.status == PQP_DUAL_INFEASIBLE) {
          vcopy(x, dx); // certificates (solver.hpp:1572-1580)
          vcopy(z, dz);
          break;
        }
        if (PQP_UNLIKELY(scaled_eps == st.eps_abs && st.primal_infeasibility_solving && info.status == PQP_PRIMAL_INFEASIBLE)) {
          // solver.hpp:1581-1595 : || A^T 1 + C^T 1 (+ i_scaled) ||_inf * eps_abs   (C diagonal)
          double m = 0;
          if (cform) {
            cgptr C = P.C();
            PQP_E(k) if (in(k)) m = vmax_abs(m, 0.0 + C[(long)idx(k) * n + idx(k)] * 1.0 + 0.0);
          } else if (boxf) {
            PQP_E(k) if (in(k)) m = vmax_abs(m, 0.0 + 0.0 + zd[k]);
          }
          scaled_eps = lane_max0(m) * st.eps_abs;
        }
        stage = 1;
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
            PQP_E(c) z[c] = lv(LV_ZP, c);
            gdr_fresh = false; // z was reset
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
        if (n_c > 0)
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
    {
      tic();
      PQP_E(k) x[k] *= lv(LV_SX, k);
      if (hasc) {
        if (cform) {
          PQP_E(k) z[k] = z[k] * lv(LV_SC, k) / ruiz_c;
        } else {
          PQP_E(k) z[k] = lv(LV_SC, k) * z[k] / ruiz_c;
        }
        if (st.primal_infeasibility_solving && info.status == PQP_PRIMAL_INFEASIBLE) {
          PQP_E(k) si[k] /= lv(LV_SC, k);
        }
      }
    }
    // objective on the unscaled model (solver.hpp:1771-1780)
    {
      double obj = 0;
      cgptr H = P.H();
      PQP_E(k) if (in(k)) obj += 0.5 * x[k] * x[k] * H[(long)idx(k) * n + idx(k)] + lv(LV_GU, k) * x[k];
      bytes((long)n * 8);
      info.objValue = lane_sum(obj);
    }
    // write back
    vstore(P.x(), x);
    if (hasc) {
      vstore(P.z(), z);
      vstore(P.si(), si);
    }
    if (batch.hx) { // host-mapped mirrors (see Batch)
      vstore((gptr)(batch.hx + P.lq() * n), x);
      if (hasc) {
        vstore((gptr)(batch.hz + P.lq() * nc), z);
        vstore((gptr)(batch.hsi + P.lq() * nc), si);
      }
    }
    if (hasc) {
      // the slot list and D_S by slot (ascending constraint order), the persistent up / low flags in bits 16-17
      bool act_[E];
      int rk[E], tot;
      PQP_E(c) act_[c] = active(c);
      ranks(act_, rk, tot);
      for (int j = lane; j < 64 * E; j += WAVE)
        lds_i[j] = -1;
      __syncthreads();
      gptr dSg = P.dS();
      PQP_E(c) if (active(c))
      {
        lds_i[rk[c]] = idx(c);
        dSg[rk[c]] = dS[c];
      }
      __syncthreads();
      PQP_GLOBAL int* ga = P.act();
      PQP_E(c) if (in(c)) ga[idx(c)] = act_pack(lds_i[idx(c)], fl[c] & 3);
    }
    toc(ST_CYC_F_WRITEBACK);
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
      W.n_slots = n_c;
      W.factor_valid = 1;
      W.ls_valid = schur_dirty ? 0 : 1;
      W.ls_edited = 0;
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

#undef PQP_E
#undef PQP_KEEP

template<int E>
__device__ __forceinline__ void
diag_solve_body(const Batch& batch, long q, lptr lds)
{
  DiagSolver<E> S(batch, q, lds);
  S.solve();
}

} // namespace pqp

#endif
