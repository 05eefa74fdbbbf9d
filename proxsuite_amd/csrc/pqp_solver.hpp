// Batched dense ProxQP on MI355X: device-side state, the setup (init/update +
// Ruiz equilibration) kernel body and the solve kernel body.
//
// What is computed follows the reference function for function (cited inline,
// paths relative to /root/reference/include/proxsuite/proxqp/); HOW it is
// computed is re-designed for a workgroup-per-QP SIMT machine:
//
//  * KKT factorisation.  The reference keeps one permuted LDL^T of the whole
//    KKT matrix and edits it with rank-1 recurrences (insert / delete /
//    diagonal update: m sequential column steps each).  Here the same LDL^T is
//    held in *block* form with a static primal-first pivot order,
//        K = [ H+rho I   B_J^T ]   H+rho I = L D L^T          (once per solve)
//            [ B_J      -M_J   ]   Z = L^{-1} B^T, G = Z^T D^{-1} Z (cached rows)
//                                   M_J + G_JJ = L_S D_S L_S^T  (dual Schur block)
//    so an active-set change or a mu update is a gather of G_JJ plus a small
//    dense LDL^T (GEMM-shaped, no sequential rank-1 chains), new constraints
//    cost one triangular mat-vec each, and L^{-1} is kept explicitly so the
//    primal-block solves are chain-free mat-vecs.
//  * Line search.  phi'(alpha) is evaluated at ALL breakpoints concurrently
//    (one thread per breakpoint) instead of sort + sequential walk.
//  * Everything else (Ruiz, residuals, BCL, Newton bookkeeping, infeasibility
//    certificates, counters) restates the reference line by line on LDS vectors.
#ifndef PQP_SOLVER_HPP
#define PQP_SOLVER_HPP

#include "pqp_block.hpp"
#include "pqp_types.h"

namespace pqp {

// branch-weight hints: the register allocator places spill code and splits live ranges by block
// frequency, so the rare, register-hungry paths are marked as such
#define PQP_UNLIKELY(x) __builtin_expect(!!(x), 0)
#define PQP_LIKELY(x) __builtin_expect(!!(x), 1)

constexpr double MACHINE_EPS = 2.220446049250313e-16;
#ifndef PQP_ZG_DEPTH
#define PQP_ZG_DEPTH 8
#endif
constexpr int ZG_DEPTH = PQP_ZG_DEPTH; // MFMA k-steps (of 4) whose operand loads are in flight together in build_ZG
// 512- / 1024-thread kernels: build_ZG works on 2 x 2 tiles per wavefront (four operand tiles feed four products: a
// third less operand traffic than the 1 x 2 units of the 256-thread kernels, whose matrices mostly sit in L2)
// Line search of shapes with more breakpoints than threads: bracket the zero of phi' first, evaluate exactly only around it
#ifndef PQP_LS_UNROLL
#define PQP_LS_UNROLL 4
#endif
#ifndef PQP_LS_BRACKET
#define PQP_LS_BRACKET 1
#endif
#ifndef PQP_ZG_BLOCK2
#define PQP_ZG_BLOCK2 1
#endif
constexpr int ZG2_DEPTH = 4; // the same for the 2 x 2 units
constexpr int SCHUR_MB = 7; // register-resident Schur factorisation up to 16 * SCHUR_MB rows
// `top`: LDS scratch of the factorisation routines (ldlt_factor_mfma: 2 * 256 + 32 doubles;
// ldlt_factor_reg: 4 * 16 * MB; ldlt_inverse_reg: 6 * 16 * MB)
constexpr int TOP_DOUBLES = (6 * 16 * SCHUR_MB > 2 * PQP_NB * PQP_NB + 2 * PQP_NB) ? 6 * 16 * SCHUR_MB
                                                                                 : 2 * PQP_NB * PQP_NB + 2 * PQP_NB;
// An active-set change of at most incr_max(r) constraints (insertions + deletions) edits the inverse
// factor of the dual Schur block in place (one rank-1 sweep each); larger ones, and every mu
// update, re-factorise it.  A sweep costs O(r^2) and a few barrier intervals, the re-factorisation
// O(r^3) and r of them: the break-even grows with r (measured: ~8 edits at r ~ 85, C2; more than
// 30 at r ~ 380, C4).  INCR_MAX is the capacity of the change lists.
#ifndef PQP_INCR_MAX
#define PQP_INCR_MAX 32
#endif
constexpr int INCR_MAX = PQP_INCR_MAX;
constexpr int PARK_DOUBLES = 48;
#ifndef PQP_INCR_BASE
#define PQP_INCR_BASE 8
#endif
__host__ __device__ inline int
incr_max(int r)
{
#ifndef PQP_INCR_DIV
#define PQP_INCR_DIV 10
#endif
  const int v = r / PQP_INCR_DIV;
  return v < PQP_INCR_BASE ? (PQP_INCR_BASE < INCR_MAX ? PQP_INCR_BASE : INCR_MAX) : (v > INCR_MAX ? INCR_MAX : v);
}

// Rows beyond the workgroup size: the stages with one thread per row walk them in chunks of NT rows in the 1024-thread
// kernels (problems above 1024 rows; reference dense/model.hpp:65-68 has no size limit).  PQP_CHUNK_ALL = 1 compiles the
// chunk loops into every kernel width -- the emulator build of tests/test_emu_parity.py, which then runs shapes wider
// than a deliberately narrow workgroup (PQP_TEST_NT_MAX) through them.
#ifndef PQP_CHUNK_ALL
#define PQP_CHUNK_ALL 0
#endif
constexpr int MAX_ROWS = 4096; // (the row limit of round 4; a batch is created under PQP_MAX_ROWS of pqp_host.hpp: nothing on the device depends on either)
// the persistent slot list packs (constraint id + 1) of a slot in bits 0-15 and the active_set_up / active_set_low
// flags of constraint i in bits 16-17 of act[i] (solver, backward, pqp_batch_get_schur_factor)
static_assert(MAX_ROWS + 1 < (1 << 16), "act[] packs constraint ids into 16 bits");
__host__ __device__ inline int
act_pack(int cid, int flags)
{
  return (cid + 1) | ((flags & 3) << 16);
}
__host__ __device__ inline int
act_cid(int packed) // -1: hole / no slot
{
  return (packed & 0xffff) - 1;
}
__host__ __device__ inline int
act_flags(int packed)
{
  return (packed >> 16) & 3;
}

// signature part of the diagonal-structure test (Solver::dm; the per-QP part is State::c_diag, set by the set-up kernel)
__host__ __device__ inline bool
diag_structure_signature(int hessian, int n_eq, int n_in, int box)
{
  return hessian != PQP_HESSIAN_DENSE && n_eq == 0 && !(n_in > 0 && box != 0);
}

struct Dims
{
  int n, n_eq, n_in; // problem sizes
  int nc;            // n_in (+ n if box constraints)
  int nd;            // n_eq + nc   : capacity of the dual block
  int ntot;          // n + nd
  int box;           // box constraints present
  int hessian;       // pqp_hessian_type
  int backend;       // resolved pqp_dense_backend (PrimalDualLDLT / PrimalLDLT)
  int _pad;
};

// commands consumed by the setup kernel (reference dense/wrapper.hpp init/update)
enum
{
  CMD_NONE = 0,
  CMD_INIT = 1,
  CMD_UPDATE = 2,
  CMD_CLEANUP = 3
};
struct Cmd
{
  int op;
  int preconditioner; // compute_preconditioner (init) / update_preconditioner (update)
  int matrices_given; // update: H, A or C given  (helpers.hpp:466-468)
  int _pad;
  double rho, mu_eq, mu_in, min_eig; // NaN == nullopt
};

// per-QP persistent scalar state (reference dense/workspace.hpp flags + Ruiz c)
struct State
{
  int dirty;
  int refactorize;
  int proximal_parameter_update;
  int is_initialized;
  int n_c;          // active inequalities kept for WARM_START_WITH_PREVIOUS_RESULT
  int factor_valid; // primal block, Z and G in HBM match (model, rho)
  int ls_valid;     // the Schur factor in HBM matches (mu_eq_fact, mu_in_fact, active list)
  int n_slots;      // inequality slots of that factor (n_c live ones + the holes deletions left)
  int c_diag;       // C has no off-diagonal entry and n_in == dim (e.g. bounds passed as C = I): set by init / update
  int ls_edited;    // that factor has taken rank-1 edits since its last full factorisation (refinement fallback, solver.hpp:474-532)
  int scaled_valid; // H_s, A_s, C_s (both orientations) in HBM are the equilibrated model: written by init / update, read-only afterwards
  int _pad2;
  double ruiz_c;
  double dual_feasibility_rhs_2;
  double correction_guess_rhs_g;
  double mu_eq_fact, mu_in_fact, rho_fact;
};

// per-QP statistics (device cycles per phase, event counts)
enum
{
  ST_CYC_TOTAL = 0,
  ST_CYC_SCALE,
  ST_CYC_FACTOR_H,
  ST_CYC_ZG,
  ST_CYC_SCHUR,
  ST_CYC_KKT_SOLVE,
  ST_CYC_RESIDUAL,
  ST_CYC_LINESEARCH,
  ST_CYC_GLOBAL_RES,
  ST_CYC_NEWTON_MISC,
  ST_N_NEWTON,
  ST_N_SCHUR_FACT,
  ST_N_NEW_ROWS,
  ST_N_KKT_SOLVES,
  ST_N_LS_BREAKPOINTS,
  ST_N_ACTIVE_FINAL,
  ST_CYC_F_LOAD,      // blocked Schur factorisation: gather of S = M_J + G_JJ
  ST_CYC_F_UPDATE,    //   its LDL^T on the matrix cores
  ST_CYC_F_PANEL,
  ST_CYC_F_WRITEBACK, //   the row-wise inverse W_S
  ST_CYC_F_TINV,
  ST_CYC_S_GATHER,
  ST_CYC_SOLVE_LDLT,
  ST_N_SCHUR_BLOCKED, // Schur factorisations that took the blocked (HBM-resident) MFMA path
  ST_N_APPEND,        // rows appended to the inverse Schur factor (reference insert_block_at)
  ST_N_DELETE,        // rows deleted from it (reference delete_at)
  ST_BYTES_ENGINE,    // compulsory HBM bytes of the engine: every matrix pass priced at its size
  ST_N_REFACTORIZE,   // refinement fallbacks that rebuilt an edited factor (reference solver.hpp:474-532)
  ST_CYC_LS_EVAL,     // line search: phi'(alpha) at every breakpoint (sub-phase of ST_CYC_LINESEARCH)
  ST_CYC_CERT,        // infeasibility certificates (sub-phase of ST_CYC_NEWTON_MISC)
  ST_CYC_UPDATE,      // iterate update + inner stopping criterion (sub-phase of ST_CYC_NEWTON_MISC)
  ST_WALL_TICKS,      // residency of the QP's workgroup in constant-rate ticks (wall_clock64): info.solve_time
  ST_FLOPS_FACT,      // sum over the LDL^T factorisations of the solve (primal block, dual Schur block / P_J) of m_f^3 / 3
  ST_COUNT
};

// device view of one batch (all pointers are HBM; QP q starts at ptr + q*stride)
struct Batch
{
  Dims d;
  long B;
  // unscaled model (row-major), reference dense/model.hpp
  double *H, *g, *A, *b, *C, *u, *l, *u_box, *l_box;
  // equilibrated copies + both orientations of A and C
  double *Hs, *gs, *As, *ATs, *bs, *Cs, *CTs, *us, *ls, *ubs, *lbs, *is;
  double* delta; // ntot (Ruiz cumulative scaling, ruiz.hpp:319)
  // results (reference results.hpp)
  double *x, *y, *z, *se, *si;
  pqp_info* info;
  const pqp_settings* settings;
  State* state;
  const Cmd* cmd;
  // block factorisation workspace
  double *F;      // n x n   mirrored LDL^T of H_s + rho I
  double *WL, *WU; // n x n  L^{-1} and its transpose
  double *dF;     // n
  double *Zr;     // nd x n  row cid = L^{-1} b_cid
  double *Zc;     // n x nd  transpose of Zr
  double *G;      // nd x nd Gram matrix  Z^T D^{-1} Z over all constraints (build_ZG)
  double *WS;     // nd x nd W_S = L_S^{-1} (row-major, lower, unit diagonal) of M_J + G_JJ = L_S D_S L_S^T, slot order
  double *LS;     // nd x nd scratch of the blocked factorisation (Schur blocks beyond the register-resident path)
  double *dS;     // nd     D_S
  int* act;       // nc      active list kept across solves
  long long* stats; // ST_COUNT per QP
  double wall_us_per_tick; // microseconds per wall_clock64() tick of this device (Info timings)
  // traffic attribution harness (instrumented build only, PQP_REPEAT_PHASE / PQP_REPEAT_COUNT at pqp_batch_create):
  // the idempotent phase `rep_phase` (1 line search, 2 KKT residual, 3 first KKT solve of a step, 4 Schur
  // re-factorisation, 5 global residuals, 6 primal block + Z / G) runs rep_count times per occurrence, so that
  // the difference of the PMC byte counters against the plain run is that phase's HBM traffic
  int rep_phase, rep_count;
  // Host-resident results (pqp_batch_enable_host_results): pinned, device-mapped mirrors [B][...] of x, y, z, se, si
  // and Info that the epilogue of the solve writes beside the device arrays -- every workgroup pushes its own QP
  // over the host link as it finishes, so the results are on the host when the launch's event fires, with no
  // device-to-host copy behind the kernel (reference parallel/qp_solve.hpp:33-37: results are host members when
  // solve_in_parallel returns).  Null = off.
  double *hx, *hy, *hz, *hse, *hsi;
  pqp_info* hinfo;
  // settings.verbose: the per-iteration lines of the reference's report (solver.hpp:1478-1485 outer, :1021-1027
  // inner) as records of 8 doubles, written by thread 0 of the QP's workgroup and printed / handed out by the host
  // after the launch (pqp_batch_get_trace).  trace_slot[q] = slab slot of QP q (-1: not verbose); slot layout:
  // record 0 = { records written, records dropped, ... }, then the lines.  Null = no verbose QP in this launch.
  double* trace;
  const int* trace_slot;
  int trace_cap;
};

// QPLayer backward (reference dense/compute_ECJ.hpp): inputs and outputs of one launch.
// `ld` is [count][n + n_eq + n_in] (dL/dx, dL/dy, dL/dz of the QPs first .. first+count-1);
// the seven outputs are batch-major arrays over the WHOLE batch ([B][...]).
struct BackwardArgs
{
  const double* ld;
  double eps, rho_new, mu_new;
  double *dL_dH, *dL_dg, *dL_dA, *dL_db, *dL_dC, *dL_du, *dL_dl;
  long first;
  const int* order; // optional (pqp_batch_backward_subset): workgroup i works on QP order[i]; row i of `ld` is its loss derivative
};

// LDS carve-up -------------------------------------------------------------------------
// The ~45 per-QP vectors are grouped by length class (n, n_eq, n_c, n_in, n_d) so that every
// LDS address is  base + (slot * class_length + class_offset)  : twelve scalars describe the
// whole layout and each pointer is rebuilt where it is used (two scalar instructions behind an
// optimisation barrier).  Holding 45 hoisted pointers instead overflowed the scalar register
// file: a fifth of the kernel's instructions were v_readlane reloads of spilled scalars.
// The Z / G build of the 512- / 1024-thread kernels runs LDS-tiled (build_ZG); those kernels order their per-QP vectors and
// their prologue for it.  The 256-thread kernels keep the layout and the prologue of rounds 1-3 (their per-wavefront Z / G
// build is faster for their small, L2-resident operands, and a C1-size launch is sensitive to every instruction of the
// prologue: profiles/r04_ab_zg_lds_tiled.txt).
#ifndef PQP_VECTORS_IN_HBM
#define PQP_VECTORS_IN_HBM 0 // 1 in the translation unit of pqp_solve_hbm_kernel: its "LDS" is an HBM slice, staging through it gains nothing
#endif
#define ZG_LDS_TILED(nt) ((nt) >= 512 && !PQP_VECTORS_IN_HBM)
template<int NT>
struct Lds
{
  static constexpr int SH = ZG_LDS_TILED(NT) ? 2 : 0;   // dF, t1 in front of the other n-vectors ...
  static constexpr int DF = ZG_LDS_TILED(NT) ? 0 : 14;  // ... or behind them
  lptr base;
  int n, ne, nc, ni, nd, tmax, nt, plen;
  int o_ne, o_nc, o_ni, o_nd, o_t2; // class offsets in doubles (class n starts at 0)
  __device__ __forceinline__ lptr at(int off) const
  {
    // the barrier is on a VECTOR register: LDS addresses are vector operands anyway, and a
    // vector constraint is legal whatever the compiler thinks of the value's uniformity
    PQP_OPAQUE_VECTOR(off);
    return base + off;
  }
#define PQP_LVEC(name, cls_off, len, slot) \
  __device__ __forceinline__ lptr name() const { return at((cls_off) + (slot) * (len)); }
  // (dF and t1 first: the factorisation of the primal block and the Z / G build own them, and everything behind
  // them up to the end of `part` -- every other per-QP vector, none of which is loaded before that prologue has run
  // -- is the staging area of the LDS-tiled GEMMs: stage())
  PQP_LVEC(dF, 0, n, DF)
  PQP_LVEC(t1, 0, n, DF + 1)
  PQP_LVEC(x, 0, n, SH + 0)
  PQP_LVEC(xp, 0, n, SH + 1)
  PQP_LVEC(gs, 0, n, SH + 2)
  PQP_LVEC(ubs, 0, n, SH + 3)
  PQP_LVEC(lbs, 0, n, SH + 4)
  PQP_LVEC(isc, 0, n, SH + 5)
  PQP_LVEC(dx, 0, n, SH + 6)
  PQP_LVEC(rx, 0, n, SH + 7)
  PQP_LVEC(ex, 0, n, SH + 8)
  PQP_LVEC(Hdx, 0, n, SH + 9)
  PQP_LVEC(ATdy, 0, n, SH + 10)
  PQP_LVEC(CTdz, 0, n, SH + 11)
  PQP_LVEC(CTzin, 0, n, SH + 12)
  PQP_LVEC(dres, 0, n, SH + 13)
  PQP_LVEC(y, o_ne, ne, 0)
  PQP_LVEC(yp, o_ne, ne, 1)
  PQP_LVEC(bs, o_ne, ne, 2)
  PQP_LVEC(dy, o_ne, ne, 3)
  PQP_LVEC(Adx, o_ne, ne, 4)
  PQP_LVEC(se, o_ne, ne, 5)
  PQP_LVEC(z, o_nc, nc, 0)
  PQP_LVEC(zp, o_nc, nc, 1)
  PQP_LVEC(dz, o_nc, nc, 2)
  PQP_LVEC(Cdx, o_nc, nc, 3)
  PQP_LVEC(si, o_nc, nc, 4)
  PQP_LVEC(rup, o_nc, nc, 5)
  PQP_LVEC(zfull, o_nc, nc, 6)
  PQP_LVEC(us, o_ni, ni, 0)
  PQP_LVEC(ls, o_ni, ni, 1)
  PQP_LVEC(rd, o_nd, nd, 0)
  PQP_LVEC(ed, o_nd, nd, 1)
  PQP_LVEC(sd, o_nd, nd, 2)
  PQP_LVEC(dS, o_nd, nd, 3)
#undef PQP_LVEC
  __device__ __forceinline__ lptr stage() const { return at(2 * n); } // (tiled layouts only) zg_stage_doubles(nt) doubles, see lds_part_len
  __device__ __forceinline__ int part_len() const { return plen; }
  __device__ __forceinline__ int red_len() const { return 2 * RED_VALS * (nt / WAVE) + 8; }
  __device__ __forceinline__ lptr t2() const { return at(o_t2); }
  __device__ __forceinline__ lptr part() const { return at(o_t2 + tmax); }
  __device__ __forceinline__ lptr red() const { return at(o_t2 + tmax + part_len()); }
  __device__ __forceinline__ lptr top() const { return at(o_t2 + tmax + part_len() + red_len()); }
  __device__ __forceinline__ PQP_LDS long long* stat() const
  {
    return (PQP_LDS long long*)at(o_t2 + tmax + part_len() + red_len() + TOP_DOUBLES + PARK_DOUBLES);
  }
  __device__ __forceinline__ lptr park() const { return at(o_t2 + tmax + part_len() + red_len() + TOP_DOUBLES); }
  __device__ __forceinline__ liptr ints(int off) const
  {
    liptr q = (liptr)(base + (o_t2 + tmax + part_len() + red_len() + TOP_DOUBLES + PARK_DOUBLES + ST_COUNT + 1));
    PQP_OPAQUE_VECTOR(off);
    return q + off;
  }
  __device__ __forceinline__ liptr slot_of() const { return ints(0); }
  __device__ __forceinline__ liptr act() const { return ints(nc); }
  __device__ __forceinline__ liptr iscr() const { return ints(2 * nc); } // n_d ints of scratch
  __device__ __forceinline__ liptr aflags() const { return ints(2 * nc + nd); }
  __device__ __forceinline__ liptr icnt() const { return ints(3 * nc + nd); }
  // constraint ids leaving / entering the active set in an incremental change (2 * INCR_MAX ints)
  __device__ __forceinline__ liptr chg() const { return ints(3 * nc + nd + nt / WAVE + 16); }
};

// (Measured and removed: H_s v from the lower triangle of H_s only -- half the bytes of a pass, slower at every batch
// size and kernel width, profiles/r03_ab_hess_lower*.txt -- and the gemv pair over A_s / C_s and their transposed
// copies in the wide kernels, profiles/r03_ab_dual_pass_wide.txt: every kernel takes a product pair from ONE pass.)
// `part`: cross-wavefront scratch of gemv and of gemv_dual (NW * min(n, 128) doubles)
__host__ __device__ inline int
part_doubles(int nt, int tmax, int n)
{
  const int a = gemv_part_len(nt, tmax);
  const int b = gemv_dual_part_len(nt, n);
  return a > b ? a : b;
}

// length of `part`: what gemv / gemv_dual need, padded so that the staging area of the Z / G build (from behind t1 to
// the end of `part`) holds zg_stage_doubles(nt) doubles whatever the problem sizes
__host__ __device__ inline int
lds_part_len(int n, int ne, int nc, int ni, int nd, int nt)
{
  const int tmax = nd > n ? nd : n;
  const int need = part_doubles(nt, tmax, n);
  const int behind_t1 = 14 * n + 6 * ne + 7 * nc + 2 * ni + 4 * nd + tmax; // ... up to the start of `part`
  const int pad = ZG_LDS_TILED(nt) ? zg_stage_doubles(nt) - behind_t1 : 0;
  return need > pad ? need : pad;
}

__host__ __device__ inline size_t
lds_doubles(const Dims& d, int nt)
{
  size_t n = d.n, ne = d.n_eq, nc = d.nc, nd = d.nd;
  size_t tmax = nd > n ? nd : n;
  size_t s = 0;
  s += 2 * (n + ne + nc);                // x y z xp yp zp
  s += n + ne + 2 * d.n_in + 3 * n;      // gs bs us ls ubs lbs isc
  s += n + ne + nc;                      // dx dy dz
  s += 2 * (n + nd);                     // rx rd ex ed
  s += nd;                               // sd
  s += n + ne + nc + 3 * n;              // Hdx Adx Cdx ATdy CTdz CTzin
  s += n + ne + 2 * nc;                  // dres se si rup
  s += n + nd;                           // dF dS
  s += n + tmax + nc;                    // t1 t2 zfull
  s += lds_part_len(d.n, d.n_eq, d.nc, d.n_in, d.nd, nt); // part
  s += 2 * RED_VALS * (nt / WAVE) + 8;   // red
  s += TOP_DOUBLES;                      // top
  s += PARK_DOUBLES;                     // outer-loop scalars parked during the Newton loop
  s += ST_COUNT + 1;                     // stat (long long) + the time mark
  return s;
}
__host__ __device__ inline size_t
lds_bytes(const Dims& d, int nt)
{
  size_t ints = (size_t)d.nc * 3 + d.nd + nt / WAVE + 16 + 2 * INCR_MAX;
  return lds_doubles(d, nt) * sizeof(double) + ints * sizeof(int) + 64;
}

template<int NT>
__device__ __forceinline__ void
lds_carve(Lds<NT>& L, lptr base, const Dims& d, int nt)
{
  // every member goes through v_readfirstlane HERE, with all lanes active, so that the offsets
  // built from them later are scalar registers whatever the divergence at the point of use
  L.base = base;
  L.n = uni(d.n);
  L.ne = uni(d.n_eq);
  L.nc = uni(d.nc);
  L.ni = uni(d.n_in);
  L.nd = uni(d.nd);
  L.tmax = uni(d.nd > d.n ? d.nd : d.n);
  L.nt = nt;
  L.plen = uni(lds_part_len(d.n, d.n_eq, d.nc, d.n_in, d.nd, nt));
  L.o_ne = uni(16 * L.n);
  L.o_nc = uni(L.o_ne + 6 * L.ne);
  L.o_ni = uni(L.o_nc + 7 * L.nc);
  L.o_nd = uni(L.o_ni + 2 * L.ni);
  L.o_t2 = uni(L.o_nd + 4 * L.nd);
}

// Per-QP HBM pointers, recomputed from the kernel argument on demand (a few
// scalar instructions) instead of being held in ~90 registers.
struct QpRef
{
  const Batch& b;
  const long q;
  __device__ __forceinline__ QpRef(const Batch& b_, long q_)
    : b(b_)
    , q(q_)
  {
  }
  __device__ __forceinline__ long n() const { return b.d.n; }
  __device__ __forceinline__ long ne() const { return b.d.n_eq; }
  __device__ __forceinline__ long ni() const { return b.d.n_in; }
  __device__ __forceinline__ long nc() const { return b.d.nc; }
  __device__ __forceinline__ long nd() const { return b.d.nd; }
  // Every pointer is rebuilt from (base, q) where it is used: `lq()` hands out q through an
  // opaque scalar move so that the compiler neither hoists the ~35 per-QP pointers out of the
  // solver's loops nor keeps them alive across them (they would occupy ~70 SGPRs for the whole
  // kernel and push the rest into spills).
  __device__ __forceinline__ long lq() const
  {
    long v = q;
    PQP_OPAQUE_SCALAR(v);
    return v;
  }
#define PQP_PTR(name, stride) \
  __device__ __forceinline__ gptr name() const { return (gptr)(b.name + lq() * (stride)); }
  PQP_PTR(H, n() * n())
  PQP_PTR(g, n())
  PQP_PTR(A, ne() * n())
  PQP_PTR(C, ni() * n())
  PQP_PTR(u, ni())
  PQP_PTR(l, ni())
  PQP_PTR(u_box, n())
  PQP_PTR(l_box, n())
  PQP_PTR(Hs, n() * n())
  PQP_PTR(gs, n())
  PQP_PTR(As, ne() * n())
  PQP_PTR(ATs, ne() * n())
  PQP_PTR(bs, ne())
  PQP_PTR(Cs, ni() * n())
  PQP_PTR(CTs, ni() * n())
  PQP_PTR(us, ni())
  PQP_PTR(ls, ni())
  PQP_PTR(ubs, n())
  PQP_PTR(lbs, n())
  PQP_PTR(is, n())
  PQP_PTR(delta, (long)b.d.ntot)
  PQP_PTR(x, n())
  PQP_PTR(y, ne())
  PQP_PTR(z, nc())
  PQP_PTR(se, ne())
  PQP_PTR(si, nc())
  PQP_PTR(F, n() * n())
  PQP_PTR(WL, n() * n())
  PQP_PTR(WU, n() * n())
  PQP_PTR(dF, n())
  PQP_PTR(Zr, nd() * n())
  PQP_PTR(Zc, nd() * n())
  PQP_PTR(G, nd() * nd())
  PQP_PTR(WS, nd() * nd())
  PQP_PTR(LS, nd() * nd())
  PQP_PTR(dS, nd())
#undef PQP_PTR
  // `b` is also the name of the equality right-hand side: spelled out
  __device__ __forceinline__ gptr bvec() const { return (gptr)(b.b + q * ne()); }
  __device__ __forceinline__ PQP_GLOBAL int* act() const { return (PQP_GLOBAL int*)(b.act + q * nc()); }
  __device__ __forceinline__ PQP_GLOBAL long long* stats() const
  {
    return (PQP_GLOBAL long long*)(b.stats + q * ST_COUNT);
  }
  __device__ __forceinline__ pqp_info* info() const { return b.info + q; }
  __device__ __forceinline__ State* state() const { return b.state + q; }
  __device__ __forceinline__ const pqp_settings& settings() const { return b.settings[q]; }
  // Ruiz scalings
  __device__ __forceinline__ cgptr dlt_x() const { return delta(); }
  __device__ __forceinline__ cgptr dlt_eq() const { return delta() + n(); }
  __device__ __forceinline__ cgptr dlt_in() const { return delta() + n() + ne(); }
  __device__ __forceinline__ cgptr dlt_box() const { return delta() + n() + ne() + ni(); }
};

__device__ __forceinline__ bool
absent(double v)
{
  return v != v;
}

// working copy of pqp_info whose doubles live in scalar registers (see UD)
struct UInfo
{
  UD mu_eq, mu_eq_inv, mu_in, mu_in_inv, rho, nu;
  long iter, iter_ext, mu_updates, rho_updates;
  UD setup_time, solve_time, run_time, objValue, pri_res, dua_res, duality_gap, iterative_residual,
    minimal_H_eigenvalue_estimate;
  int status;
  __device__ __forceinline__ void load(const pqp_info& o)
  {
    mu_eq = o.mu_eq;
    mu_eq_inv = o.mu_eq_inv;
    mu_in = o.mu_in;
    mu_in_inv = o.mu_in_inv;
    rho = o.rho;
    nu = o.nu;
    iter = uni((long)o.iter);
    iter_ext = uni((long)o.iter_ext);
    mu_updates = uni((long)o.mu_updates);
    rho_updates = uni((long)o.rho_updates);
    setup_time = o.setup_time;
    solve_time = o.solve_time;
    run_time = o.run_time;
    objValue = o.objValue;
    pri_res = o.pri_res;
    dua_res = o.dua_res;
    duality_gap = o.duality_gap;
    iterative_residual = o.iterative_residual;
    minimal_H_eigenvalue_estimate = o.minimal_H_eigenvalue_estimate;
    status = uni((int)o.status);
  }
  __device__ __forceinline__ void store(pqp_info& o) const
  {
    o.mu_eq = mu_eq;
    o.mu_eq_inv = mu_eq_inv;
    o.mu_in = mu_in;
    o.mu_in_inv = mu_in_inv;
    o.rho = rho;
    o.nu = nu;
    o.iter = iter;
    o.iter_ext = iter_ext;
    o.mu_updates = mu_updates;
    o.rho_updates = rho_updates;
    o.setup_time = setup_time;
    o.solve_time = solve_time;
    o.run_time = run_time;
    o.objValue = objValue;
    o.pri_res = pri_res;
    o.dua_res = dua_res;
    o.duality_gap = duality_gap;
    o.iterative_residual = iterative_residual;
    o.minimal_H_eigenvalue_estimate = minimal_H_eigenvalue_estimate;
    o.status = status;
    o._pad = 0;
  }
};

// reference results.hpp:157-174
template<typename Info>
__device__ __forceinline__ void
cleanup_statistics(Info& i)
{
  i.run_time = 0;
  i.setup_time = 0;
  i.solve_time = 0;
  i.objValue = 0.;
  i.iter = 0;
  i.iter_ext = 0;
  i.mu_updates = 0;
  i.rho_updates = 0;
  i.pri_res = 0.;
  i.dua_res = 0.;
  i.duality_gap = 0.;
  i.iterative_residual = 0.;
  i.status = PQP_MAX_ITER_REACHED;
}
// reference results.hpp:175-194
template<typename Info>
__device__ __forceinline__ void
cold_start(Info& i, const pqp_settings& s)
{
  i.nu = 1.;
  i.rho = s.default_rho;
  i.mu_eq = s.default_mu_eq;
  i.mu_eq_inv = 1.0 / i.mu_eq;
  i.mu_in = s.default_mu_in;
  i.mu_in_inv = 1.0 / i.mu_in;
  i.minimal_H_eigenvalue_estimate = s.default_H_eigenvalue_estimate;
  cleanup_statistics(i);
}

// reference workspace.hpp:330-377 (scalar part; LDS vectors are reset where used)
__device__ __forceinline__ void
work_cleanup_flags(State& w)
{
  w.dirty = 0;
  w.refactorize = 0;
  w.proximal_parameter_update = 0;
  w.is_initialized = 0;
  w.n_c = 0;
  w.factor_valid = 0;
  w.ls_valid = 0;
  w.n_slots = 0;
  w.ls_edited = 0;
}

// ---------------------------------------------------------------------------
// Equilibrated copies.  Restates what reference preconditioner/ruiz.hpp:205-307
// (execute) and :442-511 (re-apply) leave in the workspace, from the cumulative
// scaling S (LDS, ntot) and c:  H_s = c S_x H S_x, A_s = S_eq A S_x, ...; both
// orientations of A_s and C_s are written for the solve kernel.
// ---------------------------------------------------------------------------
// `tile` (TILED = true, the set-up kernel): 32 x 33 doubles of LDS through which the transposed copies A_s^T / C_s^T
// are written in coalesced rows -- a thread per element of A_s writes its transposed position 8 bytes at a time, one
// memory transaction each.  The solve kernels keep the plain form (their call rewrites the vectors only).
template<int NT, bool TILED = false>
__device__ PQP_CALL void
write_scaled(const Batch& batch, long q, clptr S, double c, bool clamp, bool diag_only = false, bool matrices = true,
             lptr tile = nullptr)
{
  const QpRef P(batch, q);
  const Dims& d = batch.d;
  const int n = d.n, ne = d.n_eq, ni = d.n_in;
  clptr Sx = S;
  clptr Se = S + n;
  clptr Si = S + n + ne;
  clptr Sb = S + n + ne + ni;
  // matrices == false: only the equilibrated VECTORS are rewritten.  The scaled matrices are a function of
  // (model, delta, c) alone, nothing ever modifies them between two set-ups, and a dirty re-solve re-applies the
  // SAME stored equilibration (solver.hpp:1192-1214): the copies init / update left in HBM are already the
  // bits this pass would write.  The vectors differ (the re-solve takes u, l unclamped).
  if (!matrices) {
  } else if (diag_only) {
    // diagonal structure (see Solver::dm): only the diagonals of H and C carry information; the
    // dense copies keep the zeros the init-time pass wrote, the compact ones feed the solver
    cgptr H = P.H();
    gptr Hs = P.Hs(), hd = P.F();
    for (int k = threadIdx.x; k < n; k += NT) {
      const double h = H[(long)k * n + k];
      const double v = (d.hessian == PQP_HESSIAN_DIAGONAL) ? h * Sx[k] * Sx[k] * c : h * c;
      Hs[(long)k * n + k] = v;
      hd[k] = v;
    }
    if (ni > 0) {
      cgptr C = P.C();
      gptr Cs = P.Cs(), cd = P.CTs();
      for (int k = threadIdx.x; k < n; k += NT) {
        const double v = Si[k] * C[(long)k * n + k] * Sx[k];
        Cs[(long)k * n + k] = v;
        cd[k] = v;
      }
    }
  } else {
  {
    cgptr H = P.H();
    gptr Hs = P.Hs();
    if (d.hessian == PQP_HESSIAN_DENSE) {
      for (int o = threadIdx.x; o < n * n; o += NT) {
        int r = o / n, k = o - r * n;
        Hs[o] = Sx[r] * H[o] * Sx[k] * c;
      }
    } else {
      for (int o = threadIdx.x; o < n * n; o += NT) {
        int r = o / n, k = o - r * n;
        double v = H[o];
        if (d.hessian == PQP_HESSIAN_DIAGONAL)
          v = (r == k) ? v * Sx[r] * Sx[r] * c : v * c;
        Hs[o] = v;
      }
    }
  }
  if constexpr (TILED) {
    constexpr int T = 32;
    for (int which = 0; which < 2; ++which) {
      const int R = which ? ni : ne;
      cgptr M = which ? P.C() : P.A();
      gptr Ms = which ? P.Cs() : P.As(), MTs = which ? P.CTs() : P.ATs();
      clptr Sr = which ? Si : Se;
      for (int r0 = 0; r0 < R; r0 += T)
        for (int k0 = 0; k0 < n; k0 += T) {
          for (int e = threadIdx.x; e < T * T; e += NT) {
            const int rr = e / T, kk = e - rr * T;
            const int r = r0 + rr, k = k0 + kk;
            if (r < R && k < n) {
              const long o = (long)r * n + k;
              const double v = Sr[r] * M[o] * Sx[k];
              Ms[o] = v;
              tile[rr * (T + 1) + kk] = v;
            }
          }
          __syncthreads();
          for (int e = threadIdx.x; e < T * T; e += NT) {
            const int kk = e / T, rr = e - kk * T;
            const int r = r0 + rr, k = k0 + kk;
            if (r < R && k < n)
              MTs[(long)k * R + r] = tile[rr * (T + 1) + kk];
          }
          __syncthreads();
        }
    }
  } else {
  {
    cgptr A = P.A();
    gptr As = P.As(), ATs = P.ATs();
    for (int o = threadIdx.x; o < ne * n; o += NT) {
      int r = o / n, k = o - r * n;
      double v = Se[r] * A[o] * Sx[k];
      As[o] = v;
      ATs[(long)k * ne + r] = v;
    }
  }
  {
    cgptr C = P.C();
    gptr Cs = P.Cs(), CTs = P.CTs();
    for (int o = threadIdx.x; o < ni * n; o += NT) {
      int r = o / n, k = o - r * n;
      double v = Si[r] * C[o] * Sx[k];
      Cs[o] = v;
      CTs[(long)k * ni + r] = v;
    }
  }
  }
  }
  for (int k = threadIdx.x; k < n; k += NT)
    P.gs()[k] = P.g()[k] * Sx[k] * c;
  for (int k = threadIdx.x; k < ne; k += NT)
    P.bs()[k] = P.bvec()[k] * Se[k];
  for (int k = threadIdx.x; k < ni; k += NT) {
    double uu = P.u()[k], ll = P.l()[k];
    if (clamp) { // helpers.hpp:628-637
      uu = (uu <= 1.E20) ? uu : 1.E20;
      ll = (ll >= -1.E20) ? ll : -1.E20;
    }
    P.us()[k] = uu * Si[k];
    P.ls()[k] = ll * Si[k];
  }
  if (d.box) {
    for (int k = threadIdx.x; k < n; k += NT) {
      double uu = P.u_box()[k], ll = P.l_box()[k];
      uu = (uu <= 1.E20) ? uu : 1.E20; // helpers.hpp:638-649
      ll = (ll >= -1.E20) ? ll : -1.E20;
      P.ubs()[k] = uu * Sb[k];
      P.lbs()[k] = ll * Sb[k];
      P.is()[k] = Sx[k] * Sb[k];
    }
  }
  for (int k = threadIdx.x; k < d.ntot; k += NT)
    P.delta()[k] = S[k];
  __syncthreads();
}

// ---------------------------------------------------------------------------
// Ruiz equilibration, reference preconditioner/ruiz.hpp:29-311.  The cumulative
// scaling S stays in LDS and every sweep takes its norms from the UNSCALED model
// with S applied on the fly (read-only coalesced passes instead of a read-modify-write
// of H, A and C).  One pass per sweep over each matrix:
//   * A and C: a wavefront per row gives the row norm, and the same loads feed per-lane column maxima that meet
//     across the wavefronts in LDS (`cpart`, NT / 64 x 256 doubles; columns in blocks of 256);
//   * H: a thread per column; with a dense H the cost scaling gamma of a sweep (ruiz.hpp:256-303) is a function of
//     the column maxima of H under the UPDATED scaling -- exactly what the next sweep's column pass computes first --
//     so it is taken there, one sweep late (c is not used inside the loop for a dense H), and by a pass of its own
//     only after the last sweep.
// (Round 3 read A and C twice and H twice per sweep: 3.5 ms per 2048 C2 QPs; the maxima are exact and the sums keep
// their order, so S and c are the same bits.)
// ---------------------------------------------------------------------------
template<int NT>
__device__ __forceinline__ double
ruiz_execute(const Batch& batch, long q, const pqp_settings& st, lptr S, lptr dl, Reducer<NT>& R, lptr cpart)
{
  const QpRef P(batch, q);
  const Dims& d = batch.d;
  const int n = d.n, ne = d.n_eq, ni = d.n_in, ntot = d.ntot;
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
  constexpr int NW = NT / WAVE;
  constexpr int CB = 4 * WAVE; // columns per block of the A / C pass
  double c = 1, gamma = 1;
  for (int k = threadIdx.x; k < ntot; k += NT) {
    S[k] = 1.0;
    dl[k] = 0.0; // LDLT_TEMP_VEC is zero-initialised (ruiz.hpp:75)
  }
  __syncthreads();
  lptr Sx = S;
  lptr Se = S + n;
  lptr Si = S + n + ne;
  lptr Sb = S + n + ne + ni;
  cgptr H = P.H();
  cgptr A = P.A();
  cgptr C = P.C();
  const bool infeas = st.primal_infeasibility_solving != 0;
  const bool dense = d.hessian == PQP_HESSIAN_DENSE;
  bool gamma_owed = false; // dense H: the cost scaling of the last sweep has not been taken yet
  long iter = 1;
  while (true) {
    double e = 0;
    for (int k = threadIdx.x; k < ntot; k += NT)
      e = fmax(e, fabs(1 - dl[k]));
    e = R.max(e);
    if (!(e > st.preconditioner_accuracy))
      break;
    if (iter == st.preconditioner_max_iter)
      break;
    ++iter;
    __syncthreads(); // (dl is rewritten below: everybody has read it)
    // A and C, once: row norms (raw maxima in dl[n + r] for now) and column maxima max_r S_r |M_rk| (in dl[k], k < n)
    for (int c0 = 0; c0 < n; c0 += CB) {
      double cm[4] = { 0.0, 0.0, 0.0, 0.0 };
      for (int r = wid; r < ne + ni; r += NW) {
        cgptr row = (r < ne) ? (A + (long)r * n) : (C + (long)(r - ne) * n);
        const double sr = (r < ne) ? Se[r] : Si[r - ne];
        double m = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = c0 + lane + WAVE * u;
          if (k < n) {
            const double v = fabs(row[k]);
            m = fmax(m, v * Sx[k]);
            cm[u] = fmax(cm[u], sr * v);
          }
        }
        if (!infeas) {
          m = wave_max(m);
          if (lane == 0)
            dl[n + r] = (c0 == 0) ? m : fmax(dl[n + r], m);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        cpart[wid * CB + lane + WAVE * u] = cm[u];
      __syncthreads();
      for (int j = threadIdx.x; j < CB && c0 + j < n; j += NT) {
        double a = cpart[j];
#pragma unroll
        for (int w = 1; w < NW; ++w)
          a = fmax(a, cpart[w * CB + j]);
        dl[c0 + j] = a;
      }
      __syncthreads();
    }
    if (ne + ni == 0) {
      for (int k = threadIdx.x; k < n; k += NT)
        dl[k] = 0.0;
      __syncthreads();
    }
    // columns: H (thread per column) joined with the maxima of A and C
    double acc = 0;
    for (int k = threadIdx.x; k < n; k += NT) {
      double m = 0;
      if (dense) {
        double hm = 0;
        for (int i = 0; i < n; ++i)
          hm = fmax(hm, Sx[i] * fabs(H[(long)i * n + k]));
        m = hm * Sx[k];
        acc += m; // = hm * Sx[k]: the term of the owed gamma
      } else if (d.hessian == PQP_HESSIAN_DIAGONAL) {
        m = fabs(H[(long)k * n + k]) * Sx[k] * Sx[k] * c;
      }
      if (ne + ni > 0)
        m = fmax(m, dl[k] * Sx[k]); // max(am, cm) * Sx[k] = max(am * Sx[k], cm * Sx[k]): rounding is monotone
      if (d.box)
        m = fmax(m, Sx[k] * Sb[k]);
      double aux = sqrt(m);
      dl[k] = (aux == 0.0) ? 1.0 : 1.0 / (aux + MACHINE_EPS);
    }
    if (dense && gamma_owed) { // the cost scaling of the PREVIOUS sweep (same S as this sweep's column pass)
      acc = R.sum(acc);
      gamma = 1 / fmax(1.0, acc / (double)n);
      c *= gamma;
    }
    // rows
    if (infeas) {
      for (int k = n + threadIdx.x; k < ntot; k += NT)
        dl[k] = 1.0;
    } else {
      for (int r = threadIdx.x; r < ne + ni; r += NT) {
        const double sr = (r < ne) ? Se[r] : Si[r - ne];
        const double aux = sqrt(dl[n + r] * sr);
        dl[n + r] = (aux == 0.0) ? 1.0 : 1.0 / (aux + MACHINE_EPS);
      }
      if (d.box)
        for (int k = threadIdx.x; k < n; k += NT)
          dl[n + ne + ni + k] = 1.0 / sqrt(Sx[k] * Sb[k] + MACHINE_EPS);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < ntot; k += NT)
      S[k] *= dl[k];
    __syncthreads();
    // cost scaling gamma (ruiz.hpp:256-303); NB: not applied to a Dense H
    if (dense) {
      gamma_owed = true;
    } else if (d.hessian == PQP_HESSIAN_DIAGONAL) {
      double mx = 0;
      for (int k = threadIdx.x; k < n; k += NT)
        mx = fmax(mx, fabs(H[(long)k * n + k]) * Sx[k] * Sx[k] * c);
      mx = R.max(mx);
      gamma = 1 / fmax(1.0, mx / (double)n);
      c *= gamma;
    } else {
      c *= gamma;
    }
  }
  if (dense && gamma_owed) {
    double acc = 0;
    for (int k = threadIdx.x; k < n; k += NT) {
      double hm = 0;
      for (int i = 0; i < n; ++i)
        hm = fmax(hm, Sx[i] * fabs(H[(long)i * n + k]));
      acc += hm * Sx[k];
    }
    acc = R.sum(acc);
    gamma = 1 / fmax(1.0, acc / (double)n);
    c *= gamma;
  }
  return c;
}

// ---------------------------------------------------------------------------
// setup kernel body: QP::init / QP::update / QP::cleanup
// (reference dense/wrapper.hpp:354-498, 723-807, 958-962; helpers.hpp:500-705)
// The host side has already mutated `settings` exactly as the reference does and
// copied the provided arrays into the model buffers.
// ---------------------------------------------------------------------------
__host__ __device__ inline size_t
setup_lds_bytes(const Dims& d, int nt)
{
  const int cpart = (nt / WAVE) * 4 * WAVE; // ruiz_execute's column partials; write_scaled's 32 x 33 transposition tile
  return (size_t)(2 * d.ntot + 2 * RED_VALS * (nt / WAVE) + 16 + (cpart > 33 * 32 ? cpart : 33 * 32)) * sizeof(double);
}

template<int NT>
__device__ __forceinline__ void
setup_body(const Batch& batch, long q, lptr lds_base)
{
  const Dims& d = batch.d;
  const QpRef P(batch, q);
  const Cmd cmd = batch.cmd[q];
  if (cmd.op == CMD_NONE)
    return;
  const pqp_settings& st = P.settings();
  State& W = *P.state();
  pqp_info& info = *P.info();
  const int n = d.n, ne = d.n_eq, nc = d.nc;
  lptr S = lds_base;
  lptr dl = S + d.ntot;
  lptr red = dl + d.ntot;
  Reducer<NT> R(red);

  if (cmd.op == CMD_CLEANUP) { // wrapper.hpp:958-962
    for (int k = threadIdx.x; k < n; k += NT)
      P.x()[k] = 0;
    for (int k = threadIdx.x; k < ne; k += NT) {
      P.y()[k] = 0;
      P.se()[k] = 0;
    }
    for (int k = threadIdx.x; k < nc; k += NT) {
      P.z()[k] = 0;
      P.si()[k] = 0;
    }
    if (threadIdx.x == 0) {
      cold_start(info, st);
      work_cleanup_flags(W);
    }
    return;
  }

  const bool is_init = (cmd.op == CMD_INIT) || !W.is_initialized; // wrapper.hpp:743-746
  const long long setup_t0 = wall_clock64();                      // wrapper.hpp:374-377
  __syncthreads();
  if (threadIdx.x == 0) {
    if (is_init) {
      W.refactorize = (st.initial_guess == PQP_WARM_START_WITH_PREVIOUS_RESULT) ? 1 : 0;
      W.proximal_parameter_update = 0;
    } else {
      W.refactorize = cmd.matrices_given ? 1 : 0; // helpers.hpp:466-468
      W.proximal_parameter_update = 0;
    }
    // helpers.hpp:678-705
    if (!absent(cmd.rho)) {
      info.rho = cmd.rho;
      W.proximal_parameter_update = 1;
    }
    if (!absent(cmd.mu_eq)) {
      info.mu_eq = cmd.mu_eq;
      info.mu_eq_inv = 1.0 / cmd.mu_eq;
      W.proximal_parameter_update = 1;
    }
    if (!absent(cmd.mu_in)) {
      info.mu_in = cmd.mu_in;
      info.mu_in_inv = 1.0 / cmd.mu_in;
      W.proximal_parameter_update = 1;
    }
    // helpers.hpp:174-189 (settings.default_rho already carries the shift)
    if (!absent(cmd.min_eig))
      info.minimal_H_eigenvalue_estimate = cmd.min_eig;
    info.rho = st.default_rho;
  }
  __syncthreads();
  const int precond = is_init ? (cmd.preconditioner ? 0 : 2) : (cmd.preconditioner ? 0 : 1);

  // helpers.hpp:522-572 : result / workspace reset per initial guess
  bool zero_xyz = false;
  {
    const int ig = st.initial_guess;
    const bool ppu = W.proximal_parameter_update != 0;
    const bool refac = W.refactorize != 0;
    __syncthreads();
    if (ig == PQP_EQUALITY_CONSTRAINED_INITIAL_GUESS || ig == PQP_NO_INITIAL_GUESS ||
        ig == PQP_WARM_START) {
      zero_xyz = true;
      if (threadIdx.x == 0) {
        if (ppu)
          cleanup_statistics(info);
        else
          cold_start(info, st);
        work_cleanup_flags(W);
      }
    } else if (ig == PQP_COLD_START_WITH_PREVIOUS_RESULT) {
      if (threadIdx.x == 0) {
        if (ppu)
          cleanup_statistics(info);
        else
          cold_start(info, st);
        work_cleanup_flags(W);
      }
    } else { // WARM_START_WITH_PREVIOUS_RESULT
      if (threadIdx.x == 0) {
        if (refac || ppu) {
          work_cleanup_flags(W);
          W.refactorize = 1;
        }
        cleanup_statistics(info);
      }
    }
  }
  if (zero_xyz) {
    for (int k = threadIdx.x; k < n; k += NT)
      P.x()[k] = 0;
    for (int k = threadIdx.x; k < ne; k += NT) {
      P.y()[k] = 0;
      P.se()[k] = 0;
    }
    for (int k = threadIdx.x; k < nc; k += NT) {
      P.z()[k] = 0;
      P.si()[k] = 0;
    }
  }
  // helpers.hpp:651
  {
    double m = 0;
    for (int k = threadIdx.x; k < n; k += NT)
      m = fmax(m, fabs(P.g()[k]));
    m = R.max(m);
    if (threadIdx.x == 0)
      W.dual_feasibility_rhs_2 = m;
  }
  // helpers.hpp:652-666 -> setup_equilibration (:298-329)
  double c;
  if (precond == 0) {
    c = ruiz_execute<NT>(batch, q, st, S, dl, R, red + (2 * RED_VALS * (NT / WAVE) + 16));
  } else {
    c = W.ruiz_c;
    for (int k = threadIdx.x; k < d.ntot; k += NT)
      S[k] = P.delta()[k];
    __syncthreads();
  }
  write_scaled<NT, true>(batch, q, S, c, true, false, true, red + (2 * RED_VALS * (NT / WAVE) + 16));
  {
    // structure detection for the diagonal fast path of the solve kernel (Solver::dm): no
    // off-diagonal entry in C with n_in == dim (bounds handed over as C = I, reference
    // utils/random_qp_problems.hpp:591-628), or no general inequality at all
    const int ni = d.n_in;
    double off = (ni == 0 || ni == n) ? 0.0 : 1.0;
    if (ni == n) {
      cgptr C = P.C();
      for (int o = threadIdx.x; o < ni * n; o += NT) {
        const int rr = o / n, k = o - rr * n;
        if (rr != k && C[o] != 0.0)
          off = 1.0;
      }
    }
    off = R.max(off);
    const bool dmode = d.hessian != PQP_HESSIAN_DENSE && ne == 0 && off == 0.0 && !(ni > 0 && d.box);
    if (dmode)
      write_scaled<NT>(batch, q, S, c, true, true); // compact diagonals beside the dense copies
    if (threadIdx.x == 0)
      W.c_diag = (off == 0.0) ? 1 : 0;
  }
  {
    double m = 0;
    for (int k = threadIdx.x; k < n; k += NT)
      m = fmax(m, fabs(P.g()[k] * S[k] * c));
    m = R.max(m);
    if (threadIdx.x == 0) {
      W.correction_guess_rhs_g = m;
      W.ruiz_c = c;
      W.scaled_valid = 1;
      if (is_init)
        W.is_initialized = 1;
      if (st.compute_timings) // wrapper.hpp:495-497, 804-806: microseconds spent in init / update
        info.setup_time = (double)(wall_clock64() - setup_t0) * batch.wall_us_per_tick;
    }
  }
}

// Dual Schur blocks beyond the register-resident path (more than 16 * SCHUR_MB rows, or a
// workgroup that is not 16 x 16): S = M_J + G_JJ is gathered into LS (slot -> constraint id in
// `sid`, -1 for a hole = identity row) here; the caller factorises it there on the matrix cores in the
// FULL layout (ldlt_factor_mfma) and inverts it row-wise into W_S (tri_inverse_mfma_rows).
template<int NT>
__device__ __forceinline__ void
schur_gather_blocked(cgptr G, gptr LS, int nd, int rr, int ne, double mu_eq, double mu_in, cliptr sid)
{
  // gathered loads are batched 8 deep ahead of the stores (G and LS are distinct buffers, but
  // the compiler cannot know and would serialise load/store pairs)
  for (int base = 0; base < rr * rr; base += 8 * NT) {
    double v[8];
    long dst[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int o = base + u * NT + threadIdx.x;
      dst[u] = -1;
      v[u] = 0;
      if (o < rr * rr) {
        int a = o / rr, b = o - a * rr;
        const int ca = sid[a], cb = sid[b];
        const bool live = (ca | cb) >= 0;
        v[u] = live ? G[(long)ca * nd + cb] : 0.0;
        if (a == b)
          v[u] = live ? v[u] + ((a < ne) ? mu_eq : mu_in) : 1.0;
        dst[u] = (long)a * nd + b;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (dst[u] >= 0)
        LS[dst[u]] = v[u];
  }
  __syncthreads();
}

// ===========================================================================
//                               SOLVE
// ===========================================================================
// SPEC = 1 compiles the solver for the commonest signature -- no box constraints, dense Hessian --
// with those two switches as compile-time constants (the box / diagonal / zero-Hessian branches and
// the scalars that feed them disappear from the hot kernel); SPEC = 0 keeps them at run time.
// SPEC = 2 is the diagonal-structure solver on its own (every QP of the launch has the structure `dm()` describes: the
// host checks the flags the set-up kernel left, pqp_launch_solve): no dense matrix code, no factorisation code, no
// PrimalLDLT engine in the kernel -- BASELINE.json configs[4] ("bandwidth-bound, no MFMA") gets a kernel that is
// nothing but its element-wise path, whose registers no longer depend on what the general kernel has to hold.
// (A two-kernel form of the solve -- factorisation prologue / iteration -- was measured in round 3: the iteration kernel
// alone spills half as many registers and runs at the same speed, profiles/r03_ab_split_solve.txt.  Removed.)
template<int NT, int SPEC = 0>
struct Solver
{
  __device__ __forceinline__ bool has_box() const { return SPEC == 1 ? false : (d.box != 0); }
  __device__ __forceinline__ int hess() const
  {
    if (SPEC == 1)
      return (int)PQP_HESSIAN_DENSE;
    if (SPEC == 2) // (never dense in diagonal structure)
      return d.hessian == PQP_HESSIAN_ZERO ? (int)PQP_HESSIAN_ZERO : (int)PQP_HESSIAN_DIAGONAL;
    return d.hessian;
  }
  const Batch& batch;
  const long q;
  const Dims d;
  const QpRef P;
  Lds<NT> L;
  Reducer<NT> R;
  const pqp_settings& st;
  UInfo info; // working copy (scalar registers), written back at exit
  int n_c;       // active inequality count
  int n_slots;   // inequality slots of the Schur factor: the n_c active ones + holes left by deletions
  int r;         // n_eq + n_slots : size of the dual block
  // Diagonal structure: H diagonal (or zero), no equality and every inequality row touching ONE
  // variable (box constraints, or C without off-diagonal entries, not both).  Then L = I, every row
  // of Z has one entry, the Gram matrix and the dual Schur block are DIAGONAL: nothing is factorised,
  // no matrix is read in the solve -- every step of the engine of section "dual Schur block" becomes
  // element-wise on vectors (BASELINE.json configs[4]: "bandwidth-bound, no MFMA").  Compact
  // diagonals: hd = diag(H_s) in the F buffer, cd = diag(C_s) in the C_s^T buffer, zd / gd (entry of
  // Z and of G per constraint) at the start of the Zr / G buffers.
  bool diag_mode;
  __device__ __forceinline__ bool dm() const { return SPEC == 1 ? false : (SPEC == 2 ? true : diag_mode); }
  // DenseBackend::PrimalLDLT (reference dense/solver.hpp:88-109, 171-227, 336-388; chosen by
  // dense_backend_choice when n_eq + n_c is large against dim): the DUAL block of the KKT matrix is
  // eliminated instead of the primal one,
  //     P_J = H_s + rho I + A_s^T A_s / mu_eq + C_J^T C_J / mu_in          (dim x dim, SPD)
  //     P_J x = b_x + B_J^T M^{-1} b_d ,     d = M^{-1} (B_J x - b_d),
  // so the factorisation stays dim x dim however many constraints are active.  P_J is kept in the
  // same inverse-factor form as the dual Schur block of the default engine (W = L^{-1} in the WL
  // buffer, D in L.dF): two chain-free mat-vecs per solve; an active-set change or a mu update
  // re-assembles P_J on the matrix cores (syrk_mfma) and re-factorises it.
  __device__ __forceinline__ bool pm() const { return SPEC != 0 ? false : (d.backend == PQP_BACKEND_PRIMAL_LDLT && !diag_mode); }
  __device__ __forceinline__ int dcol(int cid) const { return (cid < d.n_in) ? cid : cid - d.n_in; } // variable of constraint cid
  bool schur_dirty;       // the factor does not describe (active set, mu): re-factorise
  bool schur_incremental; // rows were appended / deleted since the last full factorisation
  bool aty_fresh; // L.ATdy / L.CTdz hold A^T y, C^T z of the current iterate (see global_primal_residual)
  bool iterate_zero; // x = y = z = 0 exactly (NO_INITIAL_GUESS before the first Newton loop): A x, C x, H x, A^T y, C^T z are zero vectors
  UD ruiz_c;
  UD dual_feasibility_rhs_2;
  bool nonfinite;

  __device__ __forceinline__ Solver(const Batch& b, long q_, lptr lds_base)
    : batch(b)
    , q(q_)
    , d(b.d)
    , P(b, q_)
    , R(nullptr)
    , st(b.settings[q_])
  {
    lds_carve(L, lds_base, d, NT);
    R = Reducer<NT>(L.red());
    nonfinite = false;
    n_c = 0;
    n_slots = 0;
    r = d.n_eq;
    schur_dirty = true;
    schur_incremental = false;
    diag_mode = false;
    iterate_zero = false;
  }
  __device__ __forceinline__ void set_diag_mode(const State& W)
  {
    diag_mode = (SPEC == 2) || ((SPEC == 0) && d.hessian != PQP_HESSIAN_DENSE && d.n_eq == 0 && W.c_diag != 0 &&
                                !(d.n_in > 0 && d.box != 0));
  }

  // phase timers / event counters: thread 0 only, accumulated in LDS
  // Phase timers and event counters are compiled in with -DPQP_STATS only (the instrumented build
  // libproxqp_hip_stats.so that bench.py uses OUTSIDE its timed region): in the product build they
  // are no-ops -- the ~150 thread-0 clock reads and LDS updates per Newton step cost the kernel a
  // third of its VGPR spills.  The total cycle count of a solve (ST_CYC_TOTAL, which the optional
  // longest-first dispatch order uses) and the final active-set size are always recorded.
  // (the time mark lives in LDS beside the counters -- slot ST_COUNT of the stat area -- and not
  // in a register pair that would stay live across every routine of the solver)
  __device__ __forceinline__ void tic()
  {
#ifdef PQP_STATS
    if (threadIdx.x == 0)
      L.stat()[ST_COUNT] = clock64();
#endif
  }
  __device__ __forceinline__ void toc(int which)
  {
#ifdef PQP_STATS
    if (threadIdx.x == 0) {
      long long t = clock64();
      L.stat()[which] += t - L.stat()[ST_COUNT];
      L.stat()[ST_COUNT] = t;
    }
#endif
  }
  // nested sub-phase timers (no shared mark: -t at the start, +t at the end)
  __device__ __forceinline__ void sub_tic(int which)
  {
#ifdef PQP_STATS
    if (threadIdx.x == 0)
      L.stat()[which] -= clock64();
#endif
  }
  __device__ __forceinline__ void sub_toc(int which)
  {
#ifdef PQP_STATS
    if (threadIdx.x == 0)
      L.stat()[which] += clock64();
#endif
  }
  __device__ __forceinline__ void count(int which, long long v = 1)
  {
#ifdef PQP_STATS
    if (threadIdx.x == 0)
      L.stat()[which] += v;
#endif
  }
  // settings.verbose: one record per line the reference prints (Batch::trace)
  __device__ __forceinline__ void trace_line(double kind, double idx, double a, double b, double c, double d4, double e)
  {
    if (batch.trace == nullptr || threadIdx.x != 0)
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
    r[1] = idx;
    r[2] = a;
    r[3] = b;
    r[4] = c;
    r[5] = d4;
    r[6] = e;
    t[0] = double(k + 1);
  }
  // compulsory HBM bytes of the engine (one pass over a matrix = its size; no reuse assumed)
  __device__ __forceinline__ void bytes(long long b) { count(ST_BYTES_ENGINE, b); }
  __device__ __forceinline__ int reps(int phase) const
  {
#ifdef PQP_STATS
    int c = (batch.rep_phase == phase && batch.rep_count > 1) ? batch.rep_count : 1;
    PQP_OPAQUE_SCALAR(c);
    return c;
#else
    (void)phase;
    return 1;
#endif
  }

  // ---- small helpers --------------------------------------------------------
  __device__ __forceinline__ void vzero(lptr v, int len)
  {
    for (int k = threadIdx.x; k < len; k += NT)
      v[k] = 0;
  }
  __device__ __forceinline__ void vcopy(lptr dst, clptr src, int len)
  {
    for (int k = threadIdx.x; k < len; k += NT)
      dst[k] = src[k];
  }
  __device__ __forceinline__ void vload(lptr dst, cgptr src, int len)
  {
    for (int k = threadIdx.x; k < len; k += NT)
      dst[k] = src[k];
  }
  __device__ __forceinline__ void vstore(gptr dst, clptr src, int len)
  {
    for (int k = threadIdx.x; k < len; k += NT)
      dst[k] = src[k];
  }
  __device__ __forceinline__ bool flag_up(int i) const { return (L.aflags()[i] & 1) != 0; }
  __device__ __forceinline__ bool flag_low(int i) const { return (L.aflags()[i] & 2) != 0; }
  __device__ __forceinline__ int cid_of_slot(int a) const
  {
    // branch-free (the LDS read is unconditional on a clamped index) so that callers can batch
    // the global loads that depend on it
    const int k = a - d.n_eq;
    const int v = L.act()[k < 0 ? 0 : k];
    return (k < 0) ? a : d.n_eq + v;
  }
  // out = H_s v for the dense H_s
  // elements of H_s one hess_mv pass reads (engine byte counter)
  __device__ __forceinline__ long hess_pass_elems() const
  {
    return (long)d.n * d.n;
  }
  __device__ __forceinline__ void hess_mv(clptr v, lptr out)
  {
    // column sums of the symmetric H_s = H_s v, with 16-byte loads
    gemv_dual<NT, true, false, false>(P.Hs(), d.n, d.n, d.n, v, v, out, out, L.part());
  }
  // plain mat-vec through the shared routine
  __device__ __forceinline__ void mv(cgptr M, int ld, int K, int J, clptr v, lptr out)
  {
    gemv<NT>(M, ld, K, J, v, out, L.part(), nullptr, 0, nullptr, 0);
  }

  // ---- primal block ---------------------------------------------------------
  // H_s + rho I = L D L^T and the explicit L^{-1} (reference helpers.hpp:252-264 for
  // the assembled block, ldlt.hpp:718-744 for the factorisation it feeds)
  template<bool STAGED = true>
  __device__ __forceinline__ void factor_primal_block()
  {
    const int n = d.n;
    const double rho = info.rho;
    if (pm()) {
      // the model-only part of P_J: A_s^T A_s (WU buffer), once per solve
      if (d.n_eq > 0)
        syrk_mfma<NT>(P.As(), n, d.n_eq, nullptr, n, 1.0, nullptr, P.WU(), n);
      bytes(((long)d.n_eq * n + (long)n * n) * 8);
      return;
    }
    if (hess() == PQP_HESSIAN_DENSE) {
      gptr F = P.F();
      cgptr Hs = P.Hs();
      bool done = false;
      if constexpr (NT == 256) {
        if (n <= 16 * SCHUR_MB) {
          // register-resident factorisation straight from H_s (upper triangle read), then the
          // explicit inverse on the matrix cores
          auto load = [&](int i, int j) -> double { return Hs[(long)j * n + i] + ((i == j) ? rho : 0.0); };
          ldlt_factor_reg<NT, SCHUR_MB>(load, F, n, n, L.dF(), L.top());
          toc(ST_CYC_F_PANEL); // (sub-phases of ST_CYC_FACTOR_H, which the caller bills in full)
          diag_block_inverses_mfma<NT>(F, n, n);
          tri_inverse_mfma<NT, SCHUR_MB>(F, n, n, P.WL(), P.WU());
          toc(ST_CYC_F_TINV);
          done = true;
        }
      }
      if (PQP_UNLIKELY(!done)) {
        for (int o = threadIdx.x; o < n * n; o += NT) {
          int rr = o / n, k = o - rr * n;
          F[o] = Hs[o] + ((rr == k) ? rho : 0.0);
        }
        __syncthreads();
        ldlt_factor_mfma<NT, true>(F, n, n, L.dF(), L.top());
        toc(ST_CYC_F_PANEL);
        if (n <= 16 * SCHUR_MB)
          tri_inverse_mfma<NT, SCHUR_MB>(F, n, n, P.WL(), P.WU());
        else
          tri_inverse_mfma_rows<NT>(F, n, n, P.WL(), P.WU());
        toc(ST_CYC_F_TINV);
      }
    } else {
      // diagonal / zero Hessian: L = I
      cgptr Hs = P.Hs();
      cgptr hd = P.F();
      for (int k = threadIdx.x; k < n; k += NT)
        L.dF()[k] = ((hess() == PQP_HESSIAN_DIAGONAL) ? (dm() ? hd[k] : Hs[(long)k * n + k]) : 0.0) + rho;
      __syncthreads();
    }
    vstore(P.dF(), L.dF(), n);
    if (hess() == PQP_HESSIAN_DENSE) {
      bytes((long)n * n * 8 * 3); // H_s read (upper triangle) + W and W^T written (lower triangle each), ~1.5 n^2 + margin for F
      count(ST_FLOPS_FACT, (long long)n * n * n / 3);
    }
    build_ZG<STAGED>();
  }

  // Z = L^{-1} B^T for EVERY constraint row (B = [A_s; C_s; diag(i_scaled)]) in both
  // orientations, and the full Gram matrix G = Z^T D^{-1} Z, right after the primal block is
  // factorised.  Two small GEMMs on the FP64 matrix cores (16x16x4 tiles, one tile per
  // wavefront at a time, operands loaded straight from L2/HBM in the MFMA lane layout):
  //   Zc[k][c] = sum_j W[k][j] Bt[j][c]   and, from the SAME two operand registers with the
  //   roles swapped, Zr[c][k] -- so both orientations are written with coalesced stores;
  //   G[c][d]  = sum_k Zc[k][c] (1/D_k) Zc[k][d], lower tiles + their mirrors.
  // ~5 MFLOP per QP at C2, done once per factorisation as dense tiles (an earlier version validated
  // rows lazily, one latency-bound pass over W and Z per newly active batch of constraints).
  // STAGED: the two GEMMs run LDS-tiled through L.stage() -- every per-QP vector behind dF / t1 is scratch then (the
  // solve's prologue, which loads its vectors afterwards); false: operands straight from L2 / HBM per wavefront (the
  // backward kernel, whose iterate is already in LDS).  Same results bit for bit.
  // (LDS-tiled in the 512- / 1024-thread kernels: C4 -3 % time, -100 MB of HBM traffic per QP; in the 256-thread
  // kernels, whose operands mostly sit in L2 and whose four wavefronts gain little from sharing, the per-wavefront form
  // stays: the tiled one is 5 % slower at C2 and 8 % at C1 -- two barriers per slab -- profiles/r04_ab_zg_lds_tiled.txt)
  template<bool STAGED = true>
  __device__ __forceinline__ void build_ZG()
  {
    const int n = d.n, ne = d.n_eq, ni = d.n_in, nd = d.nd, nb = ne + ni;
    constexpr int NWV = NT / WAVE;
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    const int lr = lane & 15, lk = lane >> 4;
    gptr Zc = P.Zc(), Zr = P.Zr();
    if (dm()) {
      // zd[c] = the one entry of row c of Z = B^T (L = I), gd[c] = zd[c]^2 / D_col(c)
      cgptr cd = P.CTs();
      cgptr isg = P.is(); // (i_scaled from HBM: the LDS copy is loaded after this prologue)
      gptr gd = P.G();
      for (int c = threadIdx.x; c < nd; c += NT) {
        const double z = (c < ni) ? cd[c] : isg[c - ni];
        Zr[c] = z;
        gd[c] = z * z / L.dF()[dcol(c)];
      }
      bytes((long)nd * 8 * 3);
      __syncthreads();
      return;
    }
    for (int k = threadIdx.x; k < n; k += NT)
      L.t1()[k] = 1.0 / L.dF()[k];
    if (hess() == PQP_HESSIAN_DENSE) {
      cgptr WU = P.WU(), ATs = P.ATs(), CTs = P.CTs();
      // work unit = one 16-row block of Z times TWO adjacent 16-column blocks: the W operand is
      // loaded once for both, and every batch keeps 3 * ZG_DEPTH loads in flight per lane
      const int KT = (n + 15) / 16, CT = (nb + 15) / 16, CP = (CT + 1) / 2;
      if constexpr (STAGED && ZG_LDS_TILED(NT)) {
        (void)KT;
        (void)CP;
        __syncthreads(); // (t1 complete; nothing else of the staging area is live)
        using T = ZgTile<NT>;
        lptr stage = L.stage();
        const int wr = w / T::WC, wc = w - wr * T::WC;
        lptr sc = stage + w * (16 * 17);
        auto loadA = [&](int j, int k) -> double { return (k < n) ? WU[(long)j * n + k] : 0.0; }; // W[k][j]
        auto loadB = [&](int j, int c) -> double {
          return (c < ne) ? ATs[(long)j * ne + c] : ((c < nb) ? CTs[(long)j * ni + (c - ne)] : 0.0);
        };
        for (int R0 = 0; R0 < n; R0 += T::TM)
          for (int C0 = 0; C0 < nb; C0 += T::TN) {
            const int jend = (R0 + T::TM < n) ? (R0 + T::TM) : n; // W[k][j] = 0 for j > k
            pqp_d4 acc[2][2];
            zg_block<NT>(loadA, loadB, R0, C0, jend, stage, acc);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
              for (int g = 0; g < 2; ++g) {
                const int k0 = R0 + wr * 32 + h * 16, c0 = C0 + wc * 32 + g * 16;
                if (k0 >= n || c0 >= nb) // (wave-uniform)
                  continue;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                  const int kr = k0 + lk + 4 * rr;
                  if (kr < n && c0 + lr < nb)
                    Zc[(long)kr * nd + c0 + lr] = acc[h][g][rr];
                }
                const pqp_d4 t = zg_transpose(acc[h][g], sc);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                  const int cr = c0 + lk + 4 * rr;
                  if (cr < nb && k0 + lr < n)
                    Zr[(long)cr * n + k0 + lr] = t[rr];
                }
              }
          }
      } else if constexpr (NT >= 512 && PQP_ZG_BLOCK2) {
        // 2 x 2: two 16-row blocks of Z (k) times two 16-column blocks (c)
        const int KP = (KT + 1) / 2;
        for (int t = w; t < KP * CP; t += NWV) {
          const int kp = t / CP, cp = t - kp * CP;
          int k0[2], k[2], kc[2];
          bool k_ok[2];
          int c0[2], c[2];
          bool c_ok[2];
          cgptr bbase[2];
          long bld[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            k0[h] = (2 * kp + h) * 16;
            k[h] = k0[h] + lr;
            k_ok[h] = k[h] < n;
            kc[h] = k_ok[h] ? k[h] : n - 1;
            c0[h] = (2 * cp + h) * 16;
            c[h] = c0[h] + lr;
            c_ok[h] = c[h] < nb;
            const bool is_eq = c[h] < ne;
            bbase[h] = is_eq ? (ATs + c[h]) : (CTs + (c_ok[h] ? c[h] - ne : 0));
            bld[h] = is_eq ? ne : ni;
          }
          pqp_d4 acc1[2][2], acc2[2][2];
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              acc1[h][g] = pqp_d4{ 0.0, 0.0, 0.0, 0.0 };
              acc2[h][g] = pqp_d4{ 0.0, 0.0, 0.0, 0.0 };
            }
          // W[k][j] = 0 for j > k (the zeros of WU's lower triangle are stored): one range for both row blocks
          const int jend = (k0[1] + 16 < n) ? (k0[1] + 16) : n;
          for (int j0 = 0; j0 < jend; j0 += 4 * ZG2_DEPTH) {
            double a[2][ZG2_DEPTH], b[2][ZG2_DEPTH];
#pragma unroll
            for (int u = 0; u < ZG2_DEPTH; ++u) {
              const int j = j0 + 4 * u + lk;
              const int jc = (j < n) ? j : n - 1;
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                a[h][u] = WU[(long)jc * n + kc[h]];
                b[h][u] = bbase[h][(long)jc * bld[h]];
              }
            }
#pragma unroll
            for (int u = 0; u < ZG2_DEPTH; ++u) {
              const int j = j0 + 4 * u + lk;
              double av[2], bv[2];
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                av[h] = (j < n && k_ok[h]) ? a[h][u] : 0.0;
                bv[h] = (j < n && c_ok[h]) ? b[h][u] : 0.0;
              }
#pragma unroll
              for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                  acc1[h][g] = mfma_f64_16x16x4(av[h], bv[g], acc1[h][g]);
                  acc2[h][g] = mfma_f64_16x16x4(bv[g], av[h], acc2[h][g]);
                }
            }
          }
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) {
                const int kr = k0[h] + lk + 4 * rr;
                if (kr < n && c_ok[g])
                  Zc[(long)kr * nd + c[g]] = acc1[h][g][rr];
                const int cr = c0[g] + lk + 4 * rr;
                if (cr < nb && k_ok[h])
                  Zr[(long)cr * n + k[h]] = acc2[h][g][rr];
              }
        }
      } else
      for (int t = w; t < KT * CP; t += NWV) {
        const int kt = t / CP, cp = t - kt * CP;
        const int k0 = kt * 16;
        const int k = k0 + lr;
        const bool k_ok = k < n;
        const int kc = k_ok ? k : n - 1;
        int c0[2], c[2];
        bool c_ok[2];
        cgptr bbase[2];
        long bld[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          c0[h] = (2 * cp + h) * 16;
          c[h] = c0[h] + lr;
          c_ok[h] = c[h] < nb;
          const bool is_eq = c[h] < ne;
          bbase[h] = is_eq ? (ATs + c[h]) : (CTs + (c_ok[h] ? c[h] - ne : 0));
          bld[h] = is_eq ? ne : ni;
        }
        pqp_d4 acc1[2], acc2[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          acc1[h] = pqp_d4{ 0.0, 0.0, 0.0, 0.0 };
          acc2[h] = pqp_d4{ 0.0, 0.0, 0.0, 0.0 };
        }
        const int jend = (k0 + 16 < n) ? (k0 + 16) : n; // W[k][j] = 0 for j > k
        for (int j0 = 0; j0 < jend; j0 += 4 * ZG_DEPTH) {
          double a[ZG_DEPTH], b0[ZG_DEPTH], b1[ZG_DEPTH];
#pragma unroll
          for (int u = 0; u < ZG_DEPTH; ++u) {
            const int j = j0 + 4 * u + lk;
            const int jc = (j < n) ? j : n - 1;
            a[u] = WU[(long)jc * n + kc];
            b0[u] = bbase[0][(long)jc * bld[0]];
            b1[u] = bbase[1][(long)jc * bld[1]];
          }
#pragma unroll
          for (int u = 0; u < ZG_DEPTH; ++u) {
            const int j = j0 + 4 * u + lk;
            const double av = (j < n && k_ok) ? a[u] : 0.0;
            const double bv0 = (j < n && c_ok[0]) ? b0[u] : 0.0;
            const double bv1 = (j < n && c_ok[1]) ? b1[u] : 0.0;
            acc1[0] = mfma_f64_16x16x4(av, bv0, acc1[0]);
            acc2[0] = mfma_f64_16x16x4(bv0, av, acc2[0]);
            acc1[1] = mfma_f64_16x16x4(av, bv1, acc1[1]);
            acc2[1] = mfma_f64_16x16x4(bv1, av, acc2[1]);
          }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int kr = k0 + lk + 4 * rr;
            if (kr < n && c_ok[h])
              Zc[(long)kr * nd + c[h]] = acc1[h][rr];
            const int cr = c0[h] + lk + 4 * rr;
            if (cr < nb && k_ok)
              Zr[(long)cr * n + k] = acc2[h][rr];
          }
      }
      if (has_box()) {
        cgptr WL = P.WL();
        cgptr isg = P.is(); // (i_scaled from HBM: the LDS copy is loaded after this prologue)
        for (int o = threadIdx.x; o < n * n; o += NT) {
          const int a = o / n, bcol = o - a * n;
          Zc[(long)a * nd + nb + bcol] = WL[o] * isg[bcol]; // Z[k=a][box bcol] = W[a][bcol] i_bcol
          Zr[(long)(nb + a) * n + bcol] = WU[o] * isg[a];   // Zr[box a][k=bcol] = W[bcol][a] i_a
        }
      }
    } else {
      // diagonal / zero Hessian: L = I, Z = B^T
      cgptr As = P.As(), Cs = P.Cs(), ATs = P.ATs(), CTs = P.CTs();
      for (int o = threadIdx.x; o < nd * n; o += NT) {
        const int c = o / n, k = o - c * n;
        double v;
        if (c < ne)
          v = As[(long)c * n + k];
        else if (c < nb)
          v = Cs[(long)(c - ne) * n + k];
        else
          v = (k == c - nb) ? P.is()[k] : 0.0;
        Zr[o] = v;
      }
      for (int o = threadIdx.x; o < n * nd; o += NT) {
        const int k = o / nd, c = o - k * nd;
        double v;
        if (c < ne)
          v = ATs[(long)k * ne + c];
        else if (c < nb)
          v = CTs[(long)k * ni + (c - ne)];
        else
          v = (k == c - nb) ? P.is()[k] : 0.0;
        Zc[o] = v;
      }
    }
    __syncthreads();
    {
      gptr G = P.G();
      cgptr Zcc = P.Zc();
      // work unit = block row ct of G times TWO adjacent block columns dt, dt+1 <= ct (lower
      // tiles; each off-diagonal tile also yields its mirror from the swapped operands)
      const int DT = (nd + 15) / 16;
      if constexpr (STAGED && ZG_LDS_TILED(NT)) {
        (void)DT;
        using T = ZgTile<NT>;
        lptr stage = L.stage();
        const int wr = w / T::WC, wc = w - wr * T::WC;
        lptr sc = stage + w * (16 * 17);
        auto loadA = [&](int k, int c) -> double { return (c < nd) ? Zcc[(long)k * nd + c] * L.t1()[k] : 0.0; };
        auto loadB = [&](int k, int dd) -> double { return (dd < nd) ? Zcc[(long)k * nd + dd] : 0.0; };
        for (int R0 = 0; R0 < nd; R0 += T::TM)
          for (int C0 = 0; C0 < nd && C0 <= R0 + T::TM - 1; C0 += T::TN) { // blocks that reach the lower triangle
            pqp_d4 acc[2][2];
            zg_block<NT>(loadA, loadB, R0, C0, n, stage, acc);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
              for (int g = 0; g < 2; ++g) {
                const int c0 = R0 + wr * 32 + h * 16, d0 = C0 + wc * 32 + g * 16; // tile rows c, columns d
                if (d0 > c0 || c0 >= nd) // above G's diagonal (its mirror below writes it) or past the last row
                  continue;
                if (d0 != c0) {
#pragma unroll
                  for (int rr = 0; rr < 4; ++rr) {
                    const int cr = c0 + lk + 4 * rr;
                    if (cr < nd && d0 + lr < nd)
                      G[(long)cr * nd + d0 + lr] = acc[h][g][rr];
                  }
                  const pqp_d4 t = zg_transpose(acc[h][g], sc);
#pragma unroll
                  for (int rr = 0; rr < 4; ++rr) {
                    const int dr = d0 + lk + 4 * rr;
                    if (dr < nd && c0 + lr < nd)
                      G[(long)dr * nd + c0 + lr] = t[rr];
                  }
                } else {
                  // diagonal tile: keep G exactly symmetric (lower part + its mirror)
#pragma unroll
                  for (int rr = 0; rr < 4; ++rr) {
                    const int cr = c0 + lk + 4 * rr, dcol = d0 + lr;
                    if (cr < nd && dcol < nd && cr >= dcol) {
                      G[(long)cr * nd + dcol] = acc[h][g][rr];
                      G[(long)dcol * nd + cr] = acc[h][g][rr];
                    }
                  }
                }
              }
          }
      } else if constexpr (NT >= 512 && PQP_ZG_BLOCK2) {
        // 2 x 2: block rows (2 sc, 2 sc + 1) of G times block columns (2 sd, 2 sd + 1), sd <= sc; on the diagonal
        // of this coarser grid the tile above G's diagonal is the mirror of the one below and is skipped
        const int ST = (DT + 1) / 2;
        const int units2 = ST * (ST + 1) / 2;
        for (int t = w; t < units2; t += NWV) {
          int sc = 0, sd = t;
          while (sd >= sc + 1) {
            sd -= sc + 1;
            ++sc;
          }
          int c0[2], c[2], cc[2], d0[2], dcol[2], dcc[2];
          bool c_ok[2], d_ok[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            c0[h] = (2 * sc + h) * 16;
            c[h] = c0[h] + lr;
            c_ok[h] = c[h] < nd;
            cc[h] = c_ok[h] ? c[h] : nd - 1;
            d0[h] = (2 * sd + h) * 16;
            dcol[h] = d0[h] + lr;
            d_ok[h] = dcol[h] < nd;
            dcc[h] = d_ok[h] ? dcol[h] : nd - 1;
          }
          const bool diag2 = (sc == sd);
          pqp_d4 acc1[2][2], acc2[2][2];
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              acc1[h][g] = pqp_d4{ 0.0, 0.0, 0.0, 0.0 };
              acc2[h][g] = pqp_d4{ 0.0, 0.0, 0.0, 0.0 };
            }
          for (int k0 = 0; k0 < n; k0 += 4 * ZG2_DEPTH) {
            double za[2][ZG2_DEPTH], zb[2][ZG2_DEPTH], sv[ZG2_DEPTH];
#pragma unroll
            for (int u = 0; u < ZG2_DEPTH; ++u) {
              const int k = k0 + 4 * u + lk;
              const int kc = (k < n) ? k : n - 1;
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                za[h][u] = Zcc[(long)kc * nd + cc[h]];
                zb[h][u] = Zcc[(long)kc * nd + dcc[h]];
              }
              sv[u] = L.t1()[kc];
            }
#pragma unroll
            for (int u = 0; u < ZG2_DEPTH; ++u) {
              const int k = k0 + 4 * u + lk;
              double av[2], bv[2];
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                av[h] = (k < n && c_ok[h]) ? za[h][u] * sv[u] : 0.0;
                bv[h] = (k < n && d_ok[h]) ? zb[h][u] : 0.0;
              }
              // (1, 0) and, off the coarse diagonal, (0, 0) (0, 1) (1, 1): tile and mirror
              acc1[1][0] = mfma_f64_16x16x4(av[1], bv[0], acc1[1][0]);
              acc2[1][0] = mfma_f64_16x16x4(bv[0], av[1], acc2[1][0]);
              acc1[0][0] = mfma_f64_16x16x4(av[0], bv[0], acc1[0][0]);
              acc1[1][1] = mfma_f64_16x16x4(av[1], bv[1], acc1[1][1]);
              if (!diag2) {
                acc2[0][0] = mfma_f64_16x16x4(bv[0], av[0], acc2[0][0]);
                acc2[1][1] = mfma_f64_16x16x4(bv[1], av[1], acc2[1][1]);
                acc1[0][1] = mfma_f64_16x16x4(av[0], bv[1], acc1[0][1]);
                acc2[0][1] = mfma_f64_16x16x4(bv[1], av[0], acc2[0][1]);
              }
            }
          }
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              const int ct = 2 * sc + h, dt = 2 * sd + g;
              if (dt > ct || ct >= DT || dt >= DT)
                continue; // above G's diagonal (coarse diagonal only) or past the last block
              if (dt != ct) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                  const int cr = c0[h] + lk + 4 * rr;
                  if (cr < nd && d_ok[g])
                    G[(long)cr * nd + dcol[g]] = acc1[h][g][rr];
                  const int dr = d0[g] + lk + 4 * rr;
                  if (dr < nd && c_ok[h])
                    G[(long)dr * nd + c[h]] = acc2[h][g][rr];
                }
              } else {
                // diagonal tile: keep G exactly symmetric (lower part + its mirror)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                  const int cr = c0[h] + lk + 4 * rr;
                  if (cr < nd && d_ok[g] && cr >= dcol[g]) {
                    G[(long)cr * nd + dcol[g]] = acc1[h][g][rr];
                    G[(long)dcol[g] * nd + cr] = acc1[h][g][rr];
                  }
                }
              }
            }
        }
      } else {
      int units = 0;
      for (int ct = 0; ct < DT; ++ct)
        units += ct / 2 + 1;
      for (int t = w; t < units; t += NWV) {
        int ct = 0, rem = t;
        while (rem >= ct / 2 + 1) {
          rem -= ct / 2 + 1;
          ++ct;
        }
        const int c0 = ct * 16;
        const int c = c0 + lr;
        const bool c_ok = c < nd;
        const int cc = c_ok ? c : nd - 1;
        int dts[2], d0[2], dcol[2], dcc[2];
        bool d_ok[2], live[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          dts[h] = 2 * rem + h;
          live[h] = dts[h] <= ct; // the second tile of the last pair may not exist
          d0[h] = dts[h] * 16;
          dcol[h] = d0[h] + lr;
          d_ok[h] = live[h] && dcol[h] < nd;
          dcc[h] = d_ok[h] ? dcol[h] : nd - 1;
        }
        pqp_d4 acc1[2], acc2[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          acc1[h] = pqp_d4{ 0.0, 0.0, 0.0, 0.0 };
          acc2[h] = pqp_d4{ 0.0, 0.0, 0.0, 0.0 };
        }
        for (int k0 = 0; k0 < n; k0 += 4 * ZG_DEPTH) {
          double za[ZG_DEPTH], zb0[ZG_DEPTH], zb1[ZG_DEPTH], sv[ZG_DEPTH];
#pragma unroll
          for (int u = 0; u < ZG_DEPTH; ++u) {
            const int k = k0 + 4 * u + lk;
            const int kc = (k < n) ? k : n - 1;
            za[u] = Zcc[(long)kc * nd + cc];
            zb0[u] = Zcc[(long)kc * nd + dcc[0]];
            zb1[u] = Zcc[(long)kc * nd + dcc[1]];
            sv[u] = L.t1()[kc];
          }
#pragma unroll
          for (int u = 0; u < ZG_DEPTH; ++u) {
            const int k = k0 + 4 * u + lk;
            const double av = (k < n && c_ok) ? za[u] * sv[u] : 0.0;
            const double bv0 = (k < n && d_ok[0]) ? zb0[u] : 0.0;
            const double bv1 = (k < n && d_ok[1]) ? zb1[u] : 0.0;
            acc1[0] = mfma_f64_16x16x4(av, bv0, acc1[0]);
            acc2[0] = mfma_f64_16x16x4(bv0, av, acc2[0]);
            acc1[1] = mfma_f64_16x16x4(av, bv1, acc1[1]);
            acc2[1] = mfma_f64_16x16x4(bv1, av, acc2[1]);
          }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (!live[h])
            continue;
          if (dts[h] != ct) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const int cr = c0 + lk + 4 * rr;
              if (cr < nd && d_ok[h])
                G[(long)cr * nd + dcol[h]] = acc1[h][rr];
              const int dr = d0[h] + lk + 4 * rr;
              if (dr < nd && c_ok)
                G[(long)dr * nd + c] = acc2[h][rr];
            }
          } else {
            // diagonal tile: keep G exactly symmetric (lower part + its mirror)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const int cr = c0 + lk + 4 * rr;
              if (cr < nd && d_ok[h] && cr >= dcol[h]) {
                G[(long)cr * nd + dcol[h]] = acc1[h][rr];
                G[(long)dcol[h] * nd + cr] = acc1[h][rr];
              }
            }
          }
        }
      }
      }
    }
    count(ST_N_NEW_ROWS, nd);
    // Z build: W^T and B^T read once per tile row pair, Z written in both orientations; Gram: Zc read, G written
    bytes(((hess() == PQP_HESSIAN_DENSE ? (long)n * n / 2 + (long)nb * n : (long)nd * n) + 2L * nd * n + (long)nd * n +
           (long)nd * nd) * 8);
    __syncthreads();
  }

  // out = L^{-1} v  /  out = L^{-T} v   (LDS vectors, may alias)
  // (out2 / div: optional epilogue out2 = out / div written in the same pass)
  __device__ __forceinline__ void apply_Linv(clptr v, lptr out, bool transposed, lptr out2 = nullptr, clptr div = nullptr)
  {
    if (hess() == PQP_HESSIAN_DENSE) {
      // W = L^{-1} is lower triangular: WU[k][j] = W[j][k] vanishes for k > j, WL[k][j] for k < j
      gemv<NT>(transposed ? (cgptr)P.WL() : (cgptr)P.WU(), d.n, d.n, d.n, v, out, L.part(), nullptr, 0, nullptr,
               0, transposed ? -1 : +1, out2, div);
    } else {
      for (int k = threadIdx.x; k < d.n; k += NT) {
        const double t = v[k];
        out[k] = t;
        if (out2)
          out2[k] = t / div[k];
      }
      __syncthreads();
    }
  }

  // ---- dual Schur block -------------------------------------------------------
  // S_J = M_J + G_JJ = L_S D_S L_S^T in slot order (equalities, then the inequality slots) is
  // kept in INVERSE-FACTOR form: W_S = L_S^{-1} (row-major in HBM) and D_S (LDS).  Then
  //   * a solve is  W_S^T D_S^{-1} W_S v : two mat-vecs, no substitution chain (schur_apply);
  //   * a constraint entering the active set appends one row:  u = S^{-1} g,  W[new] = [-u^T 1],
  //     d_new = s_cc - g.u   (the reference's insert_block_at, linalg/dense/modify.hpp:129-264,
  //     whose new row of L is l = D^{-1} W g: the same numbers, held as -l^T W);
  //   * a constraint leaving it is delete_at (modify.hpp:80-127): the trailing factor takes the
  //     rank-1 update  L33' D3' L33'^T = L33 D3 L33^T + d_p l_p l_p^T  (update.hpp:219-287).  In
  //     inverse-factor form its `p` vector is read off column p of W (p = L33^{-1} l_p = -W[.,p]),
  //     the scalars alpha_j, d'_j, beta_j of the recurrence are a prefix sum
  //     (1/alpha_{j+1} = 1/alpha_j + p_j^2 / d_j), and W' = Ltilde^{-1} W with
  //     Ltilde = I + tril(p beta^T, -1) is one two-term recurrence down every COLUMN of W
  //     independently: one thread per column, no cross-thread dependency (schur_delete).  The slot
  //     is left behind as a hole (identity row / column, d = 1) until the next full factorisation.
  //   * a mu update changes every diagonal entry of S_J (the reference's
  //     diagonal_update_clobber_indices, ldlt.hpp:516-570, a rank-(n_eq + n_c) update): r rank-1
  //     sweeps would be r^2 dependent steps, the fused re-factorisation below is r.

  // slot a (dual index) holds a live row?  (equalities always; an inequality slot is a hole once its
  // constraint has left: slot_of no longer points back at it)
  __device__ __forceinline__ bool slot_live(int a) const
  {
    const int k = a - d.n_eq;
    if (k < 0)
      return true;
    return L.slot_of()[L.act()[k]] == k;
  }
  __device__ __forceinline__ void zero_holes(lptr v)
  {
    if (n_slots > n_c) {
      for (int a = d.n_eq + threadIdx.x; a < r; a += NT)
        if (!slot_live(a))
          v[a] = 0.0;
      __syncthreads();
    }
  }

  // (emulator-only self-check of the factor identity W S W^T = D, tests/emu with -DPQP_TRACE)
  __device__ __forceinline__ void debug_check_factor(const char* tag)
  {
#ifdef PQP_TRACE
    __syncthreads();
    if (threadIdx.x == 0) {
      const int nd = d.nd, rr = r, ne = d.n_eq;
      cgptr G = P.G();
      cgptr Wd = P.WS();
      double worst = 0, wmax = 0;
      static thread_local double Sm[512 * 512], T[512 * 512];
      for (int i = 0; i < rr; ++i)
        for (int j = 0; j < rr; ++j) {
          const int ci = cid_of_slot(i), cj = cid_of_slot(j);
          const bool live = slot_live(i) && slot_live(j);
          double v = live ? G[(long)ci * nd + cj] : 0.0;
          if (i == j)
            v = live ? v + ((i < ne) ? double(info.mu_eq) : double(info.mu_in)) : 1.0;
          Sm[i * rr + j] = v;
        }
      for (int i = 0; i < rr; ++i)
        for (int j = 0; j < rr; ++j) {
          double acc = 0;
          for (int k = 0; k <= i; ++k)
            acc += Wd[(long)i * nd + k] * Sm[k * rr + j];
          T[i * rr + j] = acc;
          wmax = fmax(wmax, fabs(Wd[(long)i * nd + j]));
          if (j > i && Wd[(long)i * nd + j] != 0.0)
            printf("  !! nonzero above the diagonal of W at (%d, %d)\n", i, j);
        }
      for (int i = 0; i < rr; ++i)
        for (int j = 0; j < rr; ++j) {
          double acc = 0;
          for (int k = 0; k <= j; ++k)
            acc += T[i * rr + k] * Wd[(long)j * nd + k];
          worst = fmax(worst, fabs(acc - ((i == j) ? L.dS()[i] : 0.0)));
        }
      printf("%s q=%ld r=%d n_c=%d slots=%d  max|W S W^T - D| = %.3e  max|W| = %.3e\n", tag, q, rr, n_c, n_slots, worst, wmax);
    }
    __syncthreads();
#else
    (void)tag;
#endif
  }

  // full factorisation of the current slots (holes are kept as identity rows, so that vectors in
  // slot order stay valid); leaves W_S in HBM, D_S in LDS
  __device__ __forceinline__ void factor_schur()
  {
    const int nd = d.nd;
    const int rr = r;
    const int ne = d.n_eq;
    cgptr G = P.G();
    const double mu_eq = info.mu_eq, mu_in = info.mu_in;
    if (pm()) {
      factor_pm();
      return;
    }
    if (dm()) {
      // diagonal Schur block: D_S = mu_in + gd over the active constraints, W_S = I (never stored)
      cgptr gd = P.G();
      for (int a = threadIdx.x; a < rr; a += NT)
        L.dS()[a] = mu_in + gd[L.act()[a]];
      bytes((long)rr * 8);
      __syncthreads();
      schur_dirty = false;
      schur_incremental = false;
      count(ST_N_SCHUR_FACT);
      return;
    }
    bool done = false;
#ifndef PQP_SCHUR_REG_ROWS
#define PQP_SCHUR_REG_ROWS (16 * SCHUR_MB)
#endif
    if constexpr (NT == 256 && PQP_SCHUR_REG_ROWS > 0) {
      if (rr > 0 && rr <= PQP_SCHUR_REG_ROWS) {
        // register-resident path: gather + Gauss-Jordan factorisation + write-back of W with no
        // intermediate HBM traffic.  G is symmetric; element (i, j) is read as G[cid_j][cid_i] so
        // that the 16 lanes of a row group sweep ascending constraint ids.
        // slot -> constraint id (-1: hole), resolved once into LDS scratch
        liptr sid = L.iscr();
        for (int a = threadIdx.x; a < rr; a += NT)
          sid[a] = slot_live(a) ? cid_of_slot(a) : -1;
        __syncthreads();
        auto load = [&](int i, int j) -> double {
          const int ci = sid[i], cj = sid[j];
          const bool live = (ci | cj) >= 0;
          const double v = G[(long)(live ? cj : 0) * nd + (live ? ci : 0)];
          if (i == j)
            return live ? v + ((i < ne) ? mu_eq : mu_in) : 1.0;
          return live ? v : 0.0;
        };
        ldlt_inverse_reg<NT, SCHUR_MB>(load, P.WS(), nd, rr, L.dS(), L.top());
        bytes((long)rr * (rr + 1) * 8); // gather of the lower triangle + the lower triangle of W written
        toc(ST_CYC_S_GATHER);
        done = true;
      }
    }
    if (PQP_UNLIKELY(!done)) {
      liptr sid = L.iscr();
      for (int a = threadIdx.x; a < rr; a += NT)
        sid[a] = slot_live(a) ? cid_of_slot(a) : -1;
      __syncthreads();
      // (three stages, billed apart in the stats build: gather / factorisation / inverse)
      schur_gather_blocked<NT>(G, P.LS(), nd, rr, ne, mu_eq, mu_in, sid);
      toc(ST_CYC_F_LOAD);
      ldlt_factor_mfma<NT, true>(P.LS(), nd, rr, L.dS(), L.top());
      toc(ST_CYC_F_UPDATE);
      tri_inverse_mfma_rows<NT, false>(P.LS(), nd, rr, P.WS(), P.WS());
      bytes((long)rr * rr * 8 * 4);
      toc(ST_CYC_F_WRITEBACK);
      count(ST_N_SCHUR_BLOCKED);
    }
    debug_check_factor("factor_schur");
    tic();
    schur_dirty = false;
    schur_incremental = false;
    count(ST_N_SCHUR_FACT);
    count(ST_FLOPS_FACT, (long long)rr * rr * rr / 3);
  }

  // PrimalLDLT: assemble P_J = H_s + rho I + A^T A / mu_eq + C_J^T C_J / mu_in (+ box rows) in the F
  // buffer and factorise it into (W = L^{-1} -> WL buffer, D -> L.dF)
  __device__ __forceinline__ void factor_pm()
  {
    const int n = d.n, ne = d.n_eq, ni = d.n_in;
    const double rho = info.rho, mu_eq_inv = 1.0 / double(info.mu_eq), mu_in_inv = 1.0 / double(info.mu_in);
    gptr Pm = P.F();
    {
      cgptr Hs = P.Hs();
      cgptr PA = P.WU();
      for (int o = threadIdx.x; o < n * n; o += NT) {
        const int i = o / n, k = o - i * n;
        double v = 0.0;
        if (hess() == PQP_HESSIAN_DENSE)
          v = Hs[o];
        else if (hess() == PQP_HESSIAN_DIAGONAL && i == k)
          v = Hs[o];
        if (i == k)
          v += rho;
        if (ne > 0)
          v += mu_eq_inv * PA[o];
        Pm[o] = v;
      }
    }
    __syncthreads();
    // active general inequalities: rows act[a] < n_in of C_s, gathered (the list is ascending, so
    // they come first); active box rows add i_k^2 / mu_in on the diagonal
    int n_gen = 0;
    {
      double cnt = 0;
      for (int a = threadIdx.x; a < n_c; a += NT)
        cnt += (L.act()[a] < ni) ? 1.0 : 0.0;
      n_gen = (int)R.sum(cnt);
    }
    if (n_gen > 0)
      syrk_mfma<NT>(P.Cs(), n, n_gen, L.act(), n, mu_in_inv, Pm, Pm, n);
    if (has_box()) {
      for (int a = n_gen + threadIdx.x; a < n_c; a += NT) {
        const int k = L.act()[a] - ni;
        const double ik = L.isc()[k];
        Pm[(long)k * n + k] += mu_in_inv * ik * ik;
      }
      __syncthreads();
    }
    bytes(((long)n * n * 4 + (long)n_gen * n) * 8);
    bool done = false;
    if constexpr (NT == 256) {
      if (n <= 16 * SCHUR_MB) {
        auto load = [&](int i, int j) -> double { return Pm[(long)i * n + j]; };
        ldlt_inverse_reg<NT, SCHUR_MB>(load, P.WL(), n, n, L.dF(), L.top());
        done = true;
      }
    }
    if (!done) {
      ldlt_factor_mfma<NT, true>(Pm, n, n, L.dF(), L.top());
      tri_inverse_mfma_rows<NT, false>(Pm, n, n, P.WL(), P.WL());
    }
    bytes((long)n * n * 8 * 2);
    schur_dirty = false;
    schur_incremental = false;
    count(ST_N_SCHUR_FACT);
    count(ST_FLOPS_FACT, (long long)n * n * n / 3);
  }

  // v <- S_J^{-1} v = W^T D^{-1} W v for an LDS vector over the r slots (zero at the holes, and
  // it stays zero there).  Scratch: L.t2(), L.part().
  __device__ __forceinline__ void schur_apply(lptr v)
  {
    const int rr = r;
    if (dm()) {
      for (int a = threadIdx.x; a < rr; a += NT)
        v[a] /= L.dS()[a];
      __syncthreads();
      return;
    }
    cgptr W = P.WS();
    // t = W v : row sums (16 lanes per row of the row-major factor)
    gemv_dual<NT, false, false, true, true>(W, d.nd, rr, rr, v, v, L.t2(), L.t2(), L.part(), nullptr, 0, EPI_ROW_DIV, L.dS());
    // v = W^T (t / D) : thread per column, rows below the diagonal only
    gemv<NT>(W, d.nd, rr, rr, L.t2(), v, L.part(), nullptr, 0, nullptr, 0, -1);
    bytes((long)rr * (rr + 1) * 8);
  }

  // delete the row / column of inequality slot `s` (uniform) from the factorisation
  __device__ __forceinline__ void schur_delete(int s)
  {
    const int nd = d.nd, rr = r;
    const int p = d.n_eq + s;
    gptr W = P.WS();
    lptr pv = L.rd(), beta = L.ed(), wp = L.sd();
    // p_i = -W[i][p] (i > p): the vector L33^{-1} l_p of the rank-1 update; row p of W for the
    // columns left of it
    for (int c = threadIdx.x; c < p; c += NT)
      wp[c] = W[(long)p * nd + c];
    // one thread per trailing row; the 1024-thread kernels walk blocks beyond 1024 rows in chunks of NT rows, the
    // prefix sum carried from chunk to chunk (one chunk, carry 0, in every other kernel: the code of rounds 2-3)
    constexpr bool CHUNKED = (NT == 1024) || PQP_CHUNK_ALL;
    double carry = 0.0;
    for (int base = p + 1; base == p + 1 || (CHUNKED && base < rr); base += NT) {
      const int i_own = base + threadIdx.x;
      double my_p = 0.0, my_d = 1.0, my_e = 0.0;
      if (i_own < rr) {
        my_p = -W[(long)i_own * nd + p];
        my_d = L.dS()[i_own];
        my_e = my_p * my_p / my_d;
      }
      // 1 / alpha_{i+1} = 1 / d_p + sum_{p < k <= i} p_k^2 / d_k   (update.hpp:243-262, as a scan)
      const double incl = carry + block_scan_inclusive<NT>(my_e, L.part());
      const double inv_a0 = 1.0 / L.dS()[p];
      if (i_own < rr) {
        const double c_i = inv_a0 + incl, c_im1 = c_i - my_e;
        pv[i_own] = my_p;
        beta[i_own] = my_p / (my_d * c_i);     // beta_i = alpha_{i+1} p_i / d_i
        L.dS()[i_own] = my_d * (c_i / c_im1);  // d'_i   = d_i alpha_i / alpha_{i+1}
      }
      if (CHUNKED && base + NT < rr) {
        if (threadIdx.x == NT - 1)
          L.part()[NT / WAVE] = incl;
        __syncthreads();
        carry = L.part()[NT / WAVE];
      }
      __syncthreads();
    }
    // W' = Ltilde^{-1} (W + p w_p^T on the columns left of p):  x_i = y_i - p_i s,  s += beta_i x_i
    // down each column; all lanes walk the same row (coalesced), entries above the diagonal read
    // as the zeros they are and are not written
    for (int c = threadIdx.x; c < rr; c += NT) {
      const double wpc = (c < p) ? wp[c] : 0.0;
      const bool skip = (c == p);
      double sacc = 0.0;
      gptr col = W + c;
      // DEL_U rows per trip: their loads are issued together (one memory round trip per trip; the
      // recurrence itself is a chain of two FMAs per row)
      constexpr int DEL_U = 16;
      for (int i = p + 1; i < rr; i += DEL_U) {
        double y[DEL_U];
#pragma unroll
        for (int u = 0; u < DEL_U; ++u)
          y[u] = col[(long)((i + u < rr) ? (i + u) : (rr - 1)) * nd];
#pragma unroll
        for (int u = 0; u < DEL_U; ++u)
          if (i + u < rr) { // uniform
            const double pi = pv[i + u];
            const double x = fma(-pi, sacc, fma(pi, wpc, y[u]));
            sacc = fma(beta[i + u], x, sacc);
            if (!skip && i + u >= c)
              col[(long)(i + u) * nd] = x;
          }
      }
    }
    __syncthreads();
    // the slot becomes a hole: identity row and column, unit pivot
    for (int c = threadIdx.x; c < p; c += NT)
      W[(long)p * nd + c] = 0.0;
    for (int i = p + 1 + threadIdx.x; i < rr; i += NT)
      W[(long)i * nd + p] = 0.0;
    if (threadIdx.x == 0)
      L.dS()[p] = 1.0;
    bytes((long)(rr - p) * rr * 16);
    count(ST_N_DELETE);
    __syncthreads();
  }

  // append inequality constraint `cid` (uniform) as the new last slot; returns false when the new
  // pivot is not positive (rounding on a near-singular block: the caller re-factorises)
  __device__ __forceinline__ bool schur_append(int cid)
  {
    const int nd = d.nd, ne = d.n_eq, rr = r;
    const long gid = ne + cid;
    cgptr G = P.G();
    gptr W = P.WS();
    lptr gv = L.rd(), tv = L.ed(), uv = L.sd();
    // g = S[new][slots] : row gid of the Gram cache, gathered in slot order (zero at the holes)
    for (int j = threadIdx.x; j < rr; j += NT) {
      const int cj = cid_of_slot(j);
      const double v = G[gid * nd + cj];
      gv[j] = slot_live(j) ? v : 0.0;
    }
    const double scc = G[gid * nd + gid] + info.mu_in;
    __syncthreads();
    double delta = scc;
    if (rr > 0) {
      // u = S^{-1} g ;  d_new = s_cc - g . u = s_cc - sum t_j^2 / d_j  with t = W g
      gemv_dual<NT, false, false, true, true>(W, nd, rr, rr, gv, gv, tv, tv, L.part());
      double acc = 0.0;
      for (int j = threadIdx.x; j < rr; j += NT) {
        const double t = tv[j], dj = L.dS()[j];
        acc = fma(t, t / dj, acc);
        tv[j] = t / dj;
      }
      delta = scc - R.sum(acc);
      gemv<NT>(W, nd, rr, rr, tv, uv, L.part(), nullptr, 0, nullptr, 0, -1);
      for (int j = threadIdx.x; j < rr; j += NT)
        W[(long)rr * nd + j] = -uv[j];
    }
    if (threadIdx.x == 0) {
      W[(long)rr * nd + rr] = 1.0;
      L.dS()[rr] = delta;
      L.act()[rr - ne] = cid;
      L.slot_of()[cid] = rr - ne;
    }
    n_slots += 1;
    r += 1;
    bytes((long)rr * (rr + 1) * 8 + (long)rr * 16);
    count(ST_N_APPEND);
    __syncthreads();
    return delta > 0.0;
  }

  // PrimalLDLT: a constraint entering (sign +1) or leaving (sign -1) the active set changes P_J by
  // +- c c^T / mu_in -- a rank-1 update of its factorisation, done on the inverse factor exactly like
  // the trailing update of schur_delete (reference update.hpp:219-287): p = L^{-1} c = W c (one
  // mat-vec; for a box row, a column of W), the recurrence scalars from a prefix sum
  // (1/alpha_{i+1} = 1/alpha_i + p_i^2/d_i, 1/alpha_0 = +- mu_in), then W' = Ltilde^{-1} W down every
  // column independently.  Returns false when a pivot of the downdated factor is not positive
  // (rounding on a nearly singular P_J): the caller re-assembles and re-factorises.
  __device__ __forceinline__ bool pm_rank1(int cid, double sign)
  {
    const int n = d.n, ni = d.n_in;
    gptr W = P.WL();
    lptr pv = L.Hdx(), beta = L.ATdy(); // by-product vectors, idle while the active set is installed
    if (cid < ni) {
      vload(L.t1(), P.Cs() + (long)cid * n, n);
      __syncthreads();
      gemv_dual<NT, false, false, true, true>(W, n, n, n, L.t1(), L.t1(), pv, pv, L.part());
    } else {
      const int k = cid - ni;
      const double ik = L.isc()[k];
      for (int i = threadIdx.x; i < n; i += NT)
        pv[i] = (i >= k) ? ik * W[(long)i * n + k] : 0.0;
      __syncthreads();
    }
    const double inv_a0 = sign * double(info.mu_in); // 1 / alpha_0, alpha_0 = +- 1 / mu_in
    double bad = 0.0;
    // one thread per row; beyond 1024 rows the 1024-thread kernels walk them in chunks (see schur_delete)
    constexpr bool CHUNKED = (NT == 1024) || PQP_CHUNK_ALL;
    double carry = 0.0;
    for (int base = 0; base == 0 || (CHUNKED && base < n); base += NT) {
      const int i_own = base + threadIdx.x;
      double my_p = 0.0, my_d = 1.0, my_e = 0.0;
      if (i_own < n) {
        my_p = pv[i_own];
        my_d = L.dF()[i_own];
        my_e = my_p * my_p / my_d;
      }
      const double incl = carry + block_scan_inclusive<NT>(my_e, L.part());
      if (i_own < n) {
        const double c_i = inv_a0 + incl, c_im1 = c_i - my_e;
        const double dn = my_d * (c_i / c_im1);
        beta[i_own] = my_p / (my_d * c_i);
        L.dF()[i_own] = dn;
        if (!(dn > 0.0) || !(c_i * c_im1 > 0.0))
          bad = 1.0;
      }
      if (CHUNKED && base + NT < n) {
        if (threadIdx.x == NT - 1)
          L.part()[NT / WAVE] = incl;
        __syncthreads();
        carry = L.part()[NT / WAVE];
        __syncthreads();
      }
    }
    bad = R.max(bad);
    for (int c = threadIdx.x; c < n; c += NT) {
      double sacc = 0.0;
      gptr col = W + c;
      constexpr int U = 16;
      for (int i = 0; i < n; i += U) {
        double y[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
          y[u] = col[(long)((i + u < n) ? (i + u) : (n - 1)) * n];
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (i + u < n) {
            const double x = fma(-pv[i + u], sacc, y[u]);
            sacc = fma(beta[i + u], x, sacc);
            if (i + u >= c)
              col[(long)(i + u) * n] = x;
          }
      }
    }
    bytes((long)n * n * 16 + (long)n * 16);
    count(sign > 0 ? ST_N_APPEND : ST_N_DELETE);
    __syncthreads();
    return bad == 0.0;
  }

  // PrimalLDLT form of the KKT solve:  P_J x = bx + B_J^T M^{-1} bd ,  d = M^{-1} (B_J x - bd).
  // Scratch: t1, t2, part and the by-product vectors of kkt_residual (Hdx, ATdy, CTdz, Adx, Cdx),
  // which are idle between two residual evaluations.
  __device__ __forceinline__ void kkt_solve_pm(lptr bx, lptr bd)
  {
    const int n = d.n, ne = d.n_eq, ni = d.n_in, rr = r;
    const double mu_eq_inv = 1.0 / double(info.mu_eq), mu_in_inv = 1.0 / double(info.mu_in);
    lptr sdv = L.t2(); // M^{-1} bd, slot order
    for (int a = threadIdx.x; a < rr; a += NT)
      sdv[a] = bd[a] * ((a < ne) ? mu_eq_inv : mu_in_inv);
    int n_gen = n_c; // active general inequalities (ascending list: they precede the box rows)
    if (has_box()) {
      double cnt = 0;
      for (int a = threadIdx.x; a < n_c; a += NT)
        cnt += (L.act()[a] < ni) ? 1.0 : 0.0;
      n_gen = (int)R.sum(cnt);
    } else {
      __syncthreads();
    }
    if (ne > 0)
      gemv<NT>(P.As(), n, ne, n, sdv, L.Hdx(), L.part(), nullptr, 0, nullptr, 0);
    else
      vzero(L.Hdx(), n);
    if (n_gen > 0)
      gemv<NT>(P.Cs(), n, n_gen, n, sdv + ne, L.ATdy(), L.part(), L.act(), 0, nullptr, 0);
    else
      vzero(L.ATdy(), n);
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += NT)
      L.t1()[k] = bx[k] + L.Hdx()[k] + L.ATdy()[k];
    __syncthreads();
    if (has_box()) {
      for (int a = n_gen + threadIdx.x; a < n_c; a += NT) {
        const int k = L.act()[a] - ni;
        L.t1()[k] += L.isc()[k] * sdv[ne + a];
      }
      __syncthreads();
    }
    // x = P_J^{-1} t1 = W^T D^{-1} W t1
    gemv_dual<NT, false, false, true, true>(P.WL(), n, n, n, L.t1(), L.t1(), L.CTdz(), L.CTdz(), L.part());
    for (int k = threadIdx.x; k < n; k += NT)
      L.CTdz()[k] /= L.dF()[k];
    __syncthreads();
    gemv<NT>(P.WL(), n, n, n, L.CTdz(), bx, L.part(), nullptr, 0, nullptr, 0, -1);
    // d = M^{-1} (B_J x - bd)
    if (ne > 0)
      gemv_dual<NT, false>(P.As(), n, ne, n, bx, bx, L.Adx(), L.Adx(), L.part());
    if (n_gen > 0)
      gemv_dual<NT, false, true>(P.Cs(), n, n_gen, n, bx, bx, L.Cdx(), L.Cdx(), L.part(), L.act(), 0);
    __syncthreads();
    for (int a = threadIdx.x; a < rr; a += NT) {
      double bxv;
      if (a < ne)
        bxv = L.Adx()[a];
      else if (a - ne < n_gen)
        bxv = L.Cdx()[a - ne];
      else {
        const int k = L.act()[a - ne] - ni;
        bxv = L.isc()[k] * bx[k];
      }
      bd[a] = (bxv - bd[a]) * ((a < ne) ? mu_eq_inv : mu_in_inv);
    }
    __syncthreads();
    bytes(((long)n * (n + 1) + 2L * ne * n + 2L * n_gen * n) * 8);
    count(ST_N_KKT_SOLVES);
  }

  // Solve K [sx; sd] = [bx; bd] in place, K = [[H_s+rho I, B_J^T],[B_J, -M_J]].
  // (reference solver.hpp:320-335 -> ldlt.hpp:767-782)
  __device__ __forceinline__ void kkt_solve_in_place(lptr bx, lptr bd)
  {
    const int n = d.n;
    const int rr = r;
    if (pm()) {
      kkt_solve_pm(bx, bd);
      return;
    }
    if (dm()) {
      // diagonal structure: K = [[D, Z_J^T], [Z_J, -mu I]] with one entry per row of Z_J and at
      // most one active row per variable -- the block elimination of the general path, element-wise
      cgptr zd = P.Zr();
      for (int a = threadIdx.x; a < rr; a += NT) {
        const int cid = L.act()[a];
        const int k = dcol(cid);
        const double za = zd[cid];
        const double s = za * (bx[k] / L.dF()[k]) - bd[a];
        bd[a] = s / L.dS()[a];
      }
      __syncthreads();
      for (int k = threadIdx.x; k < n; k += NT)
        L.t1()[k] = bx[k];
      __syncthreads();
      for (int a = threadIdx.x; a < rr; a += NT) {
        const int cid = L.act()[a];
        L.t1()[dcol(cid)] -= zd[cid] * bd[a]; // (one active row per variable: no two slots share k)
      }
      __syncthreads();
      for (int k = threadIdx.x; k < n; k += NT)
        bx[k] = L.t1()[k] / L.dF()[k];
      __syncthreads();
      bytes((long)rr * 16);
      count(ST_N_KKT_SOLVES);
      return;
    }
    // (rounds 1-3 touched every cache line of the Schur factor's triangle here, ahead of the two passes over it:
    // measured in round 4, most of those lines are evicted again before their pass and fetched twice --
    // profiles/r04_ab_c2_traffic.txt: -0.75 MB of HBM traffic per QP and +1.3 % QPs/s without the touch)
    apply_Linv(bx, L.t1(), false, L.t2(), L.dF()); // t = L^{-1} bx ; t2 = t / D
    if (rr > 0) {
      // s_a = z_a . (t / D) - bd_a : row sums over the ACTIVE rows of Zr (contiguous rows; a column
      // gather of Zc would touch every cache line of that matrix).  `part` is free scratch here
      // (gemv_dual only uses it for column sums) and holds at least n_d doubles.
      // (written in place: bd_a <- z_a . t2 - bd_a; the holes are zeroed afterwards)
      gemv_dual<NT, false, true>(P.Zr(), n, rr, n, L.t2(), L.t2(), bd, bd, L.part(), L.act(), d.n_eq, EPI_ROW_RSUB);
      zero_holes(bd);
      // (M + G) dvec = s
      toc(ST_CYC_KKT_SOLVE);
      schur_apply(bd);
      toc(ST_CYC_SOLVE_LDLT);
      // t <- (t - sum_a z_a dvec_a) / D     (gather of the active rows of Zr)
      // t1 <- (t1 - Z_J^T dvec) / D in the epilogue of the column sums
      gemv_dual<NT, true, true, false>(P.Zr(), n, rr, n, bd, bd, L.t1(), L.t1(), L.part(), L.act(), d.n_eq,
                                       EPI_COL_SUBDIV, L.t1(), L.dF());
    } else {
      vcopy(L.t1(), L.t2(), n);
      __syncthreads();
    }
    apply_Linv(L.t1(), bx, true); // x = L^{-T} (.)
    bytes((long)n * (n + 1) * 8 + (long)rr * n * 16);
    count(ST_N_KKT_SOLVES);
  }

  // err = rhs - K * sol for sol = (L.dx(), L.sd()); by-products Hdx, Adx, ATdy, the
  // ACTIVE part of C^T dz in CTdz and C dx for all rows in Cdx (solver.hpp:243-318)
  // (L.zfull() must hold the inequality part of the solution by constraint id, zero where inactive:
  // iterative_solve scatters it while it accumulates the solution.  Returns the infinity norm of err.)
  __device__ __forceinline__ double kkt_residual()
  {
    const int n = d.n, ne = d.n_eq, ni = d.n_in, nc = d.nc;
    const double rho = info.rho;
    if (hess() == PQP_HESSIAN_DENSE) {
      hess_mv(L.dx(), L.Hdx());
    } else {
      cgptr Hs = P.Hs();
      cgptr hd = P.F();
      for (int k = threadIdx.x; k < n; k += NT)
        L.Hdx()[k] = (hess() == PQP_HESSIAN_DIAGONAL) ? (dm() ? hd[k] : Hs[(long)k * n + k]) * L.dx()[k] : 0.0;
    }
    if (ne > 0) {
      // A is read ONCE: row sums give A dx, column sums A^T dy
      gemv_dual<NT>(P.As(), n, ne, n, L.dx(), L.sd(), L.Adx(), L.ATdy(), L.part());
    } else {
      vzero(L.ATdy(), n);
    }
    if (ni > 0 && dm()) {
      cgptr cd = P.CTs();
      for (int k = threadIdx.x; k < n; k += NT) {
        const double ck = cd[k];
        L.Cdx()[k] = ck * L.dx()[k];
        L.CTdz()[k] = ck * L.zfull()[k];
      }
    } else if (ni > 0) {
      gemv_dual<NT>(P.Cs(), n, ni, n, L.dx(), L.zfull(), L.Cdx(), L.CTdz(), L.part());
    } else {
      vzero(L.CTdz(), n);
    }
    __syncthreads();
    if (has_box()) {
      for (int k = threadIdx.x; k < n; k += NT) {
        L.CTdz()[k] += L.zfull()[ni + k] * L.isc()[k];
        L.Cdx()[ni + k] = L.dx()[k] * L.isc()[k];
      }
      __syncthreads();
    }
    // err and its norm in one pass (the reduction's barrier publishes ex / ed; holes hold zeros)
    double m = 0;
    for (int k = threadIdx.x; k < n; k += NT) {
      const double e = L.rx()[k] - rho * L.dx()[k] - L.Hdx()[k] - L.ATdy()[k] - L.CTdz()[k];
      L.ex()[k] = e;
      m = fmax(m, fabs(e));
    }
    for (int k = threadIdx.x; k < ne; k += NT) {
      const double e = L.rd()[k] - L.Adx()[k] + L.sd()[k] * info.mu_eq;
      L.ed()[k] = e;
      m = fmax(m, fabs(e));
    }
    for (int i = threadIdx.x; i < nc; i += NT) {
      int s = L.slot_of()[i];
      if (s >= 0) {
        const double e = L.rd()[ne + s] - (L.Cdx()[i] - L.sd()[ne + s] * info.mu_in);
        L.ed()[ne + s] = e;
        m = fmax(m, fabs(e));
      }
    }
    const double nrm = R.max(m);
    zero_holes(L.ed());
    {
      const long mats = 1; // one pass over A_s / C_s
      bytes((((hess() == PQP_HESSIAN_DENSE) ? hess_pass_elems() : (long)n) +
             (dm() ? (long)ni : mats * ((long)ne * n + (long)ni * n))) * 8);
    }
    return nrm;
  }

  // reference solver.hpp:406-541: solve + iterative refinement on the unfactorised
  // operator.  In: rhs in (L.rx(), L.rd()).  Out: solution in (L.dx(), L.sd()).
  // Returns true when refinement missed eps on a Schur factor that rank-1 sweeps have edited since
  // its last full factorisation: the caller then rebuilds the factor and repeats the solve once --
  // the reference's fallback (solver.hpp:474-532: refactorize, solve again).  A fresh factor has
  // nothing to rebuild.
  __device__ __forceinline__ bool iterative_solve(double eps)
  {
    const int n = d.n;
    vzero(L.dx(), n);
    vzero(L.sd(), r);
    vcopy(L.ex(), L.rx(), n);
    vcopy(L.ed(), L.rd(), r);
    __syncthreads();
    long it = 0, it_stability = 0;
    UD preverr = 0, cur = 0;
    while (true) {
      tic();
      {
        const int nrep = (it == 0) ? reps(3) : 1;
        for (int rp = 0; rp < nrep; ++rp) {
          if (rp > 0) { // (harness: the first solve of the step again, from the same right-hand side)
            vcopy(L.ex(), L.rx(), n);
            vcopy(L.ed(), L.rd(), r);
            __syncthreads();
          }
          kkt_solve_in_place(L.ex(), L.ed());
        }
      }
      for (int k = threadIdx.x; k < n; k += NT)
        L.dx()[k] += L.ex()[k];
      // the dual part of the solution, also scattered by constraint id for the residual's C^T dz
      // (zero where a constraint is inactive; hole slots carry zeros and own no constraint)
      for (int a = threadIdx.x; a < r; a += NT) {
        const double v = L.sd()[a] + L.ed()[a];
        L.sd()[a] = v;
        if (a >= d.n_eq && slot_live(a))
          L.zfull()[L.act()[a - d.n_eq]] = v;
      }
      for (int i = threadIdx.x; i < d.nc; i += NT)
        if (L.slot_of()[i] < 0)
          L.zfull()[i] = 0.0;
      __syncthreads();
      toc(ST_CYC_KKT_SOLVE);
      for (int rp = 0; rp < reps(2); ++rp)
        cur = kkt_residual();
      toc(ST_CYC_RESIDUAL);
      ++it;
#ifdef PQP_TRACE
      if (threadIdx.x == 0)
        printf("  refine q=%ld it=%ld r=%d n_c=%d slots=%d err=%.3e eps=%.1e\n", q, it, r, n_c, n_slots, (double)cur, eps);
#endif
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
    // (:474: the fallback triggers on err >= max(eps, eps_refact); a factor without edits since its last
    // full factorisation would be rebuilt to the same bits, so only an edited one is worth the repeat)
    return (cur >= fmax(eps, st.eps_refact)) && schur_incremental && (r > 0 || pm());
  }

  // New active set from L.aflags (bit 2 = wanted active).  (reference linesearch.hpp:549-786
  // active_set_change: deletions first, then insertions at the end of the factor)
  //   * small change on a valid factor: the inverse Schur factor is edited in place, one rank-1
  //     sweep per constraint -- schur_delete for the ones that leave (their slot stays as a hole),
  //     schur_append for the ones that enter (new last slot);
  //   * otherwise (first factorisation, mu update, many changes, no room for more slots): the slot
  //     map is rebuilt in ascending constraint order by a block prefix scan and the block is
  //     re-factorised (factor_schur).
  __device__ __forceinline__ void apply_active_set()
  {
    const int nc = d.nc, ne = d.n_eq;
    tic();
    // what changes
    double n_add = 0, n_rm = 0;
    for (int i = threadIdx.x; i < nc; i += NT) {
      const bool want = (L.aflags()[i] & 4) != 0;
      const bool had = L.slot_of()[i] >= 0;
      n_add += (want && !had) ? 1.0 : 0.0;
      n_rm += (!want && had) ? 1.0 : 0.0;
    }
    R.sum2(n_add, n_rm);
    const int na = (int)n_add, nr = (int)n_rm;
    if (na + nr == 0 && !schur_dirty) { // the factor already describes this set
      toc(ST_CYC_ZG);
      return;
    }
    // (holes are dead weight in every solve: past the same limit the block is re-packed)
    if (pm() && !schur_dirty && (na + nr) <= PQP_INCR_BASE) {
      // PrimalLDLT: the factor of P_J does not depend on the slot order, so the slot map is simply
      // re-packed; the factor takes one rank-1 update per constraint that left or entered
      int tot_rm = 0, tot_add = 0;
      for (int base = 0; base < nc; base += NT) {
        const int i = base + threadIdx.x;
        const bool want = (i < nc) && ((L.aflags()[i] & 4) != 0);
        const bool had = (i < nc) && (L.slot_of()[i] >= 0);
        int t1, t2;
        const int rk1 = block_rank<NT>(!want && had, L.icnt(), t1);
        if (!want && had)
          L.chg()[tot_rm + rk1] = i;
        const int rk2 = block_rank<NT>(want && !had, L.icnt(), t2);
        if (want && !had)
          L.chg()[INCR_MAX + tot_add + rk2] = i;
        tot_rm += t1;
        tot_add += t2;
      }
      int total = 0;
      for (int base = 0; base < nc; base += NT) {
        int i = base + threadIdx.x;
        bool want = (i < nc) && ((L.aflags()[i] & 4) != 0);
        int tot;
        int rank = block_rank<NT>(want, L.icnt(), tot);
        if (want) {
          L.slot_of()[i] = total + rank;
          L.act()[total + rank] = i;
        } else if (i < nc) {
          L.slot_of()[i] = -1;
        }
        total += tot;
      }
      __syncthreads();
      n_c = total;
      n_slots = total;
      r = ne + n_slots;
      toc(ST_CYC_ZG);
      bool ok = true;
      for (int t = 0; t < nr && ok; ++t)
        ok = pm_rank1(uni(L.chg()[t]), -1.0);
      for (int t = 0; t < na && ok; ++t)
        ok = pm_rank1(uni(L.chg()[INCR_MAX + t]), +1.0);
      schur_incremental = true;
      toc(ST_CYC_SCHUR);
      if (PQP_LIKELY(ok))
        return;
      schur_dirty = true; // a pivot went non-positive: the full path below re-assembles P_J
      tic();
    }
    const int lim = incr_max(r); // edits allowed in one change, and holes tolerated in the factor
    const bool incremental = !dm() && !pm() && !schur_dirty && (na + nr) <= lim && n_slots + na <= nc &&
                             (n_slots - n_c) + nr <= lim;
    if (INCR_MAX > 0 && incremental) {
      // ids that leave -> chg[0 .. nr), ids that enter -> chg[INCR_MAX .. INCR_MAX + na), ascending
      int tot_rm = 0, tot_add = 0;
      for (int base = 0; base < nc; base += NT) {
        const int i = base + threadIdx.x;
        const bool want = (i < nc) && ((L.aflags()[i] & 4) != 0);
        const bool had = (i < nc) && (L.slot_of()[i] >= 0);
        int t1, t2;
        const int rk1 = block_rank<NT>(!want && had, L.icnt(), t1);
        if (!want && had)
          L.chg()[tot_rm + rk1] = i;
        const int rk2 = block_rank<NT>(want && !had, L.icnt(), t2);
        if (want && !had)
          L.chg()[INCR_MAX + tot_add + rk2] = i;
        tot_rm += t1;
        tot_add += t2;
      }
      __syncthreads();
      toc(ST_CYC_ZG);
      bool ok = true;
      for (int t = 0; t < nr; ++t) {
        const int i = uni(L.chg()[t]);
        const int sl = uni(L.slot_of()[i]);
        __syncthreads();
        if (threadIdx.x == 0)
          L.slot_of()[i] = -1;
        schur_delete(sl);
        n_c -= 1;
      }
      for (int t = 0; t < na; ++t) {
        const int i = uni(L.chg()[INCR_MAX + t]);
        ok = schur_append(i) && ok;
        n_c += 1;
      }
      schur_incremental = true;
      toc(ST_CYC_SCHUR);
      debug_check_factor("incremental");
      if (PQP_LIKELY(ok))
        return;
      // a non-positive pivot came out of an append: fall through to the full factorisation of
      // the set that is now installed (aflags still describe it)
      schur_dirty = true;
      tic();
    }
    int total = 0;
    for (int base = 0; base < nc; base += NT) {
      int i = base + threadIdx.x;
      bool want = (i < nc) && ((L.aflags()[i] & 4) != 0);
      int tot;
      int rank = block_rank<NT>(want, L.icnt(), tot);
      if (want) {
        L.slot_of()[i] = total + rank;
        L.act()[total + rank] = i;
      } else if (i < nc) {
        L.slot_of()[i] = -1;
      }
      total += tot;
    }
    __syncthreads(); // slot_of / act complete before the gather reads them
    n_c = total;
    n_slots = total;
    r = ne + n_slots;
    schur_dirty = true; // the slots were renumbered
    toc(ST_CYC_ZG);
    if (r > 0 || pm()) { // (PrimalLDLT factorises P_J even with no constraint in it)
      for (int rp = 0; rp < reps(4); ++rp)
        factor_schur();
    }
    else
      schur_dirty = false;
    toc(ST_CYC_SCHUR);
  }

  // reference utils.hpp:164-252
  __device__ __forceinline__ void global_primal_residual(UD& lhs, UD& eq_rhs_0, UD& in_rhs_0, UD& eq_lhs,
                                                         UD& in_lhs)
  {
    const int n = d.n, ne = d.n_eq, ni = d.n_in;
    double m_eq0 = 0, m_in0 = 0, m_eql = 0, m_inl = 0;
    if (dm()) {
      vzero(L.ATdy(), n);
      if (ni > 0) {
        cgptr cd = P.CTs();
        for (int k = threadIdx.x; k < n; k += NT) {
          L.rup()[k] = cd[k] * L.x()[k];
          L.CTdz()[k] = cd[k] * L.z()[k];
        }
      } else {
        vzero(L.CTdz(), n);
      }
      aty_fresh = true;
      __syncthreads();
    } else if (iterate_zero) {
      // (the products of a zero iterate: no pass over A_s / C_s)
      vzero(L.se(), ne);
      vzero(L.rup(), ni);
      vzero(L.ATdy(), n);
      vzero(L.CTdz(), n);
      aty_fresh = true;
      __syncthreads();
    } else {
      // one pass over A_s and C_s: the row sums are A x / C x; the column sums A^T y / C^T z are
      // what global_dual_residual needs at this same iterate, parked in the Newton by-product
      // vectors (idle between Newton loops) and flagged by `aty_fresh`
      if (ne > 0)
        gemv_dual<NT>(P.As(), n, ne, n, L.x(), L.y(), L.se(), L.ATdy(), L.part());
      else
        vzero(L.ATdy(), n);
      if (ni > 0)
        gemv_dual<NT>(P.Cs(), n, ni, n, L.x(), L.z(), L.rup(), L.CTdz(), L.part());
      else
        vzero(L.CTdz(), n);
      aty_fresh = true;
    }
    {
      cgptr de = P.dlt_eq();
      cgptr bb = P.bvec();
      for (int j = threadIdx.x; j < ne; j += NT) {
        double v = L.se()[j] / de[j]; // unscaled A x
        m_eq0 = fmax(m_eq0, fabs(v));
        v -= bb[j];
        m_eql = fmax(m_eql, fabs(v));
        L.se()[j] = v;
      }
      cgptr di = P.dlt_in();
      cgptr uu = P.u(), ll = P.l();
      for (int j = threadIdx.x; j < ni; j += NT) {
        double v = L.rup()[j] / di[j]; // unscaled C x
        L.rup()[j] = v;
        m_in0 = fmax(m_in0, fabs(v));
        double pu = v - uu[j], pl = v - ll[j];
        double sv = (pu > 0 ? pu : 0.0) + (pl < 0 ? pl : 0.0);
        L.si()[j] = sv;
        m_inl = fmax(m_inl, fabs(sv));
      }
    }
    if (has_box()) {
      cgptr dx = P.dlt_x();
      cgptr ub = P.u_box(), lb = P.l_box();
      for (int k = threadIdx.x; k < n; k += NT) {
        double v = L.x()[k] * dx[k]; // unscaled x
        L.rup()[ni + k] = v;
        double pu = v - ub[k], pl = v - lb[k];
        double sv = (pu > 0 ? pu : 0.0) + (pl < 0 ? pl : 0.0);
        L.si()[ni + k] = sv;
        m_inl = fmax(m_inl, fabs(sv));
        m_in0 = fmax(m_in0, fabs(L.x()[k] - sv)); // utils.hpp:225-229 (as written)
        m_in0 = fmax(m_in0, fabs(L.x()[k]));      // utils.hpp:230-231
      }
    }
    if (!iterate_zero)
      bytes(dm() ? (long)ni * 8 : ((long)ne * n + (long)ni * n) * 8);
    {
      double none[1] = { 0.0 };
      double mv[4] = { m_eq0, m_in0, m_eql, m_inl };
      R.template mixed<0, 4>(none, mv);
      eq_rhs_0 = mv[0];
      in_rhs_0 = mv[1];
      eq_lhs = mv[2];
      in_lhs = mv[3];
    }
    lhs = fmax(eq_lhs, in_lhs);
    if (PQP_UNLIKELY(st.primal_infeasibility_solving && info.status == PQP_PRIMAL_INFEASIBLE)) {
      // utils.hpp:241-248 : || A^T se + C^T si ||_inf on the unscaled model
      __syncthreads();
      vzero(L.t1(), n);
      vzero(L.t2(), n);
      __syncthreads();
      if (ne > 0)
        mv(P.A(), n, ne, n, L.se(), L.t1());
      if (ni > 0)
        mv(P.C(), n, ni, n, L.si(), L.t2());
      double m = 0;
      for (int k = threadIdx.x; k < n; k += NT)
        m = fmax(m, fabs(L.t1()[k] + L.t2()[k]));
      lhs = R.max(m);
    }
    {
      cgptr de = P.dlt_eq();
      for (int k = threadIdx.x; k < ne; k += NT)
        L.se()[k] *= de[k];
    }
    __syncthreads();
  }

  // reference utils.hpp:437-587
  __device__ __forceinline__ void global_dual_residual(UD& lhs, UD& rhs_0, UD& rhs_1, UD& rhs_3,
                                                       UD& rhs_duality_gap, UD& duality_gap)
  {
    const int n = d.n, ne = d.n_eq, ni = d.n_in;
    const double c = ruiz_c;
    cgptr dx = P.dlt_x();
    double m0 = 0, m1 = 0, m3 = 0, ml = 0;
    double xHx = 0, gx = 0;
    // H x -> t1, A^T y -> ATdy-free scratch (t2), C^T z -> CTzin-free... use t1/t2/zfull? keep
    // three distinct n-vectors: t1, t2 and ex (free outside the Newton loop)
    if (iterate_zero) {
      vzero(L.t1(), n); // H 0
    } else if (hess() == PQP_HESSIAN_DENSE) {
      hess_mv(L.x(), L.t1());
    } else {
      cgptr Hs = P.Hs();
      cgptr hd = P.F();
      for (int k = threadIdx.x; k < n; k += NT)
        L.t1()[k] = (hess() == PQP_HESSIAN_DIAGONAL) ? (dm() ? hd[k] : Hs[(long)k * n + k]) * L.x()[k] : 0.0;
    }
    const bool have_products = aty_fresh; // A^T y, C^T z left by global_primal_residual
    bytes(((iterate_zero ? 0L : (hess() == PQP_HESSIAN_DENSE ? hess_pass_elems() : (long)n)) +
           (have_products ? 0L : (long)ne * n + (long)ni * n)) * 8);
    if (!have_products) {
      if (ne > 0)
        mv(P.As(), n, ne, n, L.y(), L.t2());
      else
        vzero(L.t2(), n);
      if (ni > 0 && dm()) {
        cgptr cd = P.CTs();
        for (int k = threadIdx.x; k < n; k += NT)
          L.ex()[k] = cd[k] * L.z()[k];
      } else if (ni > 0)
        mv(P.Cs(), n, ni, n, L.z(), L.ex());
      else
        vzero(L.ex(), n);
    }
    __syncthreads();
    {
      cgptr g = P.g();
      clptr v_aty = have_products ? (clptr)L.ATdy() : (clptr)L.t2();
      clptr v_ctz = have_products ? (clptr)L.CTdz() : (clptr)L.ex();
      for (int k = threadIdx.x; k < n; k += NT) {
        const double sc = dx[k] * c;
        double hx = L.t1()[k], aty = v_aty[k], ctz = v_ctz[k];
        double v = hx / sc; // unscaled H x (utils.hpp:469-471)
        m0 = fmax(m0, fabs(v));
        double xu = L.x()[k] * dx[k];
        xHx += v * xu;
        gx += g[k] * xu;
        m1 = fmax(m1, fabs(aty / sc));
        double m3k = fabs(ctz / sc);
        if (has_box()) {
          double zb = L.z()[ni + k] * L.isc()[k];
          ctz += zb;
          m3k = fmax(m3k, fabs(zb / sc));
        }
        m3 = fmax(m3, m3k);
        double dr = L.gs()[k] + hx + aty + ctz;
        L.dres()[k] = dr;
        ml = fmax(ml, fabs(dr / sc));
      }
    }
    // duality gap terms (utils.hpp:482-586); the four norms above and the seven sums below close in
    // ONE fused reduction
    double by = 0, zu = 0, zl = 0, zub = 0, zlb = 0;
    const double ib = 1.3407807929942596e+154; // sqrt(DBL_MAX), helpers/common.hpp:17-25
    {
      cgptr de = P.dlt_eq();
      cgptr bb = P.bvec();
      for (int k = threadIdx.x; k < ne; k += NT)
        by += bb[k] * (L.y()[k] * de[k] / c);
      cgptr di = P.dlt_in();
      cgptr uu = P.u(), ll = P.l();
      for (int k = threadIdx.x; k < ni; k += NT) {
        double zi = L.z()[k] * di[k] / c;
        double uk = uu[k] < ib ? uu[k] : ib;
        double lk = ll[k] > -ib ? ll[k] : -ib;
        if (flag_up(k))
          zu += zi * uk;
        if (flag_low(k))
          zl += zi * lk;
      }
      if (has_box()) {
        cgptr db = P.dlt_box();
        cgptr ub = P.u_box(), lb = P.l_box();
        for (int k = threadIdx.x; k < n; k += NT) {
          double zi = db[k] * L.z()[ni + k] / c;
          double uk = ub[k] < ib ? ub[k] : ib;
          double lk = lb[k] > -ib ? lb[k] : -ib;
          if (flag_up(ni + k))
            zub += zi * uk;
          if (flag_low(ni + k))
            zlb += zi * lk;
        }
      }
    }
    {
      double sv[7] = { gx, xHx, by, zu, zl, zub, zlb };
      double mv[4] = { m0, m1, m3, ml };
      R.template mixed<7, 4>(sv, mv);
      gx = sv[0];
      xHx = sv[1];
      by = sv[2];
      zu = sv[3];
      zl = sv[4];
      zub = sv[5];
      zlb = sv[6];
      rhs_0 = (hess() == PQP_HESSIAN_ZERO) ? 0.0 : mv[0];
      rhs_1 = mv[1];
      rhs_3 = mv[2];
      lhs = mv[3];
    }
    duality_gap = gx;
    rhs_duality_gap = fabs(gx);
    if (hess() != PQP_HESSIAN_ZERO) {
      duality_gap += xHx;
      rhs_duality_gap = fmax(rhs_duality_gap, fabs(xHx));
    }
    rhs_duality_gap = fmax(rhs_duality_gap, fabs(by));
    duality_gap += by;
    rhs_duality_gap = fmax(rhs_duality_gap, fabs(zu));
    duality_gap += zu;
    rhs_duality_gap = fmax(rhs_duality_gap, fabs(zl));
    duality_gap += zl;
    if (has_box()) {
      rhs_duality_gap = fmax(rhs_duality_gap, fabs(zub));
      duality_gap += zub;
      rhs_duality_gap = fmax(rhs_duality_gap, fabs(zlb));
      duality_gap += zlb;
    }
  }

  // ---- exact line search (reference linesearch.hpp:320-538; merit terms GPDAL
  // :49-167 / PDAL :178-311).  The inequality part of (a, b) for one alpha, as a
  // serial loop over the constraints run by the thread that owns that alpha.
  __device__ __forceinline__ void ls_ineq_terms(double alpha, double& a_in, double& b_in)
  {
    const int nc = d.nc;
    const bool gpdal = st.merit_function_type == PQP_MERIT_GPDAL;
    double sa = 0, sb = 0, sa2 = 0, sb2 = 0;
    clptr vCdx = L.Cdx(), vrup = L.rup(), vsi = L.si(), vdz = L.dz(), vz = L.z();
    // (four constraints per trip: their LDS reads are issued together -- one trip of this loop is a read latency long --
    // while the sums still take their terms in the order of the constraints)
#pragma unroll PQP_LS_UNROLL
    for (int i = 0; i < nc; ++i) {
      double cdx = vCdx[i];
      double up0 = vrup[i], lo0 = vsi[i];
      bool up = (up0 + cdx * alpha) > 0.;
      bool lo = (lo0 + cdx * alpha) < 0.;
      double e = (up || lo) ? cdx : 0.0;
      double apz = (up ? up0 : 0.0) + (lo ? lo0 : 0.0);
      sa = fma(e, e, sa);
      sb = fma(apz, e, sb);
      if (!gpdal) {
        double e2 = e - vdz[i] * info.mu_in;
        double apz2 = apz - vz[i] * info.mu_in;
        sa2 = fma(e2, e2, sa2);
        sb2 = fma(e2, apz2, sb2);
      }
    }
    if (gpdal) {
      a_in = info.mu_in_inv * sa / st.alpha_gpdal;
      b_in = info.mu_in_inv * sb / st.alpha_gpdal;
    } else {
      a_in = info.mu_in_inv * sa + info.nu * info.mu_in_inv * sa2;
      b_in = info.mu_in_inv * sb + info.nu * info.mu_in_inv * sb2;
    }
  }

  // phi'(alpha) at three step lengths with the inequality sums spread over the workgroup (thread order: NOT the
  // reference's order of summation -- these values only steer the bracket), plus the number of this thread's breakpoints
  // in (lo, al[p]] and in (lo, hi], all in ONE fused reduction.  `mag` bounds the size of the terms of each value.
  // `bmag`: an alpha-independent bound on the sum of the MAGNITUDES of the terms of the b-sums (they cancel; the a-sums
  // are sums of squares), see primal_dual_ls.
  __device__ __forceinline__ void ls_grad3(const double (&al)[3], double a0, double b0, double bmag, const double (&mine)[2],
                                           double lo, double hi, double (&g)[3], double (&mag)[3], double (&cle)[3], double& ctot)
  {
    const int nc = d.nc;
    const bool gpdal = st.merit_function_type == PQP_MERIT_GPDAL;
    double sv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k)
      sv[k] = 0.0;
    for (int i = threadIdx.x; i < nc; i += NT) {
      const double cdx = L.Cdx()[i], up0 = L.rup()[i], lo0 = L.si()[i];
      const double dzi = L.dz()[i] * info.mu_in, zi = L.z()[i] * info.mu_in;
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const bool up = (up0 + cdx * al[p]) > 0.;
        const bool lw = (lo0 + cdx * al[p]) < 0.;
        const double e = (up || lw) ? cdx : 0.0;
        const double apz = (up ? up0 : 0.0) + (lw ? lo0 : 0.0);
        sv[4 * p + 0] = fma(e, e, sv[4 * p + 0]);
        sv[4 * p + 1] = fma(apz, e, sv[4 * p + 1]);
        if (!gpdal) {
          const double e2 = e - dzi, apz2 = apz - zi;
          sv[4 * p + 2] = fma(e2, e2, sv[4 * p + 2]);
          sv[4 * p + 3] = fma(e2, apz2, sv[4 * p + 3]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
      if (mine[r] > lo && mine[r] <= hi) {
        sv[15] += 1.0;
#pragma unroll
        for (int p = 0; p < 3; ++p)
          if (mine[r] <= al[p])
            sv[12 + p] += 1.0;
      }
    double none[1] = { 0.0 };
    if (gpdal) {
      // (GPDAL: the second pair of sums does not exist -- ten values instead of sixteen through the reduction)
      double pk[10] = { sv[0], sv[1], sv[4], sv[5], sv[8], sv[9], sv[12], sv[13], sv[14], sv[15] };
      R.template mixed<10, 0>(pk, none);
      sv[0] = pk[0];
      sv[1] = pk[1];
      sv[4] = pk[2];
      sv[5] = pk[3];
      sv[8] = pk[4];
      sv[9] = pk[5];
      sv[12] = pk[6];
      sv[13] = pk[7];
      sv[14] = pk[8];
      sv[15] = pk[9];
    } else {
      R.template mixed<16, 0>(sv, none);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      double ai, bi;
      if (gpdal) {
        ai = info.mu_in_inv * sv[4 * p + 0] / st.alpha_gpdal;
        bi = info.mu_in_inv * sv[4 * p + 1] / st.alpha_gpdal;
      } else {
        ai = info.mu_in_inv * sv[4 * p + 0] + info.nu * info.mu_in_inv * sv[4 * p + 2];
        bi = info.mu_in_inv * sv[4 * p + 1] + info.nu * info.mu_in_inv * sv[4 * p + 3];
      }
      g[p] = (a0 + ai) * al[p] + (b0 + bi);
      mag[p] = fabs((a0 + ai) * al[p]) + fabs(b0) + bmag; // (bmag >= |bi|)
      cle[p] = sv[12 + p];
    }
    ctot = sv[15];
  }

  // The exact line search through a bracket of its zero (see primal_dual_ls).  Returns false when the full evaluation
  // has to decide; true with the step length otherwise -- the SAME floating-point value the full evaluation returns,
  // because the two breakpoints it interpolates between and their phi' are found and evaluated identically.
  __device__ __forceinline__ bool ls_bracket(double a0, double b0, double bmag, double& result)
  {
    const int nc = d.nc;
    const double INF = __builtin_inf();
    constexpr int WCAP = 32;          // breakpoints evaluated exactly at most (one lane each, one wavefront)
    // |phi'| above this fraction of the size of its terms: the sign is trusted.  Two orders of summation of N terms differ
    // by at most 2 N u sum|terms| (u = 1.1e-16); `mag` carries that sum of magnitudes -- bmag for the b-terms, which cancel
    // (|b_in| itself underestimates it) -- and the factor below is sixteen times the bound
    const double SURE = 3.6e-15 * (double)(nc + d.n + d.n_eq);
    // this thread's breakpoints (as in the full evaluation below)
    double mine[2] = { -1.0, -1.0 };
    double cnt = 0, amax = 0, amin_neg = -INF;
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      const int t = threadIdx.x + rep * NT;
      if (t < 2 * nc) {
        const int i = t >> 1;
        const double cdx = L.Cdx()[i];
        double al = -1.0;
        if (cdx != 0.) {
          const double num = (t & 1) ? L.si()[i] : L.rup()[i];
          al = -num / (cdx + MACHINE_EPS);
        }
        if (al > MACHINE_EPS) {
          mine[rep] = al;
          cnt += 1.0;
          amax = fmax(amax, al);
          amin_neg = fmax(amin_neg, -al);
        }
      }
    }
    {
      double sv[1] = { cnt };
      double mv[2] = { amax, amin_neg };
      R.template mixed<1, 2>(sv, mv);
      cnt = sv[0];
      amax = mv[0];
      amin_neg = mv[1];
    }
    // The breakpoints spread over many decades (ratios of residuals to step components): the bracket is refined on a
    // geometric scale between `floor_` (half the smallest breakpoint: phi' has no kink below it) and hi.
    const double floor_ = -0.5 * amin_neg;
    if (!(cnt > 8.0) || !(amax < INF))
      return false; // few breakpoints (or none: linesearch.hpp:405-419): the full evaluation is as cheap
    // bracket (lo, hi]: phi'(lo) < 0 surely, phi'(hi) > 0 surely
    double lo = 0.0, hi = amax, inside = cnt;
    bool all_negative = false;
    {
      const double al[3] = { 0.0, sqrt(floor_) * sqrt(amax), amax };
      double g[3], mag[3], cle[3], ctot;
      ls_grad3(al, a0, b0, bmag, mine, 0.0, amax, g, mag, cle, ctot);
      if (!(g[0] < -SURE * mag[0]))
        return false;
      if (g[2] < -SURE * mag[2]) {
        all_negative = true; // no breakpoint with phi' >= 0: linesearch.hpp:496-526
      } else if (!(g[2] > SURE * mag[2])) {
        return false;
      } else if (g[1] < -SURE * mag[1]) {
        lo = al[1];
        inside = ctot - cle[1];
      } else if (g[1] > SURE * mag[1]) {
        hi = al[1];
        inside = cle[1];
      }
    }
    if (!all_negative) {
      for (int round = 0; round < 6 && inside > 6.0; ++round) {
        const double base = fmax(lo, floor_);
        const double r4 = sqrt(sqrt(hi / base)); // quarter steps of the logarithm
        const double al[3] = { base * r4, base * r4 * r4, base * r4 * r4 * r4 };
        double g[3], mag[3], cle[3], ctot;
        ls_grad3(al, a0, b0, bmag, mine, lo, hi, g, mag, cle, ctot);
        // the leftmost sure positive closes the bracket, the rightmost sure negative to its left opens it; a value too small
        // to trust (the zero is next to that point) tightens nothing -- a wrong bracket is caught by the exact values below
        double nlo = lo, nhi = hi, below = 0.0, upto = ctot;
        int first_pos = 3;
#pragma unroll
        for (int p = 2; p >= 0; --p)
          if (g[p] > SURE * mag[p])
            first_pos = p;
        if (first_pos < 3) {
          nhi = al[first_pos];
          upto = cle[first_pos];
        }
        bool moved = first_pos < 3;
#pragma unroll
        for (int p = 0; p < 3; ++p)
          if (p < first_pos && g[p] < -SURE * mag[p]) {
            nlo = al[p];
            below = cle[p];
            moved = true;
          }
        const bool stuck = !moved;
        if (!(nlo < nhi))
          return false;
        lo = nlo;
        hi = nhi;
        inside = upto - below;
        if (stuck)
          break;
      }
      if (inside > double(WCAP - 3))
        return false;
    }
    // the breakpoints to evaluate exactly: those in (lo, hi], the last one at or below lo, the first one above hi
    // (all_negative: the largest breakpoint alone)
    double pred = 0.0, succ = INF;
    if (!all_negative) {
      double below = 0.0, above = -INF;
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
        if (mine[rep] > 0) {
          if (mine[rep] <= lo)
            below = fmax(below, mine[rep]);
          if (mine[rep] > hi)
            above = fmax(above, -mine[rep]);
        }
      double none[1] = { 0.0 };
      double mv[2] = { below, above };
      liptr counter = L.iscr();
      if (threadIdx.x == 0)
        counter[0] = 1; // slot 0: alpha = 0 (below)
      R.template mixed<0, 2>(none, mv);
      pred = mv[0];
      succ = -mv[1];
    } else {
      liptr counter = L.iscr();
      if (threadIdx.x == 0)
        counter[0] = 1;
      __syncthreads();
    }
    lptr list = L.part(); // [0, WCAP): alpha, [WCAP, 2 WCAP): phi'
    if (L.part_len() < 2 * WCAP)
      return false;
    {
      liptr counter = L.iscr();
      // slot 0 holds alpha = 0: phi'(0) is what the interpolation starts from when no breakpoint precedes the zero
      // (linesearch.hpp:477-495), evaluated beside the others instead of in a serial loop of its own afterwards
      if (threadIdx.x == 0)
        list[0] = 0.0;
#pragma unroll
      for (int rep = 0; rep < 2; ++rep) {
        const double a = mine[rep];
        const bool take = all_negative ? (a > 0 && a == amax) : (a > 0 && ((a > lo && a <= hi) || a == pred || a == succ));
        if (take) {
          const int slot = atomicAdd((int*)counter, 1);
          if (slot < WCAP)
            list[slot] = a;
        }
      }
      __syncthreads();
    }
    const int nwin = uni(L.iscr()[0]);
    if (nwin > WCAP || nwin <= 1)
      return false;
    count(ST_N_LS_BREAKPOINTS, nwin); // (stats build: breakpoints evaluated exactly)
    // Exact values.  GPDAL merit, at most four candidates: the per-constraint terms (e_i, a_i) of every candidate -- which
    // constraints are active at that step length -- are computed by the whole workgroup into LDS vectors that are dead
    // between two Newton solves (right-hand sides, refinement errors, scratch), and each candidate's lane then only
    // runs the two chains of fused multiply-adds over them, in the order of the constraints: the same operations on the
    // same values as ls_ineq_terms, without the tests and selections inside the serial loop.
    bool chained = false;
    if (st.merit_function_type == PQP_MERIT_GPDAL && nwin <= 4) {
      const int n = d.n;
      int cap = 2; // rd / ed hold n_d >= n_c doubles, t2 max(n, n_d), zfull n_c
      if (n >= nc)
        cap = (L.part_len() - 2 * WCAP >= nc) ? 4 : 3; // rx, ex, t1 hold n doubles; the tail of `part` behind the list
      if (nwin <= cap) {
        chained = true;
        lptr e0 = L.rd(), a0v = L.ed(), e1 = L.t2(), a1v = L.zfull(), e2 = L.rx(), a2v = L.ex(), e3 = L.t1(),
             a3v = L.part() + 2 * WCAP;
        const double al0 = list[0], al1 = list[nwin > 1 ? 1 : 0], al2 = list[nwin > 2 ? 2 : 0], al3 = list[nwin > 3 ? 3 : 0];
        for (int i = threadIdx.x; i < nc; i += NT) {
          const double cdx = L.Cdx()[i], up0 = L.rup()[i], lo0 = L.si()[i];
#define PQP_LS_TERM(alpha, ev, av)                                                                                     \
  {                                                                                                                   \
    const bool up = (up0 + cdx * (alpha)) > 0.;                                                                       \
    const bool lo = (lo0 + cdx * (alpha)) < 0.;                                                                       \
    (ev)[i] = (up || lo) ? cdx : 0.0;                                                                                 \
    (av)[i] = (up ? up0 : 0.0) + (lo ? lo0 : 0.0);                                                                    \
  }
          PQP_LS_TERM(al0, e0, a0v)
          if (nwin > 1)
            PQP_LS_TERM(al1, e1, a1v)
          if (nwin > 2)
            PQP_LS_TERM(al2, e2, a2v)
          if (nwin > 3)
            PQP_LS_TERM(al3, e3, a3v)
#undef PQP_LS_TERM
        }
        __syncthreads();
        if (threadIdx.x < nwin) {
          const int c = threadIdx.x;
          clptr ev = (c == 0) ? (clptr)e0 : (c == 1) ? (clptr)e1 : (c == 2) ? (clptr)e2 : (clptr)e3;
          clptr av = (c == 0) ? (clptr)a0v : (c == 1) ? (clptr)a1v : (c == 2) ? (clptr)a2v : (clptr)a3v;
          double sa = 0, sb = 0;
#pragma unroll 4
          for (int i = 0; i < nc; ++i) {
            const double e = ev[i], apz = av[i];
            sa = fma(e, e, sa);
            sb = fma(apz, e, sb);
          }
          const double ai = info.mu_in_inv * sa / st.alpha_gpdal, bi = info.mu_in_inv * sb / st.alpha_gpdal;
          const double al = list[c];
          list[WCAP + c] = (c == 0) ? (b0 + bi) : ((a0 + ai) * al + (b0 + bi));
        }
      }
    }
    if (!chained && threadIdx.x < nwin) {
      const double al = list[threadIdx.x];
      double ai, bi;
      ls_ineq_terms(al, ai, bi);
      list[WCAP + threadIdx.x] = (threadIdx.x == 0) ? (b0 + bi) : ((a0 + ai) * al + (b0 + bi));
    }
    __syncthreads();
    // every thread reads the short list: the selections of the full evaluation without their reductions
    double afp = INF, gfp = -INF;
    for (int k = 1; k < nwin; ++k) {
      const double al = list[k], gr = list[WCAP + k];
      if (!(gr < 0) && al < afp)
        afp = al;
    }
    if (all_negative) {
      if (afp < INF)
        return false; // the exact value at the largest breakpoint is not negative after all
      const double aln = amax;
      double ai, bi;
      ls_ineq_terms(2 * aln + 1, ai, bi);
      result = -(b0 + bi) / (a0 + ai);
      return true;
    }
    if (!(afp < INF))
      return false;
    if (pred > 0.0 && !(afp > pred))
      return false; // phi'(pred) >= 0 exactly: the zero lies further left than the bracket said
    double aln = 0.0;
    for (int k = 1; k < nwin; ++k) {
      const double al = list[k], gr = list[WCAP + k];
      if (al == afp && !(gr < 0))
        gfp = fmax(gfp, gr);
      if (al < afp)
        aln = fmax(aln, al);
    }
    double gln = -INF;
    for (int k = 1; k < nwin; ++k)
      if (list[k] == aln)
        gln = fmax(gln, list[WCAP + k]);
    const double g_at_0 = list[WCAP];
    __syncthreads(); // (the list lives in `part`: nobody may reuse it before everybody has read it)
    if (aln == 0.0) { // no breakpoint before afp: linesearch.hpp:477-495
      if (pred > 0.0)
        return false;
      gln = g_at_0; // = b0 + b_in(0): see the evaluation of slot 0
    }
    result = fabs(aln - gln * (afp - aln) / (gfp - gln)); // linesearch.hpp:534-536
    return true;
  }

  __device__ __forceinline__ double primal_dual_ls(double& dw_max)
  {
    const int n = d.n, ne = d.n_eq, nc = d.nc;
    const bool gpdal = st.merit_function_type == PQP_MERIT_GPDAL;
    // alpha-independent coefficients
    double s_dxHdx = 0, s_adx2 = 0, s_dx2 = 0, s_e2 = 0;
    double s_xHdx = 0, s_errdx = 0, s_adxres = 0, s_eres = 0;
    double s_dz2 = 0, s_dzz = 0;
    double dwm = 0;
    for (int k = threadIdx.x; k < n; k += NT) {
      double dxk = L.dx()[k];
      dwm = fmax(dwm, fabs(dxk));
      s_dxHdx += dxk * L.Hdx()[k];
      s_dx2 += dxk * dxk;
      s_xHdx += L.x()[k] * L.Hdx()[k];
      s_errdx += (info.rho * (L.x()[k] - L.xp()[k]) + L.gs()[k]) * dxk;
    }
    for (int k = threadIdx.x; k < ne; k += NT) {
      double adx = L.Adx()[k];
      double e = adx - L.dy()[k] * info.mu_eq;
      dwm = fmax(dwm, fabs(L.dy()[k]));
      s_adx2 += adx * adx;
      s_e2 += e * e;
      s_adxres += adx * (L.se()[k] + L.y()[k] * info.mu_eq);
      s_eres += e * L.se()[k];
    }
    // (kernels with the bracket line search: an alpha-independent bound on sum_i |term_i| of the inequality b-sums --
    // apz_i is 0, rup_i, si_i or their sum, e_i is 0 or Cdx_i; PDAL adds (e_i - mu dz_i)(apz_i - mu z_i) -- ls_bracket)
    constexpr bool BRACKET = PQP_LS_BRACKET && SPEC != 1 && NT == 256;
    double s_bmag = 0;
    for (int k = threadIdx.x; k < nc; k += NT) {
      dwm = fmax(dwm, fabs(L.dz()[k]));
      s_dz2 += L.dz()[k] * L.dz()[k];
      s_dzz += L.dz()[k] * L.z()[k];
      if constexpr (BRACKET) {
        const double ac = fabs(L.Cdx()[k]), ar = fabs(L.rup()[k]) + fabs(L.si()[k]);
        s_bmag = fma(ar, ac, s_bmag);
        if (!gpdal)
          s_bmag = fma(double(info.nu) * (ar + fabs(L.z()[k]) * info.mu_in), ac + fabs(L.dz()[k]) * info.mu_in, s_bmag);
      }
    }
    {
      // the ten coefficient sums in one barrier interval (+ the bound above where the bracket exists)
      double sv[BRACKET ? 11 : 10] = { s_dxHdx, s_adx2, s_dx2, s_e2, s_xHdx, s_errdx, s_adxres, s_eres, s_dz2, s_dzz };
      if constexpr (BRACKET)
        sv[10] = s_bmag;
      double mx[1] = { dwm };
      R.template mixed<(BRACKET ? 11 : 10), 1>(sv, mx);
      if constexpr (BRACKET)
        s_bmag = sv[10];
      dw_max = mx[0];
      s_dxHdx = sv[0];
      s_adx2 = sv[1];
      s_dx2 = sv[2];
      s_e2 = sv[3];
      s_xHdx = sv[4];
      s_errdx = sv[5];
      s_adxres = sv[6];
      s_eres = sv[7];
      s_dz2 = sv[8];
      s_dzz = sv[9];
    }
    const double nu = gpdal ? 1.0 : double(info.nu);
    double a0 = s_dxHdx + info.mu_eq_inv * s_adx2 + info.rho * s_dx2 + s_e2 * info.mu_eq_inv * nu;
    double b0 = s_xHdx + s_errdx + info.mu_eq_inv * s_adxres + nu * info.mu_eq_inv * s_eres;
    if (gpdal) {
      a0 += info.mu_in * (1. - st.alpha_gpdal) * s_dz2;
      b0 += info.mu_in * (1. - st.alpha_gpdal) * s_dzz;
    }
    const double INF = __builtin_inf();
    // More breakpoints than threads (2 n_c > NT: the diagonal-structure configurations C5 / C5box, constraint-heavy
    // shapes): the all-breakpoints evaluation below would run its serial loop twice in every lane.  phi' is monotone,
    // and only its values at the two breakpoints around its zero are used: locate the zero with a few workgroup-parallel
    // evaluations, then evaluate -- in the reference's order, bit for bit as below -- only the handful of breakpoints
    // around it.  Any doubt (an evaluation too close to zero to trust its sign, too many breakpoints left in the bracket,
    // an exact value that contradicts the bracket) falls back to the full evaluation.
    // (the kernels that serve such shapes; in the kernel of the common signature -- C2, 200 breakpoints on 256 threads -- the bracket
    // is 6.7 % SLOWER than one pass with every breakpoint on its own thread: profiles/r03_ab_linesearch_bracket.txt)
    if constexpr (PQP_LS_BRACKET && SPEC != 1 && NT == 256)
    if (2 * nc > NT && nc <= NT) { // (its per-thread lists hold two breakpoints)
      double alpha_b;
      sub_tic(ST_CYC_LS_EVAL);
      const bool ok = ls_bracket(a0, b0, gpdal ? info.mu_in_inv * s_bmag / st.alpha_gpdal : info.mu_in_inv * s_bmag, alpha_b);
      sub_toc(ST_CYC_LS_EVAL);
      if (ok)
        return alpha_b;
    }
    // more breakpoints than two per thread (n_c > NT: only the 1024-thread kernels meet such shapes, above 1024 rows)
    if constexpr (NT == 1024 || PQP_CHUNK_ALL)
      if (PQP_UNLIKELY(nc > NT))
        return ls_all_breakpoints_wide(a0, b0);
    // breakpoints (linesearch.hpp:378-391): every breakpoint gets its own thread and
    // its own phi'(alpha) -- no sort, no sequential walk
    sub_tic(ST_CYC_LS_EVAL);
    double first_pos_alpha = INF;
    double my_alpha[2] = { -1.0, -1.0 };
    double my_grad[2] = { 0.0, 0.0 };
    int cnt = 0;
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      int t = threadIdx.x + rep * NT;
      if (t < 2 * nc) {
        int i = t >> 1;
        double cdx = L.Cdx()[i];
        double al = -1.0;
        if (cdx != 0.) {
          double num = (t & 1) ? L.si()[i] : L.rup()[i];
          al = -num / (cdx + MACHINE_EPS);
        }
        if (al > MACHINE_EPS) {
          double ai, bi;
          ls_ineq_terms(al, ai, bi);
          double gr = (a0 + ai) * al + (b0 + bi);
          my_alpha[rep] = al;
          my_grad[rep] = gr;
          ++cnt;
          if (!(gr < 0) && al < first_pos_alpha)
            first_pos_alpha = al;
        }
      }
    }
    count(ST_N_LS_BREAKPOINTS, cnt);
    sub_toc(ST_CYC_LS_EVAL);
    // smallest breakpoint with a non-negative slope: the scan of :427-468 stops there
    double afp, cntd;
    {
      double sv[1] = { (double)cnt };
      double mv[1] = { -first_pos_alpha };
      R.template mixed<1, 1>(sv, mv);
      cntd = sv[0];
      afp = -mv[0];
    }
    if (cntd == 0.0) { // :405-419
      double ai, bi;
      ls_ineq_terms(0.0, ai, bi);
      return -(b0 + bi) / (a0 + ai);
    }
    double gfp = -INF, aln = 0;
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      if (my_alpha[rep] > 0) {
        if (my_alpha[rep] == afp && !(my_grad[rep] < 0))
          gfp = fmax(gfp, my_grad[rep]);
        if (my_alpha[rep] < afp)
          aln = fmax(aln, my_alpha[rep]); // last breakpoint strictly before it
      }
    }
    {
      double none[1] = { 0.0 };
      double mv[2] = { gfp, aln };
      R.template mixed<0, 2>(none, mv);
      gfp = mv[0];
      aln = mv[1];
    }
    double gln = -INF;
#pragma unroll
    for (int rep = 0; rep < 2; ++rep)
      if (my_alpha[rep] > 0 && my_alpha[rep] == aln)
        gln = fmax(gln, my_grad[rep]);
    gln = R.max(gln);
    if (aln == 0.0) { // :477-495
      double ai, bi;
      ls_ineq_terms(0.0, ai, bi);
      gln = b0 + bi;
    }
    if (!(afp < INF)) { // :496-526
      double ai, bi;
      ls_ineq_terms(2 * aln + 1, ai, bi);
      return -(b0 + bi) / (a0 + ai);
    }
    return fabs(aln - gln * (afp - aln) / (gfp - gln)); // :534-536
  }

  // The all-breakpoints evaluation of primal_dual_ls for n_c > NT: a thread owns the breakpoints t, t + NT, t + 2 NT ...
  // and, instead of keeping (alpha, phi') of each in registers, walks them three times -- phi' (the serial loop over
  // the constraints) is only evaluated for every breakpoint in the first walk, afterwards for the ones that tie with
  // the selected step lengths.  Same expressions, same selections, same result as the two-per-thread form.
  __device__ __forceinline__ double ls_all_breakpoints_wide(double a0, double b0)
  {
    const int nc = d.nc;
    const double INF = __builtin_inf();
    auto alpha_of = [&](int t) -> double {
      const int i = t >> 1;
      const double cdx = L.Cdx()[i];
      double al = -1.0;
      if (cdx != 0.) {
        const double num = (t & 1) ? L.si()[i] : L.rup()[i];
        al = -num / (cdx + MACHINE_EPS);
      }
      return al;
    };
    auto grad_of = [&](double al) -> double {
      double ai, bi;
      ls_ineq_terms(al, ai, bi);
      return (a0 + ai) * al + (b0 + bi);
    };
    double first_pos_alpha = INF;
    int cnt = 0;
    for (int t = threadIdx.x; t < 2 * nc; t += NT) {
      const double al = alpha_of(t);
      if (al > MACHINE_EPS) {
        const double gr = grad_of(al);
        ++cnt;
        if (!(gr < 0) && al < first_pos_alpha)
          first_pos_alpha = al;
      }
    }
    count(ST_N_LS_BREAKPOINTS, cnt);
    double afp, cntd;
    {
      double sv[1] = { (double)cnt };
      double mv[1] = { -first_pos_alpha };
      R.template mixed<1, 1>(sv, mv);
      cntd = sv[0];
      afp = -mv[0];
    }
    if (cntd == 0.0) { // :405-419
      double ai, bi;
      ls_ineq_terms(0.0, ai, bi);
      return -(b0 + bi) / (a0 + ai);
    }
    double gfp = -INF, aln = 0;
    for (int t = threadIdx.x; t < 2 * nc; t += NT) {
      const double al = alpha_of(t);
      if (al > MACHINE_EPS) {
        if (al == afp) {
          const double gr = grad_of(al);
          if (!(gr < 0))
            gfp = fmax(gfp, gr);
        }
        if (al < afp)
          aln = fmax(aln, al); // last breakpoint strictly before it
      }
    }
    {
      double none[1] = { 0.0 };
      double mv[2] = { gfp, aln };
      R.template mixed<0, 2>(none, mv);
      gfp = mv[0];
      aln = mv[1];
    }
    double gln = -INF;
    for (int t = threadIdx.x; t < 2 * nc; t += NT) {
      const double al = alpha_of(t);
      if (al > MACHINE_EPS && al == aln)
        gln = fmax(gln, grad_of(al));
    }
    gln = R.max(gln);
    if (aln == 0.0) { // :477-495
      double ai, bi;
      ls_ineq_terms(0.0, ai, bi);
      gln = b0 + bi;
    }
    if (!(afp < INF)) { // :496-526
      double ai, bi;
      ls_ineq_terms(2 * aln + 1, ai, bi);
      return -(b0 + bi) / (a0 + ai);
    }
    return fabs(aln - gln * (afp - aln) / (gfp - gln)); // :534-536
  }

  // Both infeasibility certificates of one Newton step (reference utils.hpp:269-324 primal,
  // :343-419 dual; like them, mutates ATdy, CTdz, dy, dz and Adx, Cdx, Hdx, dx in place) with ALL
  // their norms and sums in ONE fused reduction -- the two tests are independent, and the reference's
  // early exits and intermediate norms only gate arithmetic whose result is then unused:
  //   * primal: dy = dz = 0  <=>  the unscaled norms nrm_dy = nrm_dz = 0 (the scalings are positive);
  //   * dual: "some constraint breaks first_cond" is a maximum of a per-constraint signed value
  //     against bound = |dx|_inf eps_dual_inf, which is known only after the reduction: the value
  //     is reduced, the comparison made afterwards.
  // The inner stopping criterion of the same step (reference solver.hpp:687-743, the saddle-point
  // error at the updated iterate) rides in the same reduction: `err_in`.  do_cert = false (uniform):
  // only the stopping criterion is evaluated, nothing is mutated.
  __device__ __forceinline__ void saddle_point_and_certificates(bool do_cert, double& err_in, bool& primal_infeasible,
                                                                bool& dual_infeasible)
  {
    const int n = d.n, ne = d.n_eq, ni = d.n_in;
    const double c = ruiz_c;
    const double NEG = -__builtin_inf();
    double lb1 = 0, gdx = 0;                          // sums
    double nrm_dy = 0, nrm_dz = 0, lb2 = 0;           // primal maxima
    double ndx = 0, nadx = 0, nhdx = 0, mviol = NEG;  // dual maxima
    double e1 = 0, e2 = 0, e3 = 0;                    // saddle-point error
    {
      const int nc = d.nc;
      const double zf = (st.merit_function_type == PQP_MERIT_GPDAL) ? st.alpha_gpdal : 1.0;
      for (int i = threadIdx.x; i < nc; i += NT) {
        double up = L.rup()[i], lo = L.si()[i];
        double v = (up > 0 ? up : 0.0) + (lo < 0 ? lo : 0.0) - zf * L.z()[i] * info.mu_in;
        e1 = fmax(e1, fabs(v));
      }
      for (int k = threadIdx.x; k < ne; k += NT)
        e2 = fmax(e2, fabs(L.se()[k]));
      for (int k = threadIdx.x; k < n; k += NT)
        e3 = fmax(e3, fabs(L.dres()[k]));
    }
    if (do_cert) {
      cgptr dx = P.dlt_x();
      for (int k = threadIdx.x; k < n; k += NT) {
        const double sc = dx[k] * c;
        L.ATdy()[k] /= sc;
        L.CTdz()[k] /= sc;
        lb2 = fmax(lb2, fabs(L.ATdy()[k] + L.CTdz()[k]));
        L.Hdx()[k] /= sc;
        nhdx = fmax(nhdx, fabs(L.Hdx()[k]));
        gdx += L.dx()[k] * L.gs()[k];
        L.dx()[k] *= dx[k];
        ndx = fmax(ndx, fabs(L.dx()[k]));
      }
      cgptr de = P.dlt_eq();
      for (int k = threadIdx.x; k < ne; k += NT) {
        lb1 += L.dy()[k] * L.bs()[k];
        L.dy()[k] = L.dy()[k] * de[k] / c;
        nrm_dy = fmax(nrm_dy, fabs(L.dy()[k]));
        L.Adx()[k] /= de[k];
        nadx = fmax(nadx, fabs(L.Adx()[k]));
      }
      cgptr di = P.dlt_in();
      for (int k = threadIdx.x; k < ni; k += NT) {
        const double v = L.dz()[k];
        lb1 += (v > 0 ? v : 0.0) * L.us()[k];
        lb1 -= (v < 0 ? v : 0.0) * L.ls()[k];
        L.dz()[k] = v * di[k] / c;
        nrm_dz = fmax(nrm_dz, fabs(L.dz()[k]));
        const double w = L.Cdx()[k] / di[k];
        L.Cdx()[k] = w;
        // utils.hpp:381-398: two-sided bound -> |w| <= bound; no upper bound -> -w <= bound;
        // no lower bound -> w <= bound
        const double val = (L.us()[k] <= 1.E20 && L.ls()[k] >= -1.E20) ? fabs(w) : ((L.us()[k] > 1.E20) ? -w : w);
        mviol = fmax(mviol, val);
      }
      if (has_box()) {
        cgptr db = P.dlt_box();
        for (int k = threadIdx.x; k < n; k += NT) {
          const double v = L.dz()[ni + k];
          lb1 += (v > 0 ? v : 0.0) * L.ubs()[k];
          lb1 -= (v < 0 ? v : 0.0) * L.lbs()[k];
          L.dz()[ni + k] = db[k] * v / c;
          nrm_dz = fmax(nrm_dz, fabs(L.dz()[ni + k]));
          L.Cdx()[ni + k] /= db[k];
          const double w = L.dx()[k]; // (scaled by this same thread in the first loop)
          const double val =
            (L.ubs()[k] <= 1.E20 && L.lbs()[k] >= -1.E20) ? fabs(w) : ((L.ubs()[k] > 1.E20) ? -w : w);
          mviol = fmax(mviol, val);
        }
      }
    }
    double sv[2] = { lb1, gdx };
    double mv[10] = { nrm_dy, nrm_dz, lb2, ndx, nadx, nhdx, mviol, e1, e2, e3 };
    R.template mixed<2, 10>(sv, mv);
    lb1 = sv[0];
    gdx = sv[1];
    nrm_dy = mv[0];
    nrm_dz = mv[1];
    lb2 = mv[2];
    ndx = mv[3];
    nadx = mv[4];
    nhdx = mv[5];
    mviol = mv[6];
    err_in = fmax(mv[7], fmax(mv[8], mv[9]));
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

  // One linear step.  mode 0: semismooth Newton step (reference solver.hpp:754-869);
  // mode 1: equality-constrained initial guess (helpers.hpp:199-228);
  // mode 2: only install the active set encoded in L.aflags (solver.hpp:1231-1240).
  __device__ __forceinline__ bool linear_step(int mode, double eps)
  {
    const int n = d.n, ne = d.n_eq, ni = d.n_in, nc = d.nc;
    const double zfac = (st.merit_function_type == PQP_MERIT_GPDAL) ? st.alpha_gpdal : 1.0;
    if (mode == 0) {
      for (int i = threadIdx.x; i < nc; i += NT) {
        int up = L.rup()[i] >= 0 ? 1 : 0;
        int lo = L.si()[i] <= 0 ? 2 : 0;
        L.aflags()[i] = up | lo | ((up | lo) ? 4 : 0);
      }
    }
    __syncthreads();
    apply_active_set();
    if (mode == 2)
      return false;
    tic();
    if (mode == 0) {
      // right-hand side (solver.hpp:787-847)
      for (int i = threadIdx.x; i < nc; i += NT)
        L.zfull()[i] = (L.slot_of()[i] >= 0) ? 0.0 : L.z()[i]; // inactive multipliers
      __syncthreads();
      if (ni > 0 && dm()) {
        cgptr cd = P.CTs();
        for (int k = threadIdx.x; k < n; k += NT)
          L.CTzin()[k] = cd[k] * L.zfull()[k];
      } else if (ni > 0) {
        // C^T z over the INACTIVE rows: only rows that have just left the active set carry a
        // nonzero multiplier (the step drives them to zero), so the rows are compacted first --
        // indices behind the active list in L.act(), values in t2 -- and only those rows of C_s
        // are read (typically a handful out of n_in; none at all once the active set settles)
        int listed = 0;
        liptr list = L.iscr();
        for (int base = 0; base < ni; base += NT) {
          const int i = base + threadIdx.x;
          const double zi = (i < ni) ? L.zfull()[i] : 0.0;
          const bool f = zi != 0.0;
          int tot;
          const int rank = block_rank<NT>(f, L.icnt(), tot);
          if (f) {
            list[listed + rank] = i;
            L.t2()[listed + rank] = zi;
          }
          listed += tot;
        }
        __syncthreads();
        bytes((long)listed * n * 8);
        if (listed > 0)
          gemv<NT>(P.Cs(), n, listed, n, L.t2(), L.CTzin(), L.part(), list, 0, nullptr, 0);
        else
          vzero(L.CTzin(), n);
      } else {
        vzero(L.CTzin(), n);
      }
      __syncthreads();
      for (int k = threadIdx.x; k < n; k += NT) {
        double s = L.CTzin()[k];
        if (has_box()) {
          s += L.zfull()[ni + k] * L.isc()[k];
          L.CTzin()[k] = s;
        }
        L.rx()[k] = -L.dres()[k] + s;
      }
      for (int k = threadIdx.x; k < ne; k += NT)
        L.rd()[k] = -L.se()[k];
      for (int i = threadIdx.x; i < nc; i += NT) {
        int s = L.slot_of()[i];
        if (s >= 0) {
          double v = 0;
          if (flag_up(i))
            v = -L.rup()[i] + L.z()[i] * info.mu_in * zfac;
          else if (flag_low(i))
            v = -L.si()[i] + L.z()[i] * info.mu_in * zfac;
          L.rd()[ne + s] = v;
        }
      }
    } else {
      for (int k = threadIdx.x; k < n; k += NT)
        L.rx()[k] = -L.gs()[k];
      for (int k = threadIdx.x; k < ne; k += NT)
        L.rd()[k] = L.bs()[k];
    }
    __syncthreads();
    zero_holes(L.rd());
    toc(ST_CYC_NEWTON_MISC);
    // true: refinement missed max(eps, eps_refact) on a factor that rank-1 sweeps have edited -- the
    // reference then rebuilds its factorisation and repeats solve + refinement once (refactorize,
    // solver.hpp:474-532); the Newton loop does that by re-entering this step with the factor marked stale
    const bool missed = iterative_solve(eps);
    if (mode == 1) {
      vcopy(L.x(), L.dx(), n);
      vcopy(L.y(), L.sd(), ne);
      __syncthreads();
      return false;
    }
    if (PQP_UNLIKELY(missed))
      return true;
    // un-permute: dz_i = solution of its slot, or -z_i when inactive (:860-868);
    // C^T dz = (active part, from the last residual) - (inactive multipliers' part)
    vcopy(L.dy(), L.sd(), ne);
    for (int i = threadIdx.x; i < nc; i += NT) {
      int s = L.slot_of()[i];
      L.dz()[i] = (s >= 0) ? L.sd()[ne + s] : -L.z()[i];
    }
    for (int k = threadIdx.x; k < n; k += NT)
      L.CTdz()[k] -= L.CTzin()[k];
    __syncthreads();
    return false;
  }

  // reference solver.hpp:882-1077
  __device__ __forceinline__ void newton_semi_smooth(double eps_int)
  {
    const int n = d.n, ne = d.n_eq, ni = d.n_in, nc = d.nc;
    bool refactorized = false;
    for (long iter = 0; iter <= st.max_iter_in; ++iter) {
      if (iter == st.max_iter_in) {
        info.iter += st.max_iter_in + 1;
        break;
      }
      count(ST_N_NEWTON);
      if (PQP_UNLIKELY(linear_step(0, eps_int)) && !refactorized) {
        // refinement fallback (solver.hpp:474-532): same iterate, same active set, factor rebuilt from
        // scratch by the step's own call of apply_active_set, solve + refinement repeated -- once
        schur_dirty = true;
        refactorized = true;
        count(ST_N_REFACTORIZE);
        --iter;
        continue;
      }
      refactorized = false;
      tic();
      if (st.merit_function_type == PQP_MERIT_GPDAL) {
        for (int i = threadIdx.x; i < nc; i += NT)
          L.Cdx()[i] += (st.alpha_gpdal - 1.) * info.mu_in * L.dz()[i];
        __syncthreads();
      }
      UD alpha = 1.0;
      // |dw|_inf of the step: the line search's first reduction carries it (solver.hpp:944-951 tests
      // |alpha dw|_inf, and max_k fl(|alpha| |dw_k|) = fl(|alpha| max_k |dw_k|): rounding is monotone)
      double dw_max = 0;
      if (ni > 0 || has_box()) {
        for (int rp = 0; rp < reps(1); ++rp)
          alpha = primal_dual_ls(dw_max);
        // (the tail of the line search still reads rup / si / z / dz, which the update below rewrites)
        __syncthreads();
      } else {
        for (int k = threadIdx.x; k < n; k += NT)
          dw_max = fmax(dw_max, fabs(L.dx()[k]));
        for (int k = threadIdx.x; k < ne; k += NT)
          dw_max = fmax(dw_max, fabs(L.dy()[k]));
        dw_max = R.max(dw_max);
      }
      toc(ST_CYC_LINESEARCH);
      sub_tic(ST_CYC_UPDATE);
      if (fabs(alpha) * dw_max < 1.E-11 && iter > 0) {
        info.iter += iter + 1;
        sub_toc(ST_CYC_UPDATE);
        break;
      }
      for (int k = threadIdx.x; k < n; k += NT) {
        L.x()[k] += alpha * L.dx()[k];
        L.dres()[k] += alpha * (info.rho * L.dx()[k] + L.Hdx()[k] + L.ATdy()[k] + L.CTdz()[k]);
      }
      for (int i = threadIdx.x; i < nc; i += NT) {
        L.rup()[i] += alpha * L.Cdx()[i];
        L.si()[i] += alpha * L.Cdx()[i];
        L.z()[i] += alpha * L.dz()[i];
      }
      for (int k = threadIdx.x; k < ne; k += NT) {
        L.se()[k] += alpha * (L.Adx()[k] - info.mu_eq * L.dy()[k]);
        L.y()[k] += alpha * L.dy()[k];
      }
      __syncthreads();
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
        // non-finite iterate: the reference would spin to max_iter and report
        // MAX_ITER_REACHED (it only asserts on NaN in debug builds, solver.hpp:1838-1840);
        // stop here with the same status instead of occupying the device
        info.iter += iter + 1;
        nonfinite = true;
        break;
      }
    }
  }

  // scale a warm start into the equilibrated space (solver.hpp:1137-1146 etc.)
  __device__ __forceinline__ void scale_warm_start()
  {
    const int n = d.n, ne = d.n_eq, ni = d.n_in;
    cgptr dx = P.dlt_x(), de = P.dlt_eq(), di = P.dlt_in();
    for (int k = threadIdx.x; k < n; k += NT)
      L.x()[k] /= dx[k];
    for (int k = threadIdx.x; k < ne; k += NT)
      L.y()[k] = L.y()[k] / de[k] * ruiz_c;
    for (int k = threadIdx.x; k < ni; k += NT)
      L.z()[k] = L.z()[k] / di[k] * ruiz_c;
    if (has_box()) {
      cgptr db = P.dlt_box();
      for (int k = threadIdx.x; k < n; k += NT)
        L.z()[ni + k] = L.z()[ni + k] / db[k] * ruiz_c;
    }
    __syncthreads();
  }

  // ---- reference solver.hpp:1088-1843 ---------------------------------------
  __device__ __forceinline__ void solve()
  {
    const int n = d.n, ne = d.n_eq, ni = d.n_in, nc = d.nc;
    State W = *P.state();
    set_diag_mode(W);
    info.load(*P.info());
    ruiz_c = W.ruiz_c;
    dual_feasibility_rhs_2 = W.dual_feasibility_rhs_2;
    for (int k = threadIdx.x; k < ST_COUNT; k += NT)
      L.stat()[k] = 0;
    if (threadIdx.x == 0)
      L.stat()[ST_CYC_TOTAL] = -clock64(); // (start time parked in its own counter: no register held)

    // (512- / 1024-thread kernels: the results and the model vectors go to LDS AFTER the factorisation prologue below,
    // whose LDS-tiled Z / G build uses every per-QP vector behind dF / t1 as its staging area)
    constexpr bool LATE_LOADS = ZG_LDS_TILED(NT);
    if constexpr (!LATE_LOADS) {
      // results -> LDS (the warm-start modes read them)
      vload(L.x(), P.x(), n);
      vload(L.y(), P.y(), ne);
      vload(L.z(), P.z(), nc);
    }
    {
      // active_set_up / active_set_low live as long as the QP object in the reference: neither
      // Workspace::cleanup (workspace.hpp:330-377) nor init / update touch them, so the duality-gap
      // terms of the first residual evaluation (utils.hpp:545-559) see the flags the LAST Newton step of
      // the previous solve (or compute_backward) left.  They are parked in the high bits of the
      // persistent slot list (bits 16-17 of act[i]; zero for a new object).
      const PQP_GLOBAL int* ga = P.act();
      for (int i = threadIdx.x; i < nc; i += NT) {
        L.aflags()[i] = act_flags(ga[i]);
        L.slot_of()[i] = -1;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0)
      L.stat()[ST_WALL_TICKS] = -wall_clock64(); // (after the barrier: another lane zeroed this counter above)

    // --- what this call has to do (solver.hpp:1125-1376), decided once so that the
    // heavy phases below have a single call site each
    const int ig = st.initial_guess;
    const bool wswpr = (ig == PQP_WARM_START_WITH_PREVIOUS_RESULT);
    const bool dirty = W.dirty != 0;
    bool do_rescale = dirty && !wswpr;
    bool do_factor;      // setup_factorization
    bool do_scale_ws;    // scale x,y,z into the equilibrated space
    bool do_aset_from_z; // active set from z != 0
    bool do_eq_guess = false;
    bool do_restore = false;
    bool zero_iterate = false; // results.cleanup of a dirty QP: x = y = z = 0 (applied when the vectors are loaded)
    if (dirty) {
      if (ig == PQP_EQUALITY_CONSTRAINED_INITIAL_GUESS || ig == PQP_NO_INITIAL_GUESS) {
        zero_iterate = true;
        if constexpr (!LATE_LOADS) {
          vzero(L.x(), n); // results.cleanup
          vzero(L.y(), ne);
          vzero(L.z(), nc);
        }
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
        // nothing usable in HBM (e.g. update() before any solve): rebuild
        do_factor = true;
        do_aset_from_z = true;
      } else {
        do_factor = false;
        do_aset_from_z = false;
        do_restore = true;
      }
    }
    __syncthreads();
    if (do_rescale) {
      // re-apply the stored equilibration (solver.hpp:1192-1214); u, l unclamped
      tic();
      lptr S = L.rd(); // scratch of ntot doubles: rd, ed, sd, dS, t2 (4 nd + max(n, nd)) are free here
      vload(S, P.delta(), d.ntot);
      __syncthreads();
      const bool rewrite_matrices = W.scaled_valid == 0; // (never after an init / update: see write_scaled)
      write_scaled<NT>(batch, q, S, ruiz_c, false, dm(), rewrite_matrices);
      // H, A, C read; H_s, A_s, A_s^T, C_s, C_s^T written (diagonal structure: the two diagonals)
      if (rewrite_matrices)
        bytes(dm() ? ((long)n * 3 + (long)ni * 3) * 8 : ((long)n * n * 2 + 3L * ne * n + 3L * ni * n) * 8);
      toc(ST_CYC_SCALE);
    }
    if constexpr (!LATE_LOADS) {
      vload(L.gs(), P.gs(), n);
      vload(L.bs(), P.bs(), ne);
      vload(L.us(), P.us(), ni);
      vload(L.ls(), P.ls(), ni);
      if (has_box()) {
        vload(L.ubs(), P.ubs(), n);
        vload(L.lbs(), P.lbs(), n);
        vload(L.isc(), P.is(), n);
      }
      __syncthreads();
      if (do_scale_ws)
        scale_warm_start();
    }
    if (do_factor) {
#ifdef PQP_STATS
      if (threadIdx.x == 0)
        L.stat()[ST_CYC_FACTOR_H] -= clock64();
#endif
      tic();
      for (int rp = 0; rp < reps(6); ++rp)
        factor_primal_block();
#ifdef PQP_STATS
      if (threadIdx.x == 0)
        L.stat()[ST_CYC_FACTOR_H] += clock64();
#endif
      tic();
      n_c = 0;
      n_slots = 0;
      r = ne;
      schur_dirty = true;
    }
    if constexpr (LATE_LOADS) {
      // results -> LDS (the warm-start modes read them) and the equilibrated model vectors
      if (zero_iterate) {
        vzero(L.x(), n);
        vzero(L.y(), ne);
        vzero(L.z(), nc);
      } else {
        vload(L.x(), P.x(), n);
        vload(L.y(), P.y(), ne);
        vload(L.z(), P.z(), nc);
      }
      vload(L.gs(), P.gs(), n);
      vload(L.bs(), P.bs(), ne);
      vload(L.us(), P.us(), ni);
      vload(L.ls(), P.ls(), ni);
      if (has_box()) {
        vload(L.ubs(), P.ubs(), n);
        vload(L.lbs(), P.lbs(), n);
        vload(L.isc(), P.is(), n);
      }
      __syncthreads();
      if (do_scale_ws)
        scale_warm_start();
    }
    if (do_restore) {
      // WARM_START_WITH_PREVIOUS_RESULT on an unchanged model: reuse the block
      // factorisation the previous solve left in HBM (solver.hpp:1173-1187, 1343-1375)
      vload(L.dF(), P.dF(), n);
      vload(L.dS(), P.dS(), d.nd);
      n_c = W.n_c;
      n_slots = W.n_slots;
      r = ne + n_slots;
      __syncthreads();
      {
        const PQP_GLOBAL int* ga = P.act();
        for (int j = threadIdx.x; j < n_slots; j += NT) {
          const int i = act_cid(ga[j]);
          L.act()[j] = (i >= 0) ? i : 0; // a hole keeps a valid row index; no slot_of points at it
          if (i >= 0)
            L.slot_of()[i] = j;
        }
      }
      __syncthreads();
      schur_dirty = !(W.ls_valid && W.mu_eq_fact == info.mu_eq && W.mu_in_fact == info.mu_in);
      schur_incremental = W.ls_edited != 0;
    }
    if (do_aset_from_z || do_eq_guess) {
      if (do_aset_from_z) {
        for (int i = threadIdx.x; i < nc; i += NT)
          L.aflags()[i] = (L.aflags()[i] & 3) | ((L.z()[i] != 0) ? 4 : 0); // only active_inequalities is rewritten (solver.hpp:1231-1238)
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
    // The loop body evaluates the residuals at three places per outer iteration in the
    // reference (top, after the inner loop, before the mu update).  `stage` walks
    // through them so that each residual routine is instantiated once.
    UD primal_feasibility_lhs_new = 0, dual_feasibility_lhs_new = 0;
    UD new_bcl_mu_in = 0, new_bcl_mu_eq = 0, new_bcl_mu_in_inv = 0, new_bcl_mu_eq_inv = 0;
    bool is_primal_feasible = false, is_dual_feasible = false;
    long iter = 0;
    int stage = 0; // 0: top of loop, 1: after the Newton loop, 2: before the mu update
    bool done = (st.max_iter <= 0);
    // The reference re-evaluates both global residuals at the top of every outer
    // iteration although (x, y, z) have not moved since the evaluations that closed the
    // previous one (solver.hpp:1598, 1697 -> :1402, 1414).  Those evaluations would
    // reproduce the same numbers bit for bit, so they are skipped: `gpr_fresh` /
    // `gdr_fresh` say that the cached values (and the LDS vectors se, rup, si / dres
    // they leave behind) still describe the current iterate.
    bool gpr_fresh = false, gdr_fresh = false;
    aty_fresh = false;
    {
      // an all-zero iterate (NO_INITIAL_GUESS: the first residual evaluations) needs no pass over H_s, A_s, C_s
      double mz = 0;
      for (int k = threadIdx.x; k < n; k += NT)
        mz = fmax(mz, fabs(L.x()[k]));
      for (int k = threadIdx.x; k < ne; k += NT)
        mz = fmax(mz, fabs(L.y()[k]));
      for (int k = threadIdx.x; k < nc; k += NT)
        mz = fmax(mz, fabs(L.z()[k]));
      iterate_zero = !dm() && R.max(mz) == 0.0;
    }
    UD pl_cache = 0, dl_cache = 0;
    while (!done) {
      tic();
      // closest-feasible mode: the primal residual norm depends on info.status (utils.hpp:241-248),
      // which the loop changes between evaluations at an unmoved iterate -- never reuse it there
      if (st.primal_infeasibility_solving)
        gpr_fresh = false;
      UD pl = pl_cache, dl = dl_cache;
      const bool want_primal = (stage != 2);
      const bool want_dual_pre = (stage != 1);
      if (want_primal && !gpr_fresh) {
        for (int rp = 0; rp < reps(5); ++rp)
          global_primal_residual(pl, primal_feasibility_eq_rhs_0, primal_feasibility_in_rhs_0,
                                 primal_feasibility_eq_lhs, primal_feasibility_in_lhs);
        pl_cache = pl;
        gpr_fresh = true;
      }
      bool want_dual = want_dual_pre;
      if (stage == 1) {
        primal_feasibility_lhs_new = pl;
        is_primal_feasible =
          primal_feasibility_lhs_new <=
          (scaled_eps + st.eps_rel * fmax(primal_feasibility_eq_rhs_0, primal_feasibility_in_rhs_0));
        info.pri_res = primal_feasibility_lhs_new;
        want_dual = is_primal_feasible;
      }
      if (want_dual && !gdr_fresh) {
        for (int rp = 0; rp < reps(5); ++rp)
          global_dual_residual(dl, dual_feasibility_rhs_0, dual_feasibility_rhs_1, dual_feasibility_rhs_3,
                               rhs_duality_gap, duality_gap);
        dl_cache = dl;
        gdr_fresh = true;
      }
      toc(ST_CYC_GLOBAL_RES);
      const UD rhs_dua_rel =
        st.eps_rel * fmax(fmax(dual_feasibility_rhs_3, dual_feasibility_rhs_0),
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
          // solver.hpp:1469-1510: `verbose` is not only printing -- the reference unscales x, y, z for its report and
          // scales them back, the identity up to rounding only: a verbose run perturbs the iterates in their last
          // bits at every outer iteration (test/src/dense_qp_wrapper.cpp:7178 runs its closest-feasible family that
          // way).  Same round trip here; the line the reference prints goes to the trace (the host prints it).
          trace_line(1.0, double(info.iter_ext + 1), info.pri_res, info.dua_res, info.duality_gap, info.mu_in, info.rho);
          cgptr dx = P.dlt_x(), de = P.dlt_eq(), di = P.dlt_in();
          for (int k = threadIdx.x; k < n; k += NT)
            L.x()[k] = (L.x()[k] * dx[k]) / dx[k];
          for (int k = threadIdx.x; k < ne; k += NT)
            L.y()[k] = (L.y()[k] * de[k] / ruiz_c) / de[k] * ruiz_c;
          for (int k = threadIdx.x; k < ni; k += NT)
            L.z()[k] = (L.z()[k] * di[k] / ruiz_c) / di[k] * ruiz_c;
          if (has_box()) {
            cgptr db = P.dlt_box();
            for (int k = threadIdx.x; k < n; k += NT)
              L.z()[ni + k] = (db[k] * L.z()[ni + k] / ruiz_c) / db[k] * ruiz_c;
          }
          __syncthreads();
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
        vcopy(L.xp(), L.x(), n);
        vcopy(L.yp(), L.y(), ne);
        vcopy(L.zp(), L.z(), nc);
        // shifted inequality residuals (solver.hpp:1523-1559)
        {
          cgptr di = P.dlt_in();
          cgptr db = P.dlt_box();
          for (int i = threadIdx.x; i < nc; i += NT) {
            double sc = (i < ni) ? di[i] : db[i - ni];
            double v = L.rup()[i] * sc + L.z()[i] * info.mu_in;
            if (st.merit_function_type == PQP_MERIT_GPDAL)
              v += (st.alpha_gpdal - 1.) * info.mu_in * L.z()[i];
            double ub = (i < ni) ? L.us()[i] : L.ubs()[i - ni];
            double lb = (i < ni) ? L.ls()[i] : L.lbs()[i - ni];
            L.rup()[i] = v - ub;
            L.si()[i] = v - lb;
          }
        }
        __syncthreads();

        // The ~35 scalars of the outer loop are not touched by the Newton loop: they are parked in
        // LDS across it (thread 0 writes, everybody reads back after the loop's closing barrier)
        // instead of occupying ~70 scalar registers -- spilled to VGPR lanes and reloaded all over
        // the hot loops -- for its whole duration.
        const double eta_in_arg = bcl_eta_in;
#define PQP_PARKED(X)                                                                              \
  X(bcl_eta_ext) X(bcl_eta_in) X(primal_feasibility_eq_rhs_0) X(primal_feasibility_in_rhs_0)       \
  X(dual_feasibility_rhs_0) X(dual_feasibility_rhs_1) X(dual_feasibility_rhs_3)                    \
  X(primal_feasibility_lhs) X(primal_feasibility_eq_lhs) X(primal_feasibility_in_lhs)              \
  X(dual_feasibility_lhs) X(duality_gap) X(rhs_duality_gap) X(scaled_eps) X(pl_cache) X(dl_cache)  \
  X(primal_feasibility_lhs_new) X(dual_feasibility_lhs_new) X(new_bcl_mu_in) X(new_bcl_mu_eq)      \
  X(new_bcl_mu_in_inv) X(new_bcl_mu_eq_inv) X(dual_feasibility_rhs_2) X(info.objValue)             \
  X(info.pri_res) X(info.dua_res) X(info.duality_gap) X(info.minimal_H_eigenvalue_estimate)        \
  X(info.setup_time) X(info.solve_time) X(info.run_time)
        {
          lptr pk = L.park();
          if (threadIdx.x == 0) {
            int k = 0;
#define PQP_PUT(v) pk[k++] = (double)(v);
            PQP_PARKED(PQP_PUT)
#undef PQP_PUT
            pk[k++] = (double)iter;
            pk[k++] = (double)info.iter_ext;
            pk[k++] = (double)info.mu_updates;
            pk[k++] = (double)info.rho_updates;
          }
        }
        newton_semi_smooth(eta_in_arg);
        {
          clptr pk = L.park();
          int k = 0;
#define PQP_GET(v) v = pk[k++];
          PQP_PARKED(PQP_GET)
#undef PQP_GET
          iter = uni((long)pk[k++]);
          info.iter_ext = uni((long)pk[k++]);
          info.mu_updates = uni((long)pk[k++]);
          info.rho_updates = uni((long)pk[k++]);
        }
#undef PQP_PARKED
        iterate_zero = false;
        gpr_fresh = false; // x, y, z moved; the shifted rup / si were consumed
        gdr_fresh = false;
        aty_fresh = false; // (and the Newton loop reused the vectors they were parked in)

        if (PQP_UNLIKELY(nonfinite)) {
          info.status = PQP_MAX_ITER_REACHED;
          break;
        }
        if ((info.status == PQP_PRIMAL_INFEASIBLE && !st.primal_infeasibility_solving) ||
            info.status == PQP_DUAL_INFEASIBLE) {
          vcopy(L.x(), L.dx(), n); // certificates (solver.hpp:1572-1580)
          vcopy(L.y(), L.dy(), ne);
          vcopy(L.z(), L.dz(), nc);
          __syncthreads();
          break;
        }
        if (PQP_UNLIKELY(scaled_eps == st.eps_abs && st.primal_infeasibility_solving &&
                         info.status == PQP_PRIMAL_INFEASIBLE)) {
          // solver.hpp:1581-1595 : || A^T 1 + C^T 1 (+ i_scaled) ||_inf * eps_abs
          lptr ones = L.zfull(); // nc >= n_in; n_eq ones taken from L.sd()
          for (int k = threadIdx.x; k < ni; k += NT)
            ones[k] = 1.0;
          for (int k = threadIdx.x; k < ne; k += NT)
            L.sd()[k] = 1.0;
          vzero(L.t1(), n);
          vzero(L.t2(), n);
          __syncthreads();
          if (ne > 0)
            mv(P.A(), n, ne, n, L.sd(), L.t1());
          if (ni > 0)
            mv(P.C(), n, ni, n, ones, L.t2());
          double m = 0;
          for (int k = threadIdx.x; k < n; k += NT)
            m = fmax(m, fabs(L.t1()[k] + L.t2()[k] + (has_box() ? L.isc()[k] : 0.0)));
          scaled_eps = R.max(m) * st.eps_abs;
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
              gap_ok = fabs(info.duality_gap) <=
                       st.eps_duality_gap_abs + st.eps_duality_gap_rel * rhs_duality_gap;
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
            vcopy(L.y(), L.yp(), ne);
            vcopy(L.z(), L.zp(), nc);
            gdr_fresh = false; // y, z were reset
            aty_fresh = false;
            __syncthreads();
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
      if (primal_feasibility_lhs_new >= primal_feasibility_lhs &&
          dual_feasibility_lhs_new >= dual_feasibility_lhs && info.mu_in <= 1e-5) {
        new_bcl_mu_in = st.cold_reset_mu_in; // cold restart
        new_bcl_mu_eq = st.cold_reset_mu_eq;
        new_bcl_mu_in_inv = st.cold_reset_mu_in_inv;
        new_bcl_mu_eq_inv = st.cold_reset_mu_eq_inv;
      }
      if (info.mu_in != new_bcl_mu_in || info.mu_eq != new_bcl_mu_eq) {
        ++info.mu_updates;
        // mu_update (solver.hpp:128-232): here a diagonal shift of the Schur block,
        // re-factorised lazily by the next linear step
        if (ne + n_c > 0)
          schur_dirty = true;
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
      cgptr dx = P.dlt_x(), de = P.dlt_eq(), di = P.dlt_in();
      for (int k = threadIdx.x; k < n; k += NT)
        L.x()[k] *= dx[k];
      for (int k = threadIdx.x; k < ne; k += NT)
        L.y()[k] = L.y()[k] * de[k] / ruiz_c;
      for (int k = threadIdx.x; k < ni; k += NT)
        L.z()[k] = L.z()[k] * di[k] / ruiz_c;
      if (has_box()) {
        cgptr db = P.dlt_box();
        for (int k = threadIdx.x; k < n; k += NT)
          L.z()[ni + k] = db[k] * L.z()[ni + k] / ruiz_c;
      }
      if (st.primal_infeasibility_solving && info.status == PQP_PRIMAL_INFEASIBLE) {
        for (int k = threadIdx.x; k < ne; k += NT)
          L.se()[k] /= de[k];
        for (int k = threadIdx.x; k < ni; k += NT)
          L.si()[k] /= di[k];
        if (has_box()) {
          cgptr db = P.dlt_box();
          for (int k = threadIdx.x; k < n; k += NT)
            L.si()[ni + k] /= db[k];
        }
      }
    }
    __syncthreads();
    // objective on the unscaled model (solver.hpp:1771-1780)
    {
      double obj = 0;
      cgptr g = P.g();
      if (hess() == PQP_HESSIAN_DENSE) {
        mv(P.H(), n, n, n, L.x(), L.t1());
        bytes((long)n * n * 8);
        for (int k = threadIdx.x; k < n; k += NT)
          obj += 0.5 * L.t1()[k] * L.x()[k] + g[k] * L.x()[k];
      } else {
        cgptr H = P.H();
        for (int k = threadIdx.x; k < n; k += NT)
          obj += 0.5 * L.x()[k] * L.x()[k] * H[(long)k * n + k] + g[k] * L.x()[k];
        bytes((long)n * 8);
      }
      info.objValue = R.sum(obj);
    }
    // write back
    vstore(P.x(), L.x(), n);
    vstore(P.y(), L.y(), ne);
    vstore(P.z(), L.z(), nc);
    vstore(P.se(), L.se(), ne);
    vstore(P.si(), L.si(), nc);
    vstore(P.dS(), L.dS(), d.nd);
    if (batch.hx) { // host-mapped mirrors (see Batch)
      vstore((gptr)(batch.hx + P.lq() * n), L.x(), n);
      vstore((gptr)(batch.hy + P.lq() * ne), L.y(), ne);
      vstore((gptr)(batch.hz + P.lq() * nc), L.z(), nc);
      vstore((gptr)(batch.hse + P.lq() * ne), L.se(), ne);
      vstore((gptr)(batch.hsi + P.lq() * nc), L.si(), nc);
    }
    if (pm())
      vstore(P.dF(), L.dF(), n); // D of P_J, beside its inverse factor in the WL buffer
    {
      // the slots of the Schur factor kept for WARM_START_WITH_PREVIOUS_RESULT: 1 + constraint id of a
      // live slot, 0 for a hole (low 16 bits); the persistent up / low flags of constraint i in bits 16-17
      PQP_GLOBAL int* ga = P.act();
      for (int i = threadIdx.x; i < nc; i += NT)
        ga[i] = act_pack((i < n_slots && slot_live(ne + i)) ? L.act()[i] : -1, L.aflags()[i]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      // Info timings (solver.hpp:1112-1115, 1783-1787): the reference reads a host timer around one QP's
      // solve; here it is the time this QP's workgroup was resident, in microseconds (the batch's launch
      // time is pqp_batch_last_solve_ms)
      L.stat()[ST_WALL_TICKS] += wall_clock64();
      if (st.compute_timings) {
        info.solve_time = (double)L.stat()[ST_WALL_TICKS] * batch.wall_us_per_tick;
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
      L.stat()[ST_N_ACTIVE_FINAL] = n_c;
      L.stat()[ST_CYC_TOTAL] += clock64();
      PQP_GLOBAL long long* gs = P.stats();
      for (int k = 0; k < ST_COUNT; ++k)
        gs[k] = L.stat()[k];
    }
  }

  // ---- The factorisation prologue of solve() on its own (pqp_prologue_kernel): the re-applied equilibration of a
  // dirty re-solve (solver.hpp:1192-1214) and setup_factorization (H_s + rho I = L D L^T, W = L^{-1}, Z, G) for the
  // QPs whose solve will need them, decided exactly as solve() decides (same state, same settings; nothing of the
  // state is written).  The one-wavefront dense kernel (pqp_dwave.hpp) runs behind it and starts from what it leaves
  // in HBM (F, dF, WL, WU, Zr, Zc, G and the equilibrated vectors): the GEMM-shaped, once-per-solve part of a solve
  // keeps its 256-thread form and its own register budget, the iteration gets a kernel of its own.
  __device__ __forceinline__ void prologue()
  {
    const int ne = d.n_eq;
    const State W = *P.state();
    set_diag_mode(W);
    info.load(*P.info());
    ruiz_c = W.ruiz_c;
#ifdef PQP_STATS
    for (int k = threadIdx.x; k < ST_COUNT; k += NT)
      L.stat()[k] = 0;
    __syncthreads();
#endif
    const int ig = st.initial_guess;
    const bool wswpr = (ig == PQP_WARM_START_WITH_PREVIOUS_RESULT);
    const bool dirty = W.dirty != 0;
    const bool do_rescale = dirty && !wswpr;
    bool do_factor = true;
    if (dirty && !wswpr)
      cold_start(info, st); // (rho of the factorisation: results.cleanup / cold_start, solver.hpp:1125-1170)
    if (wswpr && !((!dirty && W.refactorize) || !W.factor_valid))
      do_factor = false;
    if (do_rescale) {
      tic();
      lptr S = L.rd();
      vload(S, P.delta(), d.ntot);
      __syncthreads();
      const bool rewrite_matrices = W.scaled_valid == 0;
      write_scaled<NT>(batch, q, S, ruiz_c, false, dm(), rewrite_matrices);
      if (rewrite_matrices)
        bytes(((long)d.n * d.n * 2 + 3L * ne * d.n + 3L * d.n_in * d.n) * 8);
      toc(ST_CYC_SCALE);
    }
    if (do_factor) {
      tic();
      factor_primal_block();
      toc(ST_CYC_FACTOR_H);
    }
#ifdef PQP_STATS
    // (the statistics of the prologue are handed to the iteration kernel through the QP's slot: it adds its own)
    __syncthreads();
    if (threadIdx.x == 0) {
      PQP_GLOBAL long long* gs = P.stats();
      for (int k = 0; k < ST_COUNT; ++k)
        gs[k] = L.stat()[k];
    }
#endif
  }

  // ---- QPLayer backward: dense::compute_backward + compute_backward_loss_ESG
  // (reference dense/compute_ECJ.hpp:29-132, :134-189), on the state a solve left behind.
  // Same steps as the reference: active sets of the solution on the unscaled model, factorisation
  // from scratch at (rho, mu) = (rho_new, mu_new), one refined KKT solve with right-hand side
  // -dL/d(x, y, z_active) in the equilibrated space, then the seven outer products.  The
  // reference's quirks are kept: the inequality part of the right-hand side is scaled by delta_in
  // once per loop iteration after the entry is written (:100-112, position-indexed), and the
  // inactive entries of dz take loss_derivative at their PERMUTED position (:139-146).
  __device__ __forceinline__ void backward(const BackwardArgs& bw, long slot_in_launch)
  {
    const int n = d.n, ne = d.n_eq, ni = d.n_in;
    const long ntot = (long)n + ne + ni;
    cgptr ld = (cgptr)(bw.ld + slot_in_launch * ntot);
    for (int k = threadIdx.x; k < ST_COUNT; k += NT)
      L.stat()[k] = 0;
    // (fields of the per-QP state are read and written in place: a by-value copy of the struct is
    // a stack object here, and its 64-bit reloads trip a register-alignment check of this compiler)
    {
      const State& Wr = *P.state();
      diag_mode = (SPEC == 0) && d.hessian != PQP_HESSIAN_DENSE && d.n_eq == 0 && Wr.c_diag != 0 &&
                  !(d.n_in > 0 && d.box != 0);
      ruiz_c = Wr.ruiz_c;
    }
    info.load(*P.info());
    const double c = ruiz_c;
    vload(L.x(), P.x(), n);
    vload(L.y(), P.y(), ne);
    vload(L.z(), P.z(), ni);
    cgptr dX = P.dlt_x(), dE = P.dlt_eq(), dI = P.dlt_in();
    for (int k = threadIdx.x; k < n; k += NT)
      L.dx()[k] = P.x()[k] / dX[k]; // x in the equilibrated space
    for (int i = threadIdx.x; i < ni; i += NT) {
      L.aflags()[i] = 0;
      L.slot_of()[i] = -1;
    }
    __syncthreads();
    // active sets at the solution (compute_ECJ.hpp:48-57):  C x + z - u >= 0,  C x + z - l <= 0
    if (ni > 0) {
      if (dm()) {
        cgptr cd = P.CTs();
        for (int i = threadIdx.x; i < ni; i += NT)
          L.Cdx()[i] = cd[i] * L.dx()[i];
        __syncthreads();
      } else {
        mv(P.CTs(), ni, n, ni, L.dx(), L.Cdx());
      }
      cgptr gu = P.u(), gl = P.l();
      for (int i = threadIdx.x; i < ni; i += NT) {
        const double ctz = L.Cdx()[i] / dI[i] + L.z()[i];
        const bool up = (ctz - gu[i]) >= 0., lo = (ctz - gl[i]) <= 0.;
        L.aflags()[i] = (up ? 1 : 0) | (lo ? 2 : 0) | ((up || lo) ? 4 : 0);
      }
      __syncthreads();
    }
    info.rho = bw.rho_new;
    info.mu_eq = bw.mu_new;
    info.mu_in = bw.mu_new;
    // setup_factorization + active_set_change from the empty set (:66-86)
    factor_primal_block<false>(); // (the iterate is in LDS already: no staging area)
    n_c = 0;
    n_slots = 0;
    r = ne;
    schur_dirty = true;
    apply_active_set();
    const int na = n_c;
    // right-hand side (:88-112)
    for (int k = threadIdx.x; k < n; k += NT)
      L.rx()[k] = -ld[k] * (dX[k] * c);
    for (int k = threadIdx.x; k < ne; k += NT)
      L.rd()[k] = -ld[n + k] * dE[k];
    double in_any = 0.0;
    for (int i = threadIdx.x; i < ni; i += NT)
      in_any = fmax(in_any, fabs(ld[n + ne + i]));
    in_any = R.max(in_any);
    for (int i = threadIdx.x; i < ni; i += NT) {
      const int a = L.slot_of()[i];
      if (a >= 0) {
        double v = 0.0;
        if (in_any != 0.0) {
          // written at loop iteration i of the reference, then scaled by delta_in[position a]
          // at iterations i, i+1, ..., n_in-1
          v = -ld[n + ne + i];
          const double s = dI[a];
          for (int t = i; t < ni; ++t)
            v *= s;
        }
        L.rd()[ne + a] = v;
      }
    }
    __syncthreads();
    (void)iterative_solve(bw.eps);
    // compute_backward_loss_ESG (:134-189): unpermute dz, unscale, outer products
    for (int k = threadIdx.x; k < n; k += NT)
      L.ex()[k] = L.dx()[k] * dX[k];
    for (int k = threadIdx.x; k < ne; k += NT)
      L.ed()[k] = L.sd()[k] * dE[k] / c;
    for (int j = threadIdx.x; j < ni; j += NT) {
      const int a = L.slot_of()[j];
      double v;
      if (a >= 0) {
        v = L.sd()[ne + a];
      } else {
        // permuted position of an inactive constraint after active_set_change from the identity
        // map: j + #{active i > j}
        int before = 0;
        for (int t = 0; t < na; ++t)
          before += (L.act()[t] < j) ? 1 : 0;
        v = ld[n + ne + (j + na - before)];
      }
      L.zfull()[j] = v * dI[j] / c;
    }
    __syncthreads();
    {
      const long q0 = q;
      gptr oH = (gptr)(bw.dL_dH + q0 * n * n), og = (gptr)(bw.dL_dg + q0 * n);
      gptr oA = (gptr)(bw.dL_dA + q0 * ne * n), ob = (gptr)(bw.dL_db + q0 * ne);
      gptr oC = (gptr)(bw.dL_dC + q0 * ni * n), ou = (gptr)(bw.dL_du + q0 * ni), ol = (gptr)(bw.dL_dl + q0 * ni);
      clptr dxu = L.ex(), dyu = L.ed(), dzu = L.zfull(), xs = L.x(), ys = L.y(), zs = L.z();
      for (int o = threadIdx.x; o < n * n; o += NT) {
        const int i = o / n, k = o - i * n;
        oH[o] = 0.5 * (dxu[i] * xs[k] + xs[i] * dxu[k]);
      }
      for (int k = threadIdx.x; k < n; k += NT)
        og[k] = dxu[k];
      for (int o = threadIdx.x; o < ne * n; o += NT) {
        const int i = o / n, k = o - i * n;
        oA[o] = dyu[i] * xs[k] + ys[i] * dxu[k];
      }
      for (int k = threadIdx.x; k < ne; k += NT)
        ob[k] = -dyu[k];
      for (int o = threadIdx.x; o < ni * n; o += NT) {
        const int i = o / n, k = o - i * n;
        oC[o] = dzu[i] * xs[k] + zs[i] * dxu[k];
      }
      for (int i = threadIdx.x; i < ni; i += NT) {
        ou[i] = (L.aflags()[i] & 1) ? -dzu[i] : 0.0;
        ol[i] = (L.aflags()[i] & 2) ? -dzu[i] : 0.0;
      }
    }
    {
      // compute_backward rewrites work.active_set_up / active_set_low (compute_ECJ.hpp:48-57); the next
      // forward solve starts from them (see solve())
      PQP_GLOBAL int* ga = P.act();
      for (int i = threadIdx.x; i < ni; i += NT)
        ga[i] = act_pack(act_cid(ga[i]), L.aflags()[i]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      // like the reference, the proximal parameters of `results.info` keep the backward values;
      // the factorisation in HBM no longer belongs to a forward solve
      info.store(*P.info());
      State& Ww = *P.state();
      Ww.factor_valid = 0;
      Ww.ls_valid = 0;
      Ww.dirty = 1;
    }
  }
};

template<int NT>
__device__ __forceinline__ void
backward_body(const Batch& batch, const BackwardArgs& bw, long slot, lptr lds_base)
{
  Solver<NT, 0> S(batch, bw.order ? (long)bw.order[slot] : bw.first + slot, lds_base);
  S.backward(bw, slot);
}

template<int NT, int SPEC>
__device__ __forceinline__ void
solve_body(const Batch& batch, long q, lptr lds_base)
{
  Solver<NT, SPEC> S(batch, q, lds_base);
  S.solve();
}

} // namespace pqp

#endif
