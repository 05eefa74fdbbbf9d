/*
 * proxqp_hip.h -- C-ABI of libproxqp_hip.so: batched dense ProxQP on MI355X (gfx950).
 *
 * This is the drop-in boundary for ONE path of Simple-Robotics/proxsuite: the dense
 * backend's `QP` / `BatchQP` init / update / solve driven by `solve_in_parallel`.
 * ProxSuite has no FFI layer of its own; each entry point below replaces the C++
 * member it cites (paths relative to the reference's include/proxsuite/proxqp/), and
 * the header-only facade in include/proxsuite/ (C++17) plus the Python package
 * proxsuite_amd/ rebuild the reference's public names on top of these calls.
 *
 * Conventions
 *  - all matrices row-major fp64 (reference dense/fwd.hpp:16-19); a batch handle
 *    owns B QPs of identical (n, n_eq, n_in, box, hessian type);
 *  - `idx >= 0` addresses one QP, `idx == -1` the whole batch (arrays then carry a
 *    leading [B] dimension);
 *  - pointers may be host or device pointers (unified addressing);
 *  - NULL array == the reference's `nullopt`; NaN scalar == `nullopt`;
 *  - every function returns 0 on success, a negative pqp_error otherwise and leaves
 *    a message for pqp_last_error(); solver outcomes are NOT errors, they are
 *    reported in pqp_info.status (reference status.hpp:17-26);
 *  - the library fails loudly (PQP_ERR_NO_DEVICE) when no HIP device is present:
 *    there is no CPU fallback.
 */
#ifndef PROXQP_HIP_H
#define PROXQP_HIP_H

#include "pqp_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pqp_batch pqp_batch;

enum pqp_error
{
  PQP_OK = 0,
  PQP_ERR_INVALID_ARGUMENT = -1, /* the reference throws std::invalid_argument */
  PQP_ERR_NO_DEVICE = -2,
  PQP_ERR_HIP = -3,
  PQP_ERR_UNSUPPORTED = -4
};

/* number of entries of the per-QP statistics record, see pqp_batch_get_stats */
#define PQP_STATS_COUNT 33

const char* pqp_last_error(void);
int pqp_device_count(void);

/* dense::BatchQP<T>(batch_size) + B x init_qp_in_place(dim, n_eq, n_in)
 * (reference dense/wrapper.hpp:1263-1283) and the QP constructors (:140-333).
 * `dense_backend` follows dense_backend_choice (:81-113); `device` is the HIP
 * device ordinal. */
int pqp_batch_create(int64_t batch_size, int64_t dim, int64_t n_eq, int64_t n_in,
                     int box_constraints, int hessian_type, int dense_backend, int device,
                     pqp_batch** out);
void pqp_batch_destroy(pqp_batch* h);

int64_t pqp_batch_size(const pqp_batch* h);
int pqp_batch_dense_backend(const pqp_batch* h); /* QP::which_dense_backend() */

/* QP::settings (public member, mutated by users between calls; re-read at every
 * solve: reference benchmark/timings-parallel.cpp:79-88).  Host memory, valid until
 * pqp_batch_destroy. */
pqp_settings* pqp_batch_settings(pqp_batch* h, int64_t idx);

/* QP::init (reference dense/wrapper.hpp:354-498 and, with boxes, :520-703). */
int pqp_batch_init(pqp_batch* h, int64_t idx, const double* H, const double* g, const double* A,
                   const double* b, const double* C, const double* l, const double* u,
                   const double* l_box, const double* u_box, int compute_preconditioner,
                   double rho, double mu_eq, double mu_in, double manual_minimal_H_eigenvalue);

/* QP::update (reference dense/wrapper.hpp:723-807 / :831-918). */
int pqp_batch_update(pqp_batch* h, int64_t idx, const double* H, const double* g, const double* A,
                     const double* b, const double* C, const double* l, const double* u,
                     const double* l_box, const double* u_box, int update_preconditioner,
                     double rho, double mu_eq, double mu_in, double manual_minimal_H_eigenvalue);

/* the warm-start half of QP::solve(x, y, z) (reference dense/wrapper.hpp:940-957,
 * helpers.hpp:715-763): stores the guess and switches initial_guess to WARM_START. */
int pqp_batch_warm_start(pqp_batch* h, int64_t idx, const double* x, const double* y,
                         const double* z);

/* QP::cleanup (reference dense/wrapper.hpp:958-962). */
int pqp_batch_cleanup(pqp_batch* h, int64_t idx);

/* Puts slot `idx` back into the state of a freshly created batch: default Settings of the batch's
 * backend, default Results / Info, zero model with bounds +-sqrt(DBL_MAX), identity equilibration,
 * cleared workspace flags.  What a NEW QP object must see when it takes over a recycled slot
 * (reference dense/wrapper.hpp:140-333: every QP constructor starts from Settings / Results / Model
 * defaults; QP::cleanup, :958-962, keeps settings and model). */
int pqp_batch_reset_qp(pqp_batch* h, int64_t idx);

/* Runs the queued init/update work (Ruiz equilibration, reference helpers.hpp:500-667)
 * on the device.  pqp_batch_solve calls it implicitly; call it explicitly to keep
 * setup out of a timed solve, as the reference bills it to setup_time. */
int pqp_batch_flush(pqp_batch* h);

/* dense::solve_in_parallel(BatchQP&) (reference parallel/qp_solve.hpp:41-59):
 * QP::solve() on every QP of the batch, one workgroup per QP.  Synchronous. */
int pqp_batch_solve(pqp_batch* h);

/* QP::solve() on the QPs [first, first + count) only (reference dense/wrapper.hpp:922-939;
 * `qps.get(i).solve()` in bindings/python/proxsuite/torch/qplayer.py:160-162 is
 * count == 1).  pqp_batch_solve(h) == pqp_batch_solve_range(h, 0, B). */
int pqp_batch_solve_range(pqp_batch* h, int64_t first, int64_t count);

/* dense::solve_in_parallel(std::vector<QP>&) (reference parallel/qp_solve.hpp:17-39): QP::solve() on
 * the `count` QPs idx[0..count) of the batch, in ONE launch (workgroup i solves QP idx[i]). */
int pqp_batch_solve_subset(pqp_batch* h, const int64_t* idx, int64_t count);

/* Asynchronous forms (SURVEY 8(b) threading row: "synchronous by default, async if a stream is given"): the
 * launch is enqueued on the handle's stream (pqp_batch_set_stream; NULL = the null stream) and the call returns
 * without waiting for the device.  pqp_batch_wait blocks until the solve in flight has finished and runs its
 * host-side bookkeeping (device time, verbose report); every other entry of the same handle waits first, so a
 * caller can never observe a half-finished solve.  One solve in flight per handle; several handles -- on one
 * device or on several -- run concurrently: dense::solve_in_parallel over the pools of a BatchQP launches all of
 * them and then waits (reference parallel/qp_solve.hpp:41-59 returns when every QP is solved). */
int pqp_batch_solve_async(pqp_batch* h);
int pqp_batch_solve_range_async(pqp_batch* h, int64_t first, int64_t count);
int pqp_batch_solve_subset_async(pqp_batch* h, const int64_t* idx, int64_t count);
int pqp_batch_wait(pqp_batch* h);

/* Host-resident results.  In the reference `qp.results` are host members when solve / solve_in_parallel return
 * (parallel/qp_solve.hpp:33-37).  With this switch on, the handle owns pinned, device-mapped host mirrors
 * [B][...] of x, y, z, se, si and Info, and the epilogue of the solve kernel writes every QP's results into them
 * beside the device arrays -- each workgroup pushes its own QP over the host link as it finishes, overlapped with
 * the QPs still being solved -- so the results ARE on the host when the solve returns (or pqp_batch_wait does):
 * no device-to-host copy behind the kernel.  pqp_batch_host_results hands out the mirrors (valid until the
 * switch is turned off or the handle destroyed; contents defined for the QPs whose last solve finished after the
 * switch was turned on and whose results no init / update / warm_start / cleanup has touched since:
 * pqp_batch_host_results_fresh(h, idx) == 1, idx == -1: every QP).  pqp_batch_get_results is served from the
 * mirrors whenever they are fresh. */
int pqp_batch_enable_host_results(pqp_batch* h, int enable);
int pqp_batch_host_results(pqp_batch* h, const double** x, const double** y, const double** z, const double** se,
                           const double** si, const pqp_info** info);
int pqp_batch_host_results_fresh(pqp_batch* h, int64_t idx);
/* the same over the QPs [first, first + count): a pool that is only partly filled never solves its free slots, and
 * must not lose the mirror path for the ones it uses */
int pqp_batch_host_results_fresh_range(pqp_batch* h, int64_t first, int64_t count);

/* Copy construction / assignment of a QP (reference dense/wrapper.hpp: QP<T> is copyable and
 * BatchQP / std::vector<QP> rely on it): every per-QP device array, the settings and the
 * initialisation state of QP src_idx of `src` go to QP dst_idx of `dst` (same shape required;
 * the two handles may be the same). */
int pqp_batch_copy_qp(pqp_batch* dst, int64_t dst_idx, pqp_batch* src, int64_t src_idx);

/* HIP stream (hipStream_t, passed as void*) the setup and solve kernels are launched on;
 * NULL (the default) is the null stream.  pqp_batch_solve* stay synchronous with respect to the host; the
 * *_async forms above return once the launch is enqueued on this stream. */
int pqp_batch_set_stream(pqp_batch* h, void* stream);
/* gives the handle a non-blocking stream of its own (created on its device, destroyed with it): handles with their
 * own streams overlap on one device -- what the facade does for every pool, so that the pools of a BatchQP and the
 * standalone QPs of different signatures run concurrently under dense::solve_in_parallel. */
int pqp_batch_own_stream(pqp_batch* h);

/* dense::compute_backward / solve_backward_in_parallel (reference dense/compute_ECJ.hpp:29-189,
 * parallel/qp_solve.hpp:83-137): derivatives of a loss wrt (H, g, A, b, C, u, l) of SOLVED QPs.
 * `loss_derivatives` is [B][dim + n_eq + n_in] = dL/d(x, y, z) per QP (for _range: [count][...],
 * the QPs first .. first+count-1); eps / rho_backward / mu_backward default to 1e-4 / 1e-6 / 1e-6
 * in the reference.  One kernel launch, one workgroup per QP.  Returns
 * PQP_ERR_INVALID_ARGUMENT (the reference throws std::invalid_argument) if a QP of the range is
 * dual infeasible, PQP_ERR_UNSUPPORTED with box constraints. */
int pqp_batch_backward(pqp_batch* h, const double* loss_derivatives, double eps, double rho_backward,
                       double mu_backward);
int pqp_batch_backward_range(pqp_batch* h, int64_t first, int64_t count, const double* loss_derivatives,
                             double eps, double rho_backward, double mu_backward);
/* solve_backward_in_parallel(std::vector<QP>&) (reference parallel/qp_solve.hpp:83-110): compute_backward on the
 * `count` QPs idx[0..count) in ONE launch; row i of loss_derivatives ([count][dim + n_eq + n_in]) belongs to QP idx[i]. */
int pqp_batch_backward_subset(pqp_batch* h, const int64_t* idx, int64_t count, const double* loss_derivatives,
                              double eps, double rho_backward, double mu_backward);
/* Model::backward_data (reference dense/backward_data.hpp:27-133) of QP idx (-1: the whole
 * batch, arrays [B][...]); row-major; any pointer may be NULL. */
int pqp_batch_get_backward(pqp_batch* h, int64_t idx, double* dL_dH, double* dL_dg, double* dL_dA,
                           double* dL_db, double* dL_dC, double* dL_du, double* dL_dl);

/* Dispatch order of whole-batch solves: 1 (default) = longest-processing-time first, using the
 * device cycle counts of the previous whole-batch solve of this handle; 0 = index order.  QPs are
 * independent: the order changes the tail of the launch, never a result.  (Environment
 * PQP_SCHEDULE=fifo sets 0 at creation.) */
int pqp_batch_set_schedule(pqp_batch* h, int longest_first);

/* QP::results (x, y, z, se, si, info); any output pointer may be NULL. */
int pqp_batch_get_results(pqp_batch* h, int64_t idx, double* x, double* y, double* z, double* se,
                          double* si, pqp_info* info);

/* device pointers of the result arrays ([B][n], [B][n_eq], [B][n_c]) for consumers that
 * keep the solution on the GPU (the QPLayer forward). */
int pqp_batch_result_device_ptrs(pqp_batch* h, double** x, double** y, double** z);

/* (x, y, z, status, iter) of the QPs [first, first + count) packed by a device kernel into one
 * row-major [count][dim + n_eq + n_c + 2] fp64 buffer `out` (DEVICE memory), launched on `stream`
 * (hipStream_t; NULL = the null stream) and not synchronised: the payload of the one collective
 * of the sharded path, the final all_gather over RCCL (reference parallel/qp_solve.hpp:41-59 has
 * shared memory instead; proxsuite_amd/sharding.py). */
int pqp_batch_pack_results(pqp_batch* h, int64_t first, int64_t count, double* out, void* stream);

/* scaled model and equilibration of one QP (testing / reference
 * test/src/dense_ruiz_equilibration.cpp) */
int pqp_batch_get_scaled(pqp_batch* h, int64_t idx, double* H, double* g, double* A, double* b,
                         double* C, double* l, double* u, double* delta, double* c);

/* Diagnostic accessor (tests): the dual Schur block of QP `idx` as the last solve left it in HBM --
 * inverse factor W_S [nd*nd], D_S [nd], Gram cache G [nd*nd] (by constraint id), inequality slot list
 * [nc] (constraint index of slot j, -1 = hole), meta = {n_slots, n_c, ls_valid, ls_edited},
 * mus = {mu_eq, mu_in} of the factorisation.  nd = n_eq + n_in (+ dim with box constraints).  The
 * reference keeps the corresponding object in work.ldl (linalg/dense/ldlt.hpp); its row insertions and
 * deletions are linalg/dense/modify.hpp:80-264. */
int pqp_batch_get_schur_factor(pqp_batch* h, int64_t idx, double* WS, double* dS, double* G, int32_t* slots,
                               int64_t* meta, double* mus);

/* per-QP device statistics of the last solve: [B][PQP_STATS_COUNT] int64
 * (cycles per phase and event counters, see proxsuite_amd/csrc/pqp_solver.hpp ST_*) */
int pqp_batch_get_stats(pqp_batch* h, int64_t* stats);

/* settings.verbose: the per-iteration lines the reference prints while it solves (dense/solver.hpp:1478-1485
 * "[outer iteration k] | primal residual= | dual residual= | duality gap= | mu_in= | rho=", and :1021-1027
 * "[inner iteration k] | inner residual= | alpha=").  The whole solve of a QP is one kernel here: the kernel records
 * the lines of every QP whose settings have verbose set, and the library prints them (same text, between the
 * header and the statistics block) when the launch has finished.  This accessor hands out the records of QP `idx`
 * from the last launch, 8 doubles each, in printing order:
 *   { 1, k, pri_res, dua_res, duality_gap, mu_in, rho, 0 }   outer iteration k
 *   { 2, k, inner residual, alpha, 0, 0, 0, 0 }              inner iteration k of the current outer one
 * *n_records = number of lines recorded (0: the QP was not verbose); at most `capacity` are copied (records may
 * be NULL).  4095 lines are kept per QP and launch. */
int pqp_batch_get_trace(pqp_batch* h, int64_t idx, double* records, int64_t capacity, int64_t* n_records);

/* device time of the last solve in milliseconds (HIP events on the launch stream around everything the solve launched) */
double pqp_batch_last_solve_ms(const pqp_batch* h);
/* A launch of dense QPs that fills the device is TWO kernels (the 256-thread factorisation prologue and the one-wavefront
 * iteration kernel behind it, csrc/pqp_dwave.hpp): the part of pqp_batch_last_solve_ms the prologue kernel took, 0 for a
 * launch of one kernel.  (No counterpart in the reference: measurement only.) */
double pqp_batch_last_prologue_ms(const pqp_batch* h);
/* bytes of dynamic LDS and threads per workgroup chosen for this batch */
int pqp_batch_launch_config(const pqp_batch* h, int* threads, int64_t* lds_bytes);

/* ------------------------------------------------------------------------------------------------------------
 * One batch over several GPUs of the node, in ONE process (SURVEY 8(b) `device_mask`, 7 step 7; reference
 * parallel/qp_solve.hpp:41-59: solve_in_parallel uses every core of the host -- here every listed device).
 * A pqp_multi owns G shards: shard g is an ordinary pqp_batch on device devices[g] holding the contiguous range
 * [first_g, first_g + count_g) of the B QPs (count_g = B / G, the first B % G shards one more), with its own HIP
 * stream and host-resident results.  QPs are independent: nothing is exchanged during a solve.  pqp_multi_solve
 * launches every shard (no host synchronisation between the launches) and then waits for all of them; the results
 * are then in the shards' host mirrors and, through pqp_multi_gather_device, in ONE device buffer on a chosen
 * device (pack kernel per shard + peer copies over xGMI -- the in-process form of the final all_gather of
 * proxsuite_amd/sharding.py).  A device may be listed several times (logical shards on one GPU: how the path is
 * tested on a one-GPU box).  `idx` / `first` are indices into the WHOLE batch; idx == -1 addresses all of it,
 * arrays then carry a leading [B] dimension exactly as for pqp_batch_*. */
typedef struct pqp_multi pqp_multi;
int pqp_multi_create(int64_t batch_size, int64_t dim, int64_t n_eq, int64_t n_in, int box_constraints,
                     int hessian_type, int dense_backend, const int* devices, int n_devices, pqp_multi** out);
void pqp_multi_destroy(pqp_multi* m);
int64_t pqp_multi_size(const pqp_multi* m);
int pqp_multi_shard_count(const pqp_multi* m);
/* shard g: its batch handle (owned by `m`), its first QP and its number of QPs; any output may be NULL */
int pqp_multi_shard(pqp_multi* m, int g, pqp_batch** shard, int64_t* first, int64_t* count);
/* shard and local index of QP idx */
int pqp_multi_locate(const pqp_multi* m, int64_t idx, int* shard, int64_t* local);
pqp_settings* pqp_multi_settings(pqp_multi* m, int64_t idx);
int pqp_multi_init(pqp_multi* m, int64_t idx, const double* H, const double* g, const double* A, const double* b,
                   const double* C, const double* l, const double* u, const double* l_box, const double* u_box,
                   int compute_preconditioner, double rho, double mu_eq, double mu_in,
                   double manual_minimal_H_eigenvalue);
int pqp_multi_update(pqp_multi* m, int64_t idx, const double* H, const double* g, const double* A, const double* b,
                     const double* C, const double* l, const double* u, const double* l_box, const double* u_box,
                     int update_preconditioner, double rho, double mu_eq, double mu_in,
                     double manual_minimal_H_eigenvalue);
int pqp_multi_warm_start(pqp_multi* m, int64_t idx, const double* x, const double* y, const double* z);
int pqp_multi_cleanup(pqp_multi* m, int64_t idx);
int pqp_multi_flush(pqp_multi* m);
/* dense::solve_in_parallel over every device; _range: the QPs [first, first + count) of the whole batch */
int pqp_multi_solve(pqp_multi* m);
int pqp_multi_solve_range(pqp_multi* m, int64_t first, int64_t count);
int pqp_multi_solve_async(pqp_multi* m);
int pqp_multi_solve_range_async(pqp_multi* m, int64_t first, int64_t count);
int pqp_multi_wait(pqp_multi* m);
int pqp_multi_get_results(pqp_multi* m, int64_t idx, double* x, double* y, double* z, double* se, double* si,
                          pqp_info* info);
/* (x, y, z, status, iter) of ALL QPs as one row-major [B][dim + n_eq + n_c + 2] fp64 buffer `out` in the memory of
 * the device of shard `root_shard` (allocated there by the caller): pack kernel on every shard's stream, then
 * asynchronous peer copies into place; returns when the buffer is complete. */
int pqp_multi_gather_device(pqp_multi* m, int root_shard, double* out);
/* device time of the last solve: the slowest shard (the job's time), milliseconds */
/* pqp_batch_get_trace of the shard that holds QP idx (settings.verbose: the per-iteration lines of the last launch) */
int pqp_multi_get_trace(pqp_multi* m, int64_t idx, double* records, int64_t capacity, int64_t* n_records);
double pqp_multi_last_solve_ms(const pqp_multi* m);

/* ------------------------------------------------------------------------------------------------------------
 * Diagnostic: what the box delivers right now (no counterpart in the reference; bench.py's `box` record and the
 * performance guard of the GPU tests read timings of the solve kernels against it).  Three fixed kernels, ~60 ms:
 *   out[0] HBM streaming read GB/s (2 GiB, 16 B per lane), out[1] milliseconds of a fixed latency-chain kernel at the
 *   C2 kernel's residency (dependent row reads + wavefront reduction + LDS exchange + barrier), out[2] milliseconds of a
 *   fixed chain of dependent fp64 FMAs (shader-clock proxy), out[3] the shader clock in MHz that implies, out[4] the
 *   number of compute units.  n_out >= 5. */
int pqp_box_calibrate(int device, double* out, int n_out);

#ifdef __cplusplus
}
#endif

#endif /* PROXQP_HIP_H */
