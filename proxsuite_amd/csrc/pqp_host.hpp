// Host-side state of one batch handle and the launcher entry points, shared by the
// translation units of libproxqp_hip.so:
//   pqp_capi.hip      host only: the C-ABI of include/proxqp_hip.h
//   pqp_kernels.hip   the kernels, compiled once per PQP_TU value (one object per kernel
//                     family, built in parallel: every solve kernel is ~350 KB of inlined code)
#ifndef PQP_HOST_HPP
#define PQP_HOST_HPP

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "proxqp_hip.h"
#include "pqp_solver.hpp"

// max(n, n_eq + n_c) a batch may have (pqp_batch_create).  The reference has no size limit (dense/model.hpp:65-68); here the
// persistent slot list packs constraint ids into 16 bits (pqp::act_pack) and the set-up kernel keeps 2 (n + n_eq + n_c)
// doubles of Ruiz scaling in LDS (n + n_eq + n_c below ~10 000); above 1024 rows the 1024-thread kernel walks its
// one-thread-per-row stages in chunks and keeps its vectors in HBM when they outgrow the LDS.  Tested against the oracle
// at 4500 and 7000 constraint rows (tests/test_*_parity.py::test_rows_above_4096).
constexpr int PQP_MAX_ROWS = 8192;
static_assert(PQP_MAX_ROWS + 1 < (1 << 16), "act[] packs constraint ids into 16 bits");

// records `msg` for pqp_last_error() of the calling thread and returns `code`
int pqp_fail(int code, const std::string& msg);
#define fail pqp_fail

#define HIP_TRY(expr)                                                                               \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess)                                                                           \
      return pqp_fail(PQP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));              \
  } while (0)

struct pqp_batch
{
  pqp::Batch dev{};
  int device = 0;
  int n_cu = 256; // compute units of the device (hipDeviceAttributeMultiprocessorCount)
  int nt = 256;
  int backend = PQP_BACKEND_PRIMAL_DUAL_LDLT;
  size_t lds_solve = 0, lds_setup = 0;
  std::vector<pqp_settings> settings;
  std::vector<pqp_settings> settings_uploaded; // what the device holds
  // Settings of a QP at the moment its pending init / update was queued: setup() reads the settings of THAT moment
  // in the reference (helpers.hpp:522-572 resets results by the initial_guess current at the call), so the deferred
  // setup kernel is given this snapshot and not what the user wrote into the settings afterwards.
  std::vector<pqp_settings> cmd_settings;
  std::vector<pqp::Cmd> cmd;
  std::vector<char> is_initialized;
  // State::c_diag of every QP as the last set-up kernel left it (read back in pqp_batch_flush for signatures that can
  // have diagonal structure): a launch whose QPs ALL have the structure runs the dedicated kernel (pqp_launch_solve)
  std::vector<char> c_diag;
  bool settings_dirty = true;
  bool cmd_pending = false;
  pqp_settings* d_settings = nullptr;
  pqp::Cmd* d_cmd = nullptr;
  std::vector<void*> allocs;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // dense launches of the one-wavefront kernel are TWO kernels (factorisation prologue + iteration): ev_mid sits between them
  hipEvent_t ev_mid = nullptr;
  bool prologue_timed = false; // the last launch recorded ev_mid
  float last_prologue_ms = 0.f;
  float last_ms = 0.f;
  // Asynchronous solves (pqp_batch_solve*_async): the launch is enqueued, ev1 recorded, nothing waited for.
  // pqp_batch_wait -- or any other entry of the handle, which waits first -- reads the elapsed time, runs the
  // host-side bookkeeping of the finished solve (learned dispatch order, is_initialized, verbose report, mirror
  // freshness) and clears the flag.
  bool solve_in_flight = false;
  long flight_first = 0, flight_count = 0; // the range in flight ...
  std::vector<int64_t> flight_idx;         // ... or the subset (non-empty)
  // Host-resident results (pqp_batch_enable_host_results): pinned device-mapped mirrors the solve epilogue writes
  // (pqp::Batch::hx ...).  mirror_fresh[q]: the mirror of QP q holds what the device holds (set when a solve of q
  // completes, cleared by everything else that writes x, y, z, se, si or Info of q on the device).
  bool host_results = false;
  std::vector<char> mirror_fresh;
  std::vector<void*> host_allocs;
  double *m_x = nullptr, *m_y = nullptr, *m_z = nullptr, *m_se = nullptr, *m_si = nullptr; // host addresses of the mirrors
  pqp_info* m_info = nullptr;
  // settings.verbose: per-iteration trace of the last launch (pqp::Batch::trace).  The slab holds one slot of
  // PQP_TRACE_RECORDS records per verbose QP of the launch and grows on demand; trace_host is its copy on the host,
  // fetched when the solve is settled (verbose_report) and served by pqp_batch_get_trace.
  double* trace_dev = nullptr;
  int64_t trace_slots = 0; // capacity of the slab, in slots
  int* trace_slot_dev = nullptr;
  std::vector<int> trace_slot;   // [B], -1 = QP not traced by the last launch
  std::vector<double> trace_host;
  hipStream_t stream = nullptr; // launch stream (pqp_batch_set_stream); null = default stream
  hipStream_t owned_stream = nullptr; // pqp_batch_own_stream: a non-blocking stream created for (and destroyed with) the handle
  double* vec_scratch = nullptr; // non-null: per-QP vectors live in HBM (B slices of lds_solve bytes), see pqp_kernels.hip TU 9
  long range_first = 0, range_count = 0;
  long setup_first = 0, setup_count = 0; // QPs with a queued init / update / cleanup command
  const int* subset_order = nullptr;     // pqp_batch_solve_subset: slot of workgroup i (device memory)
  const std::vector<int>* subset_host = nullptr; // ... and the same list on the host, for the duration of the launch call
  // every per-QP device array with its element size and per-QP element count (pqp_batch_copy_qp)
  struct Arr
  {
    char* base;
    size_t bytes_per_qp;
  };
  std::vector<Arr> per_qp;
  // QPLayer backward outputs ([B][...], allocated at the first pqp_batch_backward)
  double *bw_dH = nullptr, *bw_dg = nullptr, *bw_dA = nullptr, *bw_db = nullptr, *bw_dC = nullptr,
         *bw_du = nullptr, *bw_dl = nullptr, *bw_ld = nullptr;
  // Longest-processing-time-first dispatch (OFF by default, pqp_batch_set_schedule): after a
  // whole-batch solve the per-QP device cycle counts order the NEXT whole-batch solve of the same
  // handle, most expensive QP first.  QPs are independent, so the order changes nothing but the
  // tail of the launch.
  bool lpt = false;
  bool order_valid = false;
  int* d_order = nullptr;
};

// the caller's current device is restored when a C-ABI entry returns (PyTorch shares the
// thread's current HIP device with this library)
struct DeviceGuard
{
  int prev = -1;
  bool switched = false;
  bool failed = false; // the switch to `device` did not happen: the entry must not touch device memory
  explicit DeviceGuard(int device)
  {
    if (hipGetDevice(&prev) != hipSuccess)
      failed = true;
    else if (prev != device) {
      switched = hipSetDevice(device) == hipSuccess;
      failed = !switched;
    }
  }
  bool ok() const { return !failed; }
  ~DeviceGuard()
  {
    if (switched)
      (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};
// every C-ABI entry that touches device memory: switch to the handle's device or fail the call
#define PQP_ON_DEVICE(dev)                                                                          \
  DeviceGuard guard_(dev);                                                                          \
  if (!guard_.ok())                                                                                 \
    return pqp_fail(PQP_ERR_HIP, "hipSetDevice(" + std::to_string(dev) + ") failed: the call did not run")

// launchers (pqp_kernels.hip); each picks the instantiation for h->nt / the model signature
int pqp_launch_setup(pqp_batch* h);
int pqp_launch_solve(pqp_batch* h);
int pqp_diag_wave_slots(int dim); // register slots per vector of that kernel for a dimension (1, 2 or 4)
int pqp_diag_dispatch(const pqp_batch* h, bool whole_batch = false); // 1: the launch (or, whole_batch, a launch of every QP) goes to the one-wavefront diagonal kernel (pqp_diag.hpp)
int pqp_dense_wave_dispatch(const pqp_batch* h, long count); // 1: a launch of `count` QPs goes to the one-wavefront dense kernel (pqp_dwave.hpp)
size_t pqp_dense_wave_lds_bytes();
int pqp_launch_backward(pqp_batch* h, const pqp::BackwardArgs& bw, long count);
int pqp_launch_order(pqp_batch* h, long count);
int pqp_launch_pack(pqp_batch* h, long first, long count, double* out, hipStream_t stream);

#endif
