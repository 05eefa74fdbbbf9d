// libproxqp_hip.so -- host side of the C-ABI declared in include/proxqp_hip.h:
// HBM layout of a batch, the host mirror of the reference's settings state machine
// (dense/wrapper.hpp init/update, helpers.hpp:174-189, 678-763) and the launches of
// the two kernels (setup = init/update + Ruiz; solve = qp_solve on every QP).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "pqp_host.hpp"
#include "pqp_diag.hpp"

namespace {

thread_local std::string g_err;

bool
absent(double v)
{
  return std::isnan(v);
}

// reference dense/wrapper.hpp:81-113
int
dense_backend_choice(int backend, int64_t dim, int64_t n_eq, int64_t n_in, bool box)
{
  if (backend != PQP_BACKEND_AUTOMATIC)
    return backend;
  int64_t n_constraints = n_in + (box ? dim : 0);
  double threshold = 1.5, frequence = 0.2, d = double(dim);
  double PrimalDualLDLTCost =
    0.5 * std::pow(double(n_eq) / d, 2) +
    0.17 * (std::pow(double(n_eq) / d, 3) + std::pow(double(n_constraints) / d, 3)) +
    frequence * std::pow(double(n_eq + n_constraints) / d, 2) / d;
  double PrimalLDLTCost = threshold * ((0.5 * double(n_eq) + double(n_constraints)) / d + frequence / d);
  return PrimalDualLDLTCost > PrimalLDLTCost ? PQP_BACKEND_PRIMAL_LDLT : PQP_BACKEND_PRIMAL_DUAL_LDLT;
}

} // namespace

int
pqp_fail(int code, const std::string& msg)
{
  g_err = msg;
  return code;
}

namespace {

template<typename T>
int
dalloc(pqp_batch* h, T** p, size_t count)
{
  void* q = nullptr;
  size_t bytes = (count ? count : 1) * sizeof(T);
  HIP_TRY(hipMalloc(&q, bytes));
  HIP_TRY(hipMemset(q, 0, bytes));
  h->allocs.push_back(q);
  *p = static_cast<T*>(q);
  return PQP_OK;
}

int
upload_settings(pqp_batch* h)
{
  if (!h->settings_dirty)
    return PQP_OK;
  // users write into the host records at any time, so every solve has to look at them; the
  // upload itself is skipped when nothing changed since the last one
  const size_t bytes = h->settings.size() * sizeof(pqp_settings);
  if (h->settings_uploaded.size() == h->settings.size() &&
      std::memcmp(h->settings_uploaded.data(), h->settings.data(), bytes) == 0)
    return PQP_OK;
  HIP_TRY(hipMemcpy(h->d_settings, h->settings.data(), bytes, hipMemcpyHostToDevice));
  h->settings_uploaded = h->settings;
  // the host copy stays "dirty" for ever: users hold raw pointers into it and may
  // write at any time (reference: qp.settings is a public member)
  return PQP_OK;
}

int
check_idx(pqp_batch* h, int64_t idx)
{
  if (!h)
    return fail(PQP_ERR_INVALID_ARGUMENT, "null batch handle");
  if (idx < -1 || idx >= h->dev.B)
    return fail(PQP_ERR_INVALID_ARGUMENT, "QP index out of range");
  return PQP_OK;
}

// an asynchronous solve still in flight is waited for before anything else touches the handle
int
settle(pqp_batch* h)
{
  return (h && h->solve_in_flight) ? pqp_batch_wait(h) : PQP_OK;
}

// the device copy of the results of QP idx (-1: all) is about to change outside a solve: its host mirror is stale
void
mirror_stale(pqp_batch* h, int64_t idx)
{
  if (!h->host_results)
    return;
  if (idx < 0)
    std::fill(h->mirror_fresh.begin(), h->mirror_fresh.end(), char(0));
  else
    h->mirror_fresh[size_t(idx)] = 0;
}

// copies `count` QPs worth of one model array
int
copy_in(double* dst_base, const double* src, int64_t idx, int64_t B, size_t per_qp)
{
  if (!src || per_qp == 0)
    return PQP_OK;
  if (idx < 0) {
    HIP_TRY(hipMemcpy(dst_base, src, per_qp * size_t(B) * sizeof(double), hipMemcpyDefault));
  } else {
    HIP_TRY(hipMemcpy(dst_base + size_t(idx) * per_qp, src, per_qp * sizeof(double), hipMemcpyDefault));
  }
  return PQP_OK;
}

int
copy_out(double* dst, const double* src_base, int64_t idx, int64_t B, size_t per_qp)
{
  if (!dst || per_qp == 0)
    return PQP_OK;
  if (idx < 0) {
    HIP_TRY(hipMemcpy(dst, src_base, per_qp * size_t(B) * sizeof(double), hipMemcpyDefault));
  } else {
    HIP_TRY(hipMemcpy(dst, src_base + size_t(idx) * per_qp, per_qp * sizeof(double), hipMemcpyDefault));
  }
  return PQP_OK;
}

// the scalar half of QP::init / QP::update that lives in `settings`
// (reference dense/wrapper.hpp:375, 754-759; helpers.hpp:174-189, 678-705)
void
host_settings_state_machine(pqp_settings& st, bool is_init, int precond_flag, double rho, double mu_eq,
                            double mu_in, double min_eig)
{
  if (is_init)
    st.compute_preconditioner = precond_flag;
  else
    st.update_preconditioner = precond_flag;
  if (!absent(rho))
    st.default_rho = rho;
  if (!absent(mu_eq))
    st.default_mu_eq = mu_eq;
  if (!absent(mu_in))
    st.default_mu_in = mu_in;
  if (!absent(min_eig))
    st.default_H_eigenvalue_estimate = min_eig;
  // results.info.minimal_H_eigenvalue_estimate always equals
  // settings.default_H_eigenvalue_estimate (results.hpp:175-194, helpers.hpp:181-186)
  st.default_rho += std::fabs(st.default_H_eigenvalue_estimate);
}

int
enqueue_setup(pqp_batch* h, int64_t idx, bool update_call, const double* H, const double* g,
              const double* A, const double* b, const double* C, const double* l, const double* u,
              const double* l_box, const double* u_box, int precond_flag, double rho, double mu_eq,
              double mu_in, double min_eig)
{
  if (int rc = check_idx(h, idx))
    return rc;
  if (int rc = settle(h))
    return rc;
  mirror_stale(h, idx); // (setup resets or rescales the results by the initial guess, helpers.hpp:522-572)
  const pqp::Dims& d = h->dev.d;
  // wrapper.hpp:367-372, 542-546, 736-741, 846-850
  if (!d.box && (l_box || u_box))
    return fail(PQP_ERR_INVALID_ARGUMENT,
                "wrong model setup: the QP object is designed without box constraints, but is "
                "initialized or updated with lower or upper box inequalities.");
  PQP_ON_DEVICE(h->device);
  const int64_t lo = idx < 0 ? 0 : idx, hi = idx < 0 ? h->dev.B : idx + 1;
  // one queued command per QP: run what is pending before stacking another one
  for (int64_t q = lo; q < hi; ++q)
    if (h->cmd[size_t(q)].op != pqp::CMD_NONE) {
      if (int rc = pqp_batch_flush(h))
        return rc;
      break;
    }
  const size_t n = size_t(d.n), ne = size_t(d.n_eq), ni = size_t(d.n_in);
  pqp::Batch& D = h->dev;
  int rc = 0;
  if ((rc = copy_in(D.H, H, idx, D.B, n * n)) || (rc = copy_in(D.g, g, idx, D.B, n)) ||
      (rc = copy_in(D.A, A, idx, D.B, ne * n)) || (rc = copy_in(D.b, b, idx, D.B, ne)) ||
      (rc = copy_in(D.C, C, idx, D.B, ni * n)) || (rc = copy_in(D.l, l, idx, D.B, ni)) ||
      (rc = copy_in(D.u, u, idx, D.B, ni)))
    return rc;
  if (d.box && ((rc = copy_in(D.l_box, l_box, idx, D.B, n)) || (rc = copy_in(D.u_box, u_box, idx, D.B, n))))
    return rc;
  for (int64_t q = lo; q < hi; ++q) {
    pqp_settings& st = h->settings[size_t(q)];
    bool is_init = !update_call || !h->is_initialized[size_t(q)];
    double me = min_eig;
    if (update_call && is_init)
      me = std::numeric_limits<double>::quiet_NaN(); // wrapper.hpp:743-746 does not forward it
    host_settings_state_machine(st, is_init, precond_flag, rho, mu_eq, mu_in, me);
    pqp::Cmd c{};
    c.op = is_init ? pqp::CMD_INIT : pqp::CMD_UPDATE;
    c.preconditioner = precond_flag;
    c.matrices_given = (H || A || C) ? 1 : 0;
    c.rho = rho;
    c.mu_eq = mu_eq;
    c.mu_in = mu_in;
    c.min_eig = me;
    h->cmd[size_t(q)] = c;
    if (h->cmd_settings.size() != h->settings.size())
      h->cmd_settings = h->settings;
    h->cmd_settings[size_t(q)] = st;
    if (is_init) {
      h->is_initialized[size_t(q)] = 1;
    } else {
      // setup() -> work.cleanup() clears is_initialized except on the keep-everything
      // WARM_START_WITH_PREVIOUS_RESULT path (helpers.hpp:522-572, workspace.hpp:372)
      bool ppu = !absent(rho) || !absent(mu_eq) || !absent(mu_in);
      bool keep = st.initial_guess == PQP_WARM_START_WITH_PREVIOUS_RESULT && !c.matrices_given && !ppu;
      if (!keep)
        h->is_initialized[size_t(q)] = 0;
    }
  }
  h->cmd_pending = true;
  return PQP_OK;
}

} // namespace

extern "C" {

const char*
pqp_last_error(void)
{
  return g_err.c_str();
}

int
pqp_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess)
    return 0;
  return n;
}

int
pqp_batch_create(int64_t batch_size, int64_t dim, int64_t n_eq, int64_t n_in, int box_constraints,
                 int hessian_type, int dense_backend, int device, pqp_batch** out)
{
  if (!out)
    return fail(PQP_ERR_INVALID_ARGUMENT, "null output handle");
  *out = nullptr;
  if (dim <= 0) // reference dense/model.hpp:65-68
    return fail(PQP_ERR_INVALID_ARGUMENT,
                "wrong argument size: the dimension wrt the primal variable x should be strictly positive.");
  if (batch_size < 0 || n_eq < 0 || n_in < 0)
    return fail(PQP_ERR_INVALID_ARGUMENT, "negative size");
  if (hessian_type < PQP_HESSIAN_ZERO || hessian_type > PQP_HESSIAN_DIAGONAL)
    return fail(PQP_ERR_INVALID_ARGUMENT, "unknown hessian type");
  int ndev = pqp_device_count();
  if (ndev <= 0)
    return fail(PQP_ERR_NO_DEVICE, "no HIP device: libproxqp_hip has no CPU fallback");
  if (device < 0 || device >= ndev)
    return fail(PQP_ERR_INVALID_ARGUMENT, "device ordinal out of range");
  PQP_ON_DEVICE(device);

  pqp_batch* h = new pqp_batch();
  h->device = device;
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
      h->n_cu = cus;
    // wall_clock64() of the device: constant-rate counter, rate in kHz (100 MHz on CDNA) -> Info timings
    int khz = 0;
    h->dev.wall_us_per_tick =
      (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) ? 1.0e3 / double(khz)
                                                                                                      : 1.0e-2;
  }
  h->backend = dense_backend_choice(dense_backend, dim, n_eq, n_in, box_constraints != 0);
  pqp::Dims& d = h->dev.d;
  d.n = int(dim);
  d.n_eq = int(n_eq);
  d.n_in = int(n_in);
  d.box = box_constraints ? 1 : 0;
  d.nc = int(n_in + (box_constraints ? dim : 0));
  d.nd = d.n_eq + d.nc;
  d.ntot = d.n + d.nd;
  d.hessian = hessian_type;
  d.backend = h->backend;
  d._pad = 0;
  h->dev.B = batch_size;
  const int need = d.nd > d.n ? d.nd : d.n;
  h->nt = need <= 256 ? 256 : (need <= 512 ? 512 : 1024);
  // Above 1024 rows the 1024-thread kernel walks its one-thread-per-row stages in chunks (the reference has no size
  // limit, dense/model.hpp:65-68; linalg/dense/factorize.hpp:360-370 switches to a blocked factorisation there).  What
  // bounds a batch here: the 16-bit constraint ids of the persistent slot list and the LDS of the set-up kernel
  // (2 (n + n_eq + n_c) doubles of Ruiz scaling).
  if (need > PQP_MAX_ROWS || pqp::setup_lds_bytes(d, 1024) > 160 * 1024) {
    delete h;
    return fail(PQP_ERR_UNSUPPORTED, "max(n, n_eq + n_in (+ n)) > " + std::to_string(PQP_MAX_ROWS) +
                                       " (or n + n_eq + n_in (+ n) > ~10000) is not supported by this build");
  }
#if PQP_CHUNK_ALL
  if (const char* e = std::getenv("PQP_TEST_NT_MAX")) { // test hook of the emulator variant: a workgroup narrower than the rows
    const int f = std::atoi(e);
    if ((f == 256 || f == 512) && h->nt > f)
      h->nt = f;
  }
#endif
  if (const char* e = std::getenv("PQP_SCHEDULE"))
    h->lpt = std::string(e) == "lpt";
  h->dev.rep_phase = 0;
  h->dev.rep_count = 1;
  h->dev.trace = nullptr;
  h->dev.trace_slot = nullptr;
  h->dev.trace_cap = 0;
  if (const char* e = std::getenv("PQP_REPEAT_PHASE")) // (only the instrumented build reads them: traffic attribution)
    h->dev.rep_phase = std::atoi(e);
  if (const char* e = std::getenv("PQP_REPEAT_COUNT"))
    h->dev.rep_count = std::atoi(e);
  h->lds_solve = pqp::lds_bytes(d, h->nt);
  h->lds_setup = pqp::setup_lds_bytes(d, h->nt);
  // per-QP vector state beyond the 160 KiB of LDS of one CU (e.g. n = 760 with 837 constraint rows): the
  // solver then runs with 1024 threads on a slice of HBM per workgroup (allocated below)
  // (PQP_FORCE_HBM_VECTORS=1: any shape takes that path -- test hook: small problems through the large-shape kernel)
  const char* force_hbm = std::getenv("PQP_FORCE_HBM_VECTORS");
  const bool vectors_in_hbm = h->lds_solve > 160 * 1024 || (force_hbm && force_hbm[0] == '1');
  if (vectors_in_hbm)
    h->nt = 1024;
  const size_t B = size_t(batch_size), n = size_t(dim), ne = size_t(n_eq), ni = size_t(n_in),
               nc = size_t(d.nc), nd = size_t(d.nd);
  pqp::Batch& D = h->dev;
  int rc = 0;
#define ALLOC(ptr, cnt)                                                                             \
  if ((rc = dalloc(h, &(ptr), (cnt)))) {                                                            \
    pqp_batch_destroy(h);                                                                           \
    return rc;                                                                                      \
  }                                                                                                 \
  if (B > 0 && size_t(cnt) % B == 0 && (void*)(ptr) != (void*)h->d_order && (void*)(ptr) != (void*)h->d_cmd)           \
    h->per_qp.push_back({ reinterpret_cast<char*>(ptr), (size_t(cnt) / B) * sizeof(*(ptr)) });
  ALLOC(D.H, B * n * n)
  ALLOC(D.g, B * n)
  ALLOC(D.A, B * ne * n)
  ALLOC(D.b, B * ne)
  ALLOC(D.C, B * ni * n)
  ALLOC(D.u, B * ni)
  ALLOC(D.l, B * ni)
  ALLOC(D.u_box, B * n)
  ALLOC(D.l_box, B * n)
  ALLOC(D.Hs, B * n * n)
  ALLOC(D.gs, B * n)
  ALLOC(D.As, B * ne * n)
  ALLOC(D.ATs, B * ne * n)
  ALLOC(D.bs, B * ne)
  ALLOC(D.Cs, B * ni * n)
  ALLOC(D.CTs, B * ni * n)
  ALLOC(D.us, B * ni)
  ALLOC(D.ls, B * ni)
  ALLOC(D.ubs, B * n)
  ALLOC(D.lbs, B * n)
  ALLOC(D.is, B * n)
  ALLOC(D.delta, B * size_t(d.ntot))
  ALLOC(D.x, B * n)
  ALLOC(D.y, B * ne)
  ALLOC(D.z, B * nc)
  ALLOC(D.se, B * ne)
  ALLOC(D.si, B * nc)
  ALLOC(D.info, B)
  ALLOC(D.state, B)
  ALLOC(D.F, B * n * n)
  ALLOC(D.WL, B * n * n)
  ALLOC(D.WU, B * n * n)
  ALLOC(D.dF, B * n)
  ALLOC(D.Zr, B * nd * n)
  ALLOC(D.Zc, B * nd * n)
  ALLOC(D.G, B * nd * nd)
  ALLOC(D.WS, B * nd * nd)
  ALLOC(D.LS, B * nd * nd)
  ALLOC(D.dS, B * nd)
  ALLOC(D.act, B * nc)
  ALLOC(D.stats, B * size_t(pqp::ST_COUNT))
  if (vectors_in_hbm) {
    h->lds_solve = pqp::lds_bytes(d, h->nt);
    h->lds_setup = pqp::setup_lds_bytes(d, h->nt);
    if ((rc = dalloc(h, &h->vec_scratch, B * ((h->lds_solve + 7) / 8)))) {
      pqp_batch_destroy(h);
      return rc;
    }
  }
  ALLOC(h->d_order, B)
  ALLOC(h->d_settings, B)
  ALLOC(h->d_cmd, B)
#undef ALLOC
  D.settings = h->d_settings;
  D.cmd = h->d_cmd;

  // host-side defaults: Settings(dense_backend) (settings.hpp:213-315), Results(...)
  // (results.hpp:90-144), Model bounds +-sqrt(DBL_MAX) (model.hpp:70-91), Ruiz delta = 1
  h->settings.resize(B);
  for (auto& s : h->settings)
    pqp_settings_default(&s, h->backend);
  h->cmd.assign(B, pqp::Cmd{});
  h->is_initialized.assign(B, 0);
  h->c_diag.assign(B, 0);
  {
    std::vector<pqp_info> info(B);
    for (auto& i : info)
      pqp_info_default(&i, h->backend);
    if (B)
      HIP_TRY(hipMemcpy(D.info, info.data(), B * sizeof(pqp_info), hipMemcpyHostToDevice));
    std::vector<pqp::State> st(B);
    for (auto& s : st) {
      std::memset(&s, 0, sizeof(s));
      s.ruiz_c = 1.0;
    }
    if (B)
      HIP_TRY(hipMemcpy(D.state, st.data(), B * sizeof(pqp::State), hipMemcpyHostToDevice));
    const double ib = std::sqrt(std::numeric_limits<double>::max());
    std::vector<double> tmp;
    auto fill = [&](double* dst, size_t cnt, double v) -> int {
      if (!cnt)
        return PQP_OK;
      tmp.assign(cnt, v);
      HIP_TRY(hipMemcpy(dst, tmp.data(), cnt * sizeof(double), hipMemcpyHostToDevice));
      return PQP_OK;
    };
    if ((rc = fill(D.u, B * ni, +ib)) || (rc = fill(D.l, B * ni, -ib)) ||
        (rc = fill(D.u_box, B * n, +ib)) || (rc = fill(D.l_box, B * n, -ib)) ||
        (rc = fill(D.delta, B * size_t(d.ntot), 1.0)) || (rc = fill(D.is, B * n, 1.0))) {
      pqp_batch_destroy(h);
      return rc;
    }
  }
  HIP_TRY(hipEventCreate(&h->ev0));
  HIP_TRY(hipEventCreate(&h->ev1));
  HIP_TRY(hipEventCreate(&h->ev_mid));
  *out = h;
  return PQP_OK;
}

void
pqp_batch_destroy(pqp_batch* h)
{
  if (!h)
    return;
  (void)settle(h);
  DeviceGuard guard_(h->device);
  for (void* p : h->host_allocs)
    (void)hipHostFree(p);
  if (h->owned_stream)
    (void)hipStreamDestroy(h->owned_stream);
  for (void* p : h->allocs)
    (void)hipFree(p);
  if (h->trace_dev)
    (void)hipFree(h->trace_dev);
  if (h->trace_slot_dev)
    (void)hipFree(h->trace_slot_dev);
  if (h->ev0)
    (void)hipEventDestroy(h->ev0);
  if (h->ev1)
    (void)hipEventDestroy(h->ev1);
  if (h->ev_mid)
    (void)hipEventDestroy(h->ev_mid);
  delete h;
}

int64_t
pqp_batch_size(const pqp_batch* h)
{
  return h ? h->dev.B : 0;
}

int
pqp_batch_dense_backend(const pqp_batch* h)
{
  return h ? h->backend : PQP_BACKEND_AUTOMATIC;
}

pqp_settings*
pqp_batch_settings(pqp_batch* h, int64_t idx)
{
  if (!h || idx < 0 || idx >= h->dev.B)
    return nullptr;
  return &h->settings[size_t(idx)];
}

int
pqp_batch_init(pqp_batch* h, int64_t idx, const double* H, const double* g, const double* A,
               const double* b, const double* C, const double* l, const double* u,
               const double* l_box, const double* u_box, int compute_preconditioner, double rho,
               double mu_eq, double mu_in, double manual_minimal_H_eigenvalue)
{
  if (h && h->dev.d.box == 0 && (l_box || u_box))
    return fail(PQP_ERR_INVALID_ARGUMENT,
                "wrong model setup: the QP object is designed without box constraints, but is "
                "initialized with lower or upper box inequalities."); // wrapper.hpp:542-546
  return enqueue_setup(h, idx, false, H, g, A, b, C, l, u, l_box, u_box, compute_preconditioner ? 1 : 0,
                       rho, mu_eq, mu_in, manual_minimal_H_eigenvalue);
}

int
pqp_batch_update(pqp_batch* h, int64_t idx, const double* H, const double* g, const double* A,
                 const double* b, const double* C, const double* l, const double* u,
                 const double* l_box, const double* u_box, int update_preconditioner, double rho,
                 double mu_eq, double mu_in, double manual_minimal_H_eigenvalue)
{
  return enqueue_setup(h, idx, true, H, g, A, b, C, l, u, l_box, u_box, update_preconditioner ? 1 : 0,
                       rho, mu_eq, mu_in, manual_minimal_H_eigenvalue);
}

int
pqp_batch_warm_start(pqp_batch* h, int64_t idx, const double* x, const double* y, const double* z)
{
  if (int rc = check_idx(h, idx))
    return rc;
  if (!x && !y && !z) // helpers.hpp:724-725
    return PQP_OK;
  if (int rc = settle(h))
    return rc;
  mirror_stale(h, idx);
  PQP_ON_DEVICE(h->device);
  // the guess must land after any queued cleanup of the results
  if (h->cmd_pending)
    if (int rc = pqp_batch_flush(h))
      return rc;
  const pqp::Dims& d = h->dev.d;
  int rc = 0;
  if ((rc = copy_in(h->dev.x, x, idx, h->dev.B, size_t(d.n))) ||
      (rc = copy_in(h->dev.y, y, idx, h->dev.B, size_t(d.n_eq))) ||
      (rc = copy_in(h->dev.z, z, idx, h->dev.B, size_t(d.nc))))
    return rc;
  const int64_t lo = idx < 0 ? 0 : idx, hi = idx < 0 ? h->dev.B : idx + 1;
  for (int64_t q = lo; q < hi; ++q)
    h->settings[size_t(q)].initial_guess = PQP_WARM_START; // helpers.hpp:727
  return PQP_OK;
}

int
pqp_batch_cleanup(pqp_batch* h, int64_t idx)
{
  if (int rc = check_idx(h, idx))
    return rc;
  if (int rc = settle(h))
    return rc;
  mirror_stale(h, idx);
  if (h->cmd_pending)
    if (int rc = pqp_batch_flush(h))
      return rc;
  const int64_t lo = idx < 0 ? 0 : idx, hi = idx < 0 ? h->dev.B : idx + 1;
  for (int64_t q = lo; q < hi; ++q) {
    pqp::Cmd c{};
    c.op = pqp::CMD_CLEANUP;
    h->cmd[size_t(q)] = c;
    h->is_initialized[size_t(q)] = 0;
  }
  h->cmd_pending = true;
  return pqp_batch_flush(h);
}

// A slot handed to a NEW QP object (the C++ facade recycles the slots of its pools): everything a
// freshly created batch holds for that QP -- Settings(dense_backend), Results, Model (zero matrices,
// bounds +-sqrt(DBL_MAX)), workspace flags, Ruiz delta = 1 (reference dense/wrapper.hpp:140-333: every
// constructor starts from defaults).
int
pqp_batch_reset_qp(pqp_batch* h, int64_t idx)
{
  if (!h || idx < 0 || idx >= h->dev.B)
    return fail(PQP_ERR_INVALID_ARGUMENT, "QP index out of range");
  if (int rc = settle(h))
    return rc;
  mirror_stale(h, idx);
  PQP_ON_DEVICE(h->device);
  if (h->cmd_pending)
    if (int rc = pqp_batch_flush(h))
      return rc;
  const size_t q = size_t(idx);
  for (const auto& a : h->per_qp)
    HIP_TRY(hipMemset(a.base + q * a.bytes_per_qp, 0, a.bytes_per_qp));
  pqp::Batch& D = h->dev;
  const pqp::Dims& d = D.d;
  const size_t n = size_t(d.n), ni = size_t(d.n_in);
  pqp_info info;
  pqp_info_default(&info, h->backend);
  HIP_TRY(hipMemcpy(D.info + q, &info, sizeof(info), hipMemcpyHostToDevice));
  pqp::State st;
  std::memset(&st, 0, sizeof(st));
  st.ruiz_c = 1.0;
  HIP_TRY(hipMemcpy(D.state + q, &st, sizeof(st), hipMemcpyHostToDevice));
  const double ib = std::sqrt(std::numeric_limits<double>::max());
  std::vector<double> tmp;
  auto fill = [&](double* dst, size_t cnt, double v) -> int {
    if (!cnt)
      return PQP_OK;
    tmp.assign(cnt, v);
    HIP_TRY(hipMemcpy(dst, tmp.data(), cnt * sizeof(double), hipMemcpyHostToDevice));
    return PQP_OK;
  };
  int rc = 0;
  if ((rc = fill(D.u + q * ni, ni, +ib)) || (rc = fill(D.l + q * ni, ni, -ib)) ||
      (rc = fill(D.u_box + q * n, n, +ib)) || (rc = fill(D.l_box + q * n, n, -ib)) ||
      (rc = fill(D.delta + q * size_t(d.ntot), size_t(d.ntot), 1.0)) || (rc = fill(D.is + q * n, n, 1.0)))
    return rc;
  pqp_settings_default(&h->settings[q], h->backend);
  h->settings_dirty = true;
  h->settings_uploaded.clear(); // (the slice of d_settings was zeroed above: force a re-upload)
  h->cmd[q] = pqp::Cmd{};
  h->is_initialized[q] = 0;
  h->c_diag[q] = 0;
  h->order_valid = false;
  return PQP_OK;
}

int
pqp_batch_flush(pqp_batch* h)
{
  if (!h)
    return fail(PQP_ERR_INVALID_ARGUMENT, "null batch handle");
  if (!h->cmd_pending || h->dev.B == 0)
    return PQP_OK;
  if (int rc = settle(h))
    return rc;
  PQP_ON_DEVICE(h->device);
  if (int rc = upload_settings(h))
    return rc;
  // only the span of QPs that carry a command is uploaded and launched (a BatchQP filled QP by
  // QP through the facade flushes once per QP: B launches of one workgroup, not B launches of B)
  size_t lo = h->cmd.size(), hi = 0;
  for (size_t q = 0; q < h->cmd.size(); ++q)
    if (h->cmd[q].op != pqp::CMD_NONE) {
      lo = std::min(lo, q);
      hi = q + 1;
    }
  if (lo < hi) {
    HIP_TRY(hipMemcpy(h->d_cmd + lo, h->cmd.data() + lo, (hi - lo) * sizeof(pqp::Cmd), hipMemcpyHostToDevice));
    // init / update run under the settings of the moment they were called
    bool stale = false;
    if (h->cmd_settings.size() == h->settings.size()) {
      std::vector<pqp_settings> snap(h->settings.begin() + long(lo), h->settings.begin() + long(hi));
      for (size_t q = lo; q < hi; ++q)
        if ((h->cmd[q].op == pqp::CMD_INIT || h->cmd[q].op == pqp::CMD_UPDATE) &&
            std::memcmp(&h->cmd_settings[q], &h->settings[q], sizeof(pqp_settings)) != 0) {
          snap[q - lo] = h->cmd_settings[q];
          stale = true;
        }
      if (stale)
        HIP_TRY(hipMemcpy(h->d_settings + lo, snap.data(), (hi - lo) * sizeof(pqp_settings), hipMemcpyHostToDevice));
    }
    h->setup_first = long(lo);
    h->setup_count = long(hi - lo);
    if (int rc = pqp_launch_setup(h))
      return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    {
      const pqp::Dims& dd = h->dev.d;
      if (h->nt == 256 && pqp::diag_structure_signature(dd.hessian, dd.n_eq, dd.n_in, dd.box)) {
        // which QPs have diagonal structure (decided by the kernel that just ran, from the model's data)
        std::vector<pqp::State> stt(hi - lo);
        HIP_TRY(hipMemcpy(stt.data(), h->dev.state + lo, (hi - lo) * sizeof(pqp::State), hipMemcpyDeviceToHost));
        for (size_t q = lo; q < hi; ++q)
          h->c_diag[q] = stt[q - lo].c_diag != 0;
      }
    }
    if (stale) // back to the live settings for the solve
      HIP_TRY(hipMemcpy(h->d_settings + lo, h->settings.data() + lo, (hi - lo) * sizeof(pqp_settings),
                        hipMemcpyHostToDevice));
    for (size_t q = lo; q < hi; ++q)
      h->cmd[q].op = pqp::CMD_NONE;
  }
  h->cmd_pending = false;
  return PQP_OK;
}

int
pqp_batch_set_stream(pqp_batch* h, void* stream)
{
  if (!h)
    return fail(PQP_ERR_INVALID_ARGUMENT, "null batch handle");
  if (int rc = settle(h)) // (a solve in flight stays on the stream it was launched on)
    return rc;
  if (h->owned_stream && static_cast<hipStream_t>(stream) != h->owned_stream) {
    PQP_ON_DEVICE(h->device);
    (void)hipStreamDestroy(h->owned_stream);
    h->owned_stream = nullptr;
  }
  h->stream = static_cast<hipStream_t>(stream);
  return PQP_OK;
}

int
pqp_batch_own_stream(pqp_batch* h)
{
  if (!h)
    return fail(PQP_ERR_INVALID_ARGUMENT, "null batch handle");
  if (h->owned_stream)
    return PQP_OK;
  if (int rc = settle(h))
    return rc;
  PQP_ON_DEVICE(h->device);
  HIP_TRY(hipStreamCreateWithFlags(&h->owned_stream, hipStreamNonBlocking));
  h->stream = h->owned_stream;
  return PQP_OK;
}

int
pqp_batch_set_schedule(pqp_batch* h, int longest_first)
{
  if (!h)
    return fail(PQP_ERR_INVALID_ARGUMENT, "null batch handle");
  h->lpt = longest_first != 0;
  return PQP_OK;
}

int
pqp_batch_solve(pqp_batch* h)
{
  if (!h)
    return fail(PQP_ERR_INVALID_ARGUMENT, "null batch handle");
  return pqp_batch_solve_range(h, 0, h->dev.B);
}

constexpr int PQP_TRACE_RECORDS = 4096; // per verbose QP and launch, the header record included (256 KB)

// settings.verbose: the slab of per-iteration records for the verbose QPs of the launch about to be enqueued
// (pqp::Batch::trace); no verbose QP: the kernel gets a null pointer and the previous trace is dropped
static int
prepare_trace(pqp_batch* h, const int64_t* idx, int64_t first, int64_t count)
{
  h->dev.trace = nullptr;
  h->dev.trace_slot = nullptr;
  h->dev.trace_cap = 0;
  h->trace_host.clear();
  int64_t nv = 0;
  for (int64_t i = 0; i < count; ++i)
    nv += h->settings[size_t(idx ? idx[i] : first + i)].verbose != 0;
  if (nv == 0) {
    h->trace_slot.clear();
    return PQP_OK;
  }
  h->trace_slot.assign(size_t(h->dev.B), -1);
  int slot = 0;
  for (int64_t i = 0; i < count; ++i) {
    const size_t q = size_t(idx ? idx[i] : first + i);
    if (h->settings[q].verbose != 0)
      h->trace_slot[q] = slot++;
  }
  const size_t slot_bytes = size_t(PQP_TRACE_RECORDS) * 8 * sizeof(double);
  if (nv > h->trace_slots) {
    if (h->trace_dev)
      (void)hipFree(h->trace_dev);
    h->trace_dev = nullptr;
    h->trace_slots = 0;
    HIP_TRY(hipMalloc((void**)&h->trace_dev, size_t(nv) * slot_bytes));
    h->trace_slots = nv;
  }
  if (!h->trace_slot_dev)
    HIP_TRY(hipMalloc((void**)&h->trace_slot_dev, size_t(h->dev.B) * sizeof(int)));
  // (a synchronous copy: a verbose solve is a debugging run)
  HIP_TRY(hipMemcpy(h->trace_slot_dev, h->trace_slot.data(), size_t(h->dev.B) * sizeof(int), hipMemcpyHostToDevice));
  HIP_TRY(hipMemsetAsync(h->trace_dev, 0, size_t(nv) * slot_bytes, h->stream)); // (on the launch stream: ordered in front of the kernel)
  h->dev.trace = h->trace_dev;
  h->dev.trace_slot = h->trace_slot_dev;
  h->dev.trace_cap = PQP_TRACE_RECORDS;
  return PQP_OK;
}

// the lines of one traced QP, in the reference's format (dense/solver.hpp:1478-1485 and :1021-1027)
static void
print_trace(const double* slot)
{
  const int64_t n = int64_t(slot[0]);
  for (int64_t k = 1; k <= n; ++k) {
    const double* r = slot + k * 8;
    if (r[0] == 1.0)
      std::printf("\033[1;32m[outer iteration %lld]\033[0m\n| primal residual=%.2e | dual residual=%.2e | duality gap=%.2e | "
                  "mu_in=%.2e | rho=%.2e\n",
                  (long long)r[1], r[2], r[3], r[4], r[5], r[6]);
    else
      std::printf("\033[1;34m[inner iteration %lld]\033[0m\n| inner residual=%.2e | alpha=%.2e\n", (long long)r[1], r[2], r[3]);
  }
  if (slot[1] > 0)
    std::printf("(%lld more iteration lines not recorded: the trace holds %d)\n", (long long)slot[1], PQP_TRACE_RECORDS - 1);
}

// settings.verbose (reference dense/utils.hpp:33-131 header, dense/solver.hpp:1789-1830 statistics): the
// whole solve of a QP runs inside one kernel, which records the per-iteration lines of the reference's report
// (pqp::Batch::trace); the header, those lines and the final statistics block are printed here once the launch has
// finished, one block per verbose QP, in index order.
static int
verbose_report(pqp_batch* h, const int64_t* idx, int64_t first, int64_t count)
{
  bool any = false;
  for (int64_t i = 0; i < count && !any; ++i)
    any = h->settings[size_t(idx ? idx[i] : first + i)].verbose != 0;
  if (!any)
    return PQP_OK;
  if (h->dev.trace) {
    int nv = 0;
    for (int s : h->trace_slot)
      nv = std::max(nv, s + 1);
    h->trace_host.resize(size_t(nv) * PQP_TRACE_RECORDS * 8);
    HIP_TRY(hipMemcpy(h->trace_host.data(), h->trace_dev, h->trace_host.size() * sizeof(double), hipMemcpyDeviceToHost));
  }
  const pqp::Dims& d = h->dev.d;
  static const char* const status_name[] = { "Solved", "Maximum number of iterations reached", "Primal infeasible",
                                             "Solved closest primal feasible", "Dual infeasible", "Solver not run" };
  for (int64_t i = 0; i < count; ++i) {
    const int64_t q = idx ? idx[i] : first + i;
    const pqp_settings& st = h->settings[size_t(q)];
    if (!st.verbose)
      continue;
    pqp_info info;
    HIP_TRY(hipMemcpy(&info, h->dev.info + q, sizeof(info), hipMemcpyDeviceToHost));
    std::printf("-------------------------------------------------------------------------------------------------\n"
                "ProxQP dense backend, batched on MI355X (QP %lld of the batch)\n"
                "problem:  \n          variables n = %d, equality constraints n_eq = %d,\n"
                "          inequality constraints n_in = %d\n"
                "settings: \n          backend = dense,\n          eps_abs = %g eps_rel = %g\n"
                "          eps_prim_inf = %g, eps_dual_inf = %g,\n          rho = %g, mu_eq = %g, mu_in = %g,\n"
                "          max_iter = %lld, max_iter_in = %lld,\n          box constraints: %s, \n"
                "          dense backend: %s, \n          problem type: %s, \n          scaling: %s, \n"
                "          timings: %s, \n",
                (long long)q, d.n, d.n_eq, d.n_in, st.eps_abs, st.eps_rel, st.eps_primal_inf, st.eps_dual_inf, info.rho,
                info.mu_eq, info.mu_in, (long long)st.max_iter, (long long)st.max_iter_in, d.box ? "on" : "off",
                h->backend == PQP_BACKEND_PRIMAL_LDLT ? "PrimalLDLT" : "PrimalDualLDLT",
                d.hessian == PQP_HESSIAN_DENSE
                  ? "Quadratic Program"
                  : (d.hessian == PQP_HESSIAN_ZERO ? "Linear Program" : "Quadratic Program with diagonal Hessian"),
                st.compute_preconditioner ? "on" : "off", st.compute_timings ? "on" : "off");
    if (!h->trace_host.empty() && h->trace_slot[size_t(q)] >= 0)
      print_trace(h->trace_host.data() + size_t(h->trace_slot[size_t(q)]) * PQP_TRACE_RECORDS * 8);
    std::printf("-------------------SOLVER STATISTICS-------------------\n"
                "outer iter:     %lld\ntotal iter:     %lld\nmu updates:     %lld\nrho updates:    %lld\n"
                "objective:      %g\nstatus:         %s\n",
                (long long)info.iter_ext, (long long)info.iter, (long long)info.mu_updates,
                (long long)info.rho_updates, info.objValue,
                (info.status >= 0 && info.status <= 5) ? status_name[info.status] : "?");
    if (st.compute_timings)
      std::printf("run time [\xce\xbcs]:  %g\n", info.solve_time);
    std::printf("--------------------------------------------------------\n");
  }
  std::fflush(stdout);
  return PQP_OK;
}

// Host-side bookkeeping of a finished solve (qp_solve ends with work.is_initialized = true, solver.hpp:1836)
static int
finish_solve(pqp_batch* h)
{
  const int64_t* idx = h->flight_idx.empty() ? nullptr : h->flight_idx.data();
  const int64_t first = h->flight_first, count = h->flight_count;
  for (int64_t i = 0; i < count; ++i) {
    const size_t q = size_t(idx ? idx[i] : first + i);
    h->is_initialized[q] = 1;
    if (h->host_results)
      h->mirror_fresh[q] = 1; // the epilogue of the solve wrote the mirror of this QP
  }
  return verbose_report(h, idx, first, count);
}

// one launch of the solve kernel over a range or a subset; `async`: return once it is enqueued
static int
solve_impl(pqp_batch* h, int64_t first, int64_t count, const int64_t* idx, bool async)
{
  if (!h)
    return fail(PQP_ERR_INVALID_ARGUMENT, "null batch handle");
  if (int rc = settle(h))
    return rc;
  std::vector<int> order;
  if (idx) {
    if (count < 0 || count > h->dev.B)
      return fail(PQP_ERR_INVALID_ARGUMENT, "subset larger than the batch");
    if (count == 0)
      return PQP_OK;
    order.resize(static_cast<size_t>(count));
    std::vector<char> seen(static_cast<size_t>(h->dev.B), 0);
    for (int64_t i = 0; i < count; ++i) {
      if (idx[i] < 0 || idx[i] >= h->dev.B)
        return fail(PQP_ERR_INVALID_ARGUMENT, "QP index out of range");
      if (seen[size_t(idx[i])]) // two workgroups on one QP would race on its x / y / z / state / factors
        return fail(PQP_ERR_INVALID_ARGUMENT, "pqp_batch_solve_subset: a QP index is listed twice");
      seen[size_t(idx[i])] = 1;
      order[size_t(i)] = int(idx[i]);
    }
  } else {
    if (first < 0 || count < 0 || first + count > h->dev.B)
      return fail(PQP_ERR_INVALID_ARGUMENT, "solve range [" + std::to_string(first) + ", " +
                                              std::to_string(first + count) + ") outside the batch of " +
                                              std::to_string(h->dev.B) + " QPs");
    if (count == 0)
      return PQP_OK;
  }
  PQP_ON_DEVICE(h->device);
  if (int rc = pqp_batch_flush(h))
    return rc;
  if (int rc = upload_settings(h))
    return rc;
  if (int rc = prepare_trace(h, idx, first, count))
    return rc;
  int rc = 0;
  if (idx) {
    // the dispatch-order array doubles as the subset list (a learned order is dropped)
    HIP_TRY(hipMemcpy(h->d_order, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice));
    h->order_valid = false;
    h->range_first = 0;
    h->range_count = long(count);
    h->subset_order = h->d_order;
    h->subset_host = &order;
    rc = pqp_launch_solve(h);
    h->subset_order = nullptr;
    h->subset_host = nullptr;
    h->flight_idx.assign(idx, idx + count);
  } else {
    h->range_first = first;
    h->range_count = count;
    rc = pqp_launch_solve(h);
    h->flight_idx.clear();
    if (!rc && h->lpt && first == 0 && count == h->dev.B && count > 1) {
      // feedback for the next whole-batch launch: order by the device cycles this solve took (same stream: ordered
      // behind the solve, in front of the next launch)
      rc = pqp_launch_order(h, long(count));
      h->order_valid = rc == 0;
    }
  }
  if (rc)
    return rc;
  h->flight_first = long(first);
  h->flight_count = long(count);
  h->solve_in_flight = true;
  return async ? PQP_OK : pqp_batch_wait(h);
}

int
pqp_batch_wait(pqp_batch* h)
{
  if (!h)
    return fail(PQP_ERR_INVALID_ARGUMENT, "null batch handle");
  if (!h->solve_in_flight)
    return PQP_OK;
  PQP_ON_DEVICE(h->device);
  h->solve_in_flight = false;
  HIP_TRY(hipEventSynchronize(h->ev1));
  HIP_TRY(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
  h->last_prologue_ms = 0.f;
  if (h->prologue_timed)
    HIP_TRY(hipEventElapsedTime(&h->last_prologue_ms, h->ev0, h->ev_mid));
  return finish_solve(h);
}

int
pqp_batch_solve_range(pqp_batch* h, int64_t first, int64_t count)
{
  return solve_impl(h, first, count, nullptr, false);
}

int
pqp_batch_solve_subset(pqp_batch* h, const int64_t* idx, int64_t count)
{
  if (!h || (count > 0 && !idx))
    return fail(PQP_ERR_INVALID_ARGUMENT, "null argument");
  if (count == 0)
    return PQP_OK;
  return solve_impl(h, 0, count, idx, false);
}

int
pqp_batch_solve_async(pqp_batch* h)
{
  if (!h)
    return fail(PQP_ERR_INVALID_ARGUMENT, "null batch handle");
  return solve_impl(h, 0, h->dev.B, nullptr, true);
}

int
pqp_batch_solve_range_async(pqp_batch* h, int64_t first, int64_t count)
{
  return solve_impl(h, first, count, nullptr, true);
}

int
pqp_batch_solve_subset_async(pqp_batch* h, const int64_t* idx, int64_t count)
{
  if (!h || (count > 0 && !idx))
    return fail(PQP_ERR_INVALID_ARGUMENT, "null argument");
  if (count == 0)
    return PQP_OK;
  return solve_impl(h, 0, count, idx, true);
}

// Host-resident results: pinned, device-mapped mirrors of (x, y, z, se, si, Info) that the solve kernel's epilogue
// writes beside the device arrays (pqp::Batch::hx ...).
int
pqp_batch_enable_host_results(pqp_batch* h, int enable)
{
  if (!h)
    return fail(PQP_ERR_INVALID_ARGUMENT, "null batch handle");
  if (int rc = settle(h))
    return rc;
  if ((enable != 0) == h->host_results)
    return PQP_OK;
  PQP_ON_DEVICE(h->device);
  pqp::Batch& D = h->dev;
  if (!enable) {
    for (void* p : h->host_allocs)
      (void)hipHostFree(p);
    h->host_allocs.clear();
    D.hx = D.hy = D.hz = D.hse = D.hsi = nullptr;
    D.hinfo = nullptr;
    h->m_x = h->m_y = h->m_z = h->m_se = h->m_si = nullptr;
    h->m_info = nullptr;
    h->host_results = false;
    return PQP_OK;
  }
  const pqp::Dims& d = D.d;
  const size_t B = size_t(D.B);
  auto halloc = [&](size_t bytes, void** host, void** dev) -> int {
    void* p = nullptr;
    HIP_TRY(hipHostMalloc(&p, bytes ? bytes : 8, hipHostMallocMapped));
    h->host_allocs.push_back(p);
    std::memset(p, 0, bytes ? bytes : 8);
    void* dp = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&dp, p, 0));
    *host = p;
    *dev = dp;
    return PQP_OK;
  };
  int rc = 0;
  if ((rc = halloc(B * size_t(d.n) * 8, (void**)&h->m_x, (void**)&D.hx)) ||
      (rc = halloc(B * size_t(d.n_eq) * 8, (void**)&h->m_y, (void**)&D.hy)) ||
      (rc = halloc(B * size_t(d.nc) * 8, (void**)&h->m_z, (void**)&D.hz)) ||
      (rc = halloc(B * size_t(d.n_eq) * 8, (void**)&h->m_se, (void**)&D.hse)) ||
      (rc = halloc(B * size_t(d.nc) * 8, (void**)&h->m_si, (void**)&D.hsi)) ||
      (rc = halloc(B * sizeof(pqp_info), (void**)&h->m_info, (void**)&D.hinfo))) {
    for (void* p : h->host_allocs)
      (void)hipHostFree(p);
    h->host_allocs.clear();
    D.hx = D.hy = D.hz = D.hse = D.hsi = nullptr;
    D.hinfo = nullptr;
    return rc;
  }
  h->mirror_fresh.assign(B, 0);
  h->host_results = true;
  return PQP_OK;
}

int
pqp_batch_host_results(pqp_batch* h, const double** x, const double** y, const double** z, const double** se,
                       const double** si, const pqp_info** info)
{
  if (!h)
    return fail(PQP_ERR_INVALID_ARGUMENT, "null batch handle");
  if (!h->host_results)
    return fail(PQP_ERR_INVALID_ARGUMENT, "pqp_batch_enable_host_results has not been called on this batch");
  if (int rc = settle(h))
    return rc;
  if (x)
    *x = h->m_x;
  if (y)
    *y = h->m_y;
  if (z)
    *z = h->m_z;
  if (se)
    *se = h->m_se;
  if (si)
    *si = h->m_si;
  if (info)
    *info = h->m_info;
  return PQP_OK;
}

int
pqp_batch_host_results_fresh(pqp_batch* h, int64_t idx)
{
  if (!h || !h->host_results || idx < -1 || idx >= h->dev.B)
    return 0;
  if (settle(h))
    return 0;
  if (idx >= 0)
    return h->mirror_fresh[size_t(idx)] ? 1 : 0;
  for (char c : h->mirror_fresh)
    if (!c)
      return 0;
  return 1;
}

int
pqp_batch_host_results_fresh_range(pqp_batch* h, int64_t first, int64_t count)
{
  if (!h || !h->host_results || first < 0 || count < 0 || first + count > h->dev.B)
    return 0;
  if (settle(h))
    return 0;
  for (int64_t q = first; q < first + count; ++q)
    if (!h->mirror_fresh[size_t(q)])
      return 0;
  return 1;
}

int
pqp_batch_copy_qp(pqp_batch* dst, int64_t dst_idx, pqp_batch* src, int64_t src_idx)
{
  if (!dst || !src)
    return fail(PQP_ERR_INVALID_ARGUMENT, "null batch handle");
  if (dst_idx < 0 || dst_idx >= dst->dev.B || src_idx < 0 || src_idx >= src->dev.B)
    return fail(PQP_ERR_INVALID_ARGUMENT, "QP index out of range");
  const pqp::Dims &a = dst->dev.d, &b = src->dev.d;
  if (a.n != b.n || a.n_eq != b.n_eq || a.n_in != b.n_in || a.box != b.box || a.hessian != b.hessian ||
      dst->per_qp.size() != src->per_qp.size())
    return fail(PQP_ERR_INVALID_ARGUMENT, "pqp_batch_copy_qp: the two batches hold QPs of different shapes");
  if (int rc = settle(src))
    return rc;
  if (int rc = settle(dst))
    return rc;
  mirror_stale(dst, dst_idx);
  PQP_ON_DEVICE(src->device);
  if (int rc = pqp_batch_flush(src))
    return rc;
  if (int rc = pqp_batch_flush(dst))
    return rc;
  for (size_t k = 0; k < src->per_qp.size(); ++k) {
    const size_t nb = src->per_qp[k].bytes_per_qp;
    if (nb != dst->per_qp[k].bytes_per_qp)
      return fail(PQP_ERR_INVALID_ARGUMENT, "pqp_batch_copy_qp: layout mismatch");
    HIP_TRY(hipMemcpy(dst->per_qp[k].base + size_t(dst_idx) * nb, src->per_qp[k].base + size_t(src_idx) * nb, nb,
                      hipMemcpyDefault));
  }
  dst->settings[size_t(dst_idx)] = src->settings[size_t(src_idx)];
  dst->is_initialized[size_t(dst_idx)] = src->is_initialized[size_t(src_idx)];
  dst->c_diag[size_t(dst_idx)] = src->c_diag[size_t(src_idx)];
  dst->settings_dirty = true;
  dst->settings_uploaded.clear(); // (d_settings was overwritten by the array copy: force a re-upload)
  return PQP_OK;
}

static int
backward_impl(pqp_batch* h, int64_t first, int64_t count, const int64_t* idx, const double* loss_derivatives,
              double eps, double rho_backward, double mu_backward)
{
  if (!h)
    return fail(PQP_ERR_INVALID_ARGUMENT, "null batch handle");
  const pqp::Dims& d = h->dev.d;
  std::vector<int> order;
  if (idx) {
    if (count < 0 || count > h->dev.B)
      return fail(PQP_ERR_INVALID_ARGUMENT, "subset larger than the batch");
    order.resize(size_t(count));
    std::vector<char> seen(size_t(h->dev.B), 0);
    for (int64_t i = 0; i < count; ++i) {
      if (idx[i] < 0 || idx[i] >= h->dev.B)
        return fail(PQP_ERR_INVALID_ARGUMENT, "QP index out of range");
      if (seen[size_t(idx[i])])
        return fail(PQP_ERR_INVALID_ARGUMENT, "pqp_batch_backward_subset: a QP index is listed twice");
      seen[size_t(idx[i])] = 1;
      order[size_t(i)] = int(idx[i]);
    }
  } else if (first < 0 || count < 0 || first + count > h->dev.B)
    return fail(PQP_ERR_INVALID_ARGUMENT, "backward range outside the batch");
  if (h->vec_scratch)
    return fail(PQP_ERR_UNSUPPORTED, "compute_backward is not built for shapes whose per-QP vectors exceed the LDS of a "
                                     "CU (n + constraint rows above ~1100): the forward solve is");
  if (d.box)
    return fail(PQP_ERR_UNSUPPORTED, "compute_backward is defined for QPs without box constraints "
                                     "(reference dense/compute_ECJ.hpp ignores them)");
  if (!loss_derivatives)
    return fail(PQP_ERR_INVALID_ARGUMENT, "loss_derivatives is required");
  if (count == 0)
    return PQP_OK;
  if (int rc = settle(h))
    return rc;
  PQP_ON_DEVICE(h->device);
  if (h->cmd_pending)
    if (int rc = pqp_batch_flush(h))
      return rc;
  const size_t B = size_t(h->dev.B), n = size_t(d.n), ne = size_t(d.n_eq), ni = size_t(d.n_in);
  const size_t ntot = n + ne + ni;
  // the reference throws for a dual infeasible QP (compute_ECJ.hpp:37-45)
  {
    std::vector<pqp_info> info;
    if (idx) {
      std::vector<pqp_info> all(B);
      HIP_TRY(hipMemcpy(all.data(), h->dev.info, B * sizeof(pqp_info), hipMemcpyDeviceToHost));
      for (int64_t i = 0; i < count; ++i)
        info.push_back(all[size_t(idx[i])]);
    } else {
      info.resize(size_t(count));
      HIP_TRY(hipMemcpy(info.data(), h->dev.info + first, size_t(count) * sizeof(pqp_info), hipMemcpyDeviceToHost));
    }
    for (int64_t i = 0; i < count; ++i)
      if (info[size_t(i)].status == PQP_DUAL_INFEASIBLE)
        return fail(PQP_ERR_INVALID_ARGUMENT,
                    "the QP problem is not feasible, so computing the derivatives is not valid in this setting. "
                    "Try enabling infeasible solving if the problem is only primally infeasible.");
  }
  if (!h->bw_dH) {
    int rc = 0;
    if ((rc = dalloc(h, &h->bw_dH, B * n * n)) || (rc = dalloc(h, &h->bw_dg, B * n)) ||
        (rc = dalloc(h, &h->bw_dA, B * ne * n)) || (rc = dalloc(h, &h->bw_db, B * ne)) ||
        (rc = dalloc(h, &h->bw_dC, B * ni * n)) || (rc = dalloc(h, &h->bw_du, B * ni)) ||
        (rc = dalloc(h, &h->bw_dl, B * ni)) || (rc = dalloc(h, &h->bw_ld, B * ntot)))
      return rc;
  }
  HIP_TRY(hipMemcpy(h->bw_ld, loss_derivatives, size_t(count) * ntot * sizeof(double), hipMemcpyDefault));
  if (int rc = upload_settings(h))
    return rc;
  pqp::BackwardArgs bw{};
  bw.ld = h->bw_ld;
  bw.eps = eps;
  bw.rho_new = rho_backward;
  bw.mu_new = mu_backward;
  bw.dL_dH = h->bw_dH;
  bw.dL_dg = h->bw_dg;
  bw.dL_dA = h->bw_dA;
  bw.dL_db = h->bw_db;
  bw.dL_dC = h->bw_dC;
  bw.dL_du = h->bw_du;
  bw.dL_dl = h->bw_dl;
  bw.first = long(first);
  bw.order = nullptr;
  if (idx) {
    // (the dispatch-order array doubles as the subset list: a learned order is dropped)
    HIP_TRY(hipMemcpy(h->d_order, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice));
    h->order_valid = false;
    bw.order = h->d_order;
  }
  int rc = pqp_launch_backward(h, bw, long(count));
  if (rc)
    return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  return PQP_OK;
}

int
pqp_batch_backward_range(pqp_batch* h, int64_t first, int64_t count, const double* loss_derivatives, double eps,
                         double rho_backward, double mu_backward)
{
  return backward_impl(h, first, count, nullptr, loss_derivatives, eps, rho_backward, mu_backward);
}

int
pqp_batch_backward_subset(pqp_batch* h, const int64_t* idx, int64_t count, const double* loss_derivatives, double eps,
                          double rho_backward, double mu_backward)
{
  if (!h || (count > 0 && !idx))
    return fail(PQP_ERR_INVALID_ARGUMENT, "null argument");
  return backward_impl(h, 0, count, idx, loss_derivatives, eps, rho_backward, mu_backward);
}

int
pqp_batch_backward(pqp_batch* h, const double* loss_derivatives, double eps, double rho_backward,
                   double mu_backward)
{
  if (!h)
    return fail(PQP_ERR_INVALID_ARGUMENT, "null batch handle");
  return pqp_batch_backward_range(h, 0, h->dev.B, loss_derivatives, eps, rho_backward, mu_backward);
}

int
pqp_batch_get_backward(pqp_batch* h, int64_t idx, double* dL_dH, double* dL_dg, double* dL_dA, double* dL_db,
                       double* dL_dC, double* dL_du, double* dL_dl)
{
  if (int rc = check_idx(h, idx))
    return rc;
  if (!h->bw_dH)
    return fail(PQP_ERR_INVALID_ARGUMENT, "pqp_batch_backward has not been called on this batch");
  if (int rc = settle(h))
    return rc;
  PQP_ON_DEVICE(h->device);
  const pqp::Dims& d = h->dev.d;
  const size_t n = size_t(d.n), ne = size_t(d.n_eq), ni = size_t(d.n_in);
  const int64_t B = h->dev.B;
  int rc = 0;
  if ((rc = copy_out(dL_dH, h->bw_dH, idx, B, n * n)) || (rc = copy_out(dL_dg, h->bw_dg, idx, B, n)) ||
      (rc = copy_out(dL_dA, h->bw_dA, idx, B, ne * n)) || (rc = copy_out(dL_db, h->bw_db, idx, B, ne)) ||
      (rc = copy_out(dL_dC, h->bw_dC, idx, B, ni * n)) || (rc = copy_out(dL_du, h->bw_du, idx, B, ni)) ||
      (rc = copy_out(dL_dl, h->bw_dl, idx, B, ni)))
    return rc;
  return PQP_OK;
}

int
pqp_batch_get_results(pqp_batch* h, int64_t idx, double* x, double* y, double* z, double* se, double* si,
                      pqp_info* info)
{
  if (int rc = check_idx(h, idx))
    return rc;
  if (int rc = settle(h))
    return rc;
  PQP_ON_DEVICE(h->device);
  if (h->cmd_pending)
    if (int rc = pqp_batch_flush(h))
      return rc;
  const pqp::Dims& d = h->dev.d;
  const pqp::Batch& D = h->dev;
  // the host mirrors serve the call when they hold what the device holds (no device-to-host copy)
  const bool mirrored = pqp_batch_host_results_fresh(h, idx) != 0;
  const double *sx = mirrored ? h->m_x : D.x, *sy = mirrored ? h->m_y : D.y, *sz = mirrored ? h->m_z : D.z,
               *sse = mirrored ? h->m_se : D.se, *ssi = mirrored ? h->m_si : D.si;
  const pqp_info* sinfo = mirrored ? h->m_info : D.info;
  int rc = 0;
  if ((rc = copy_out(x, sx, idx, D.B, size_t(d.n))) || (rc = copy_out(y, sy, idx, D.B, size_t(d.n_eq))) ||
      (rc = copy_out(z, sz, idx, D.B, size_t(d.nc))) || (rc = copy_out(se, sse, idx, D.B, size_t(d.n_eq))) ||
      (rc = copy_out(si, ssi, idx, D.B, size_t(d.nc))))
    return rc;
  if (info) {
    if (idx < 0)
      HIP_TRY(hipMemcpy(info, sinfo, size_t(D.B) * sizeof(pqp_info), hipMemcpyDefault));
    else
      HIP_TRY(hipMemcpy(info, sinfo + idx, sizeof(pqp_info), hipMemcpyDefault));
  }
  return PQP_OK;
}

int
pqp_batch_result_device_ptrs(pqp_batch* h, double** x, double** y, double** z)
{
  if (!h)
    return fail(PQP_ERR_INVALID_ARGUMENT, "null batch handle");
  if (x)
    *x = h->dev.x;
  if (y)
    *y = h->dev.y;
  if (z)
    *z = h->dev.z;
  return PQP_OK;
}

int
pqp_batch_pack_results(pqp_batch* h, int64_t first, int64_t count, double* out, void* stream)
{
  if (!h || !out)
    return fail(PQP_ERR_INVALID_ARGUMENT, "null argument");
  if (first < 0 || count < 0 || first + count > h->dev.B)
    return fail(PQP_ERR_INVALID_ARGUMENT, "pack range outside the batch");
  if (count == 0)
    return PQP_OK;
  // (on the launch stream of a solve in flight the pack kernel is ordered behind it: no host wait)
  if (static_cast<hipStream_t>(stream) != h->stream || h->cmd_pending)
    if (int rc = settle(h))
      return rc;
  PQP_ON_DEVICE(h->device);
  if (h->cmd_pending)
    if (int rc = pqp_batch_flush(h))
      return rc;
  return pqp_launch_pack(h, long(first), long(count), out, static_cast<hipStream_t>(stream));
}

int
pqp_batch_get_scaled(pqp_batch* h, int64_t idx, double* H, double* g, double* A, double* b, double* C,
                     double* l, double* u, double* delta, double* c)
{
  if (int rc = check_idx(h, idx))
    return rc;
  if (idx < 0)
    return fail(PQP_ERR_INVALID_ARGUMENT, "pqp_batch_get_scaled addresses one QP");
  if (int rc = settle(h))
    return rc;
  PQP_ON_DEVICE(h->device);
  if (h->cmd_pending)
    if (int rc = pqp_batch_flush(h))
      return rc;
  const pqp::Dims& d = h->dev.d;
  const pqp::Batch& D = h->dev;
  const size_t n = size_t(d.n), ne = size_t(d.n_eq), ni = size_t(d.n_in);
  int rc = 0;
  if ((rc = copy_out(H, D.Hs, idx, D.B, n * n)) || (rc = copy_out(g, D.gs, idx, D.B, n)) ||
      (rc = copy_out(A, D.As, idx, D.B, ne * n)) || (rc = copy_out(b, D.bs, idx, D.B, ne)) ||
      (rc = copy_out(C, D.Cs, idx, D.B, ni * n)) || (rc = copy_out(l, D.ls, idx, D.B, ni)) ||
      (rc = copy_out(u, D.us, idx, D.B, ni)) || (rc = copy_out(delta, D.delta, idx, D.B, size_t(d.ntot))))
    return rc;
  if (c) {
    pqp::State s;
    HIP_TRY(hipMemcpy(&s, D.state + idx, sizeof(s), hipMemcpyDefault));
    *c = s.ruiz_c;
  }
  return PQP_OK;
}

// Diagnostic: the dual Schur block of QP `idx` as the last solve left it in HBM (DenseBackend::PrimalDualLDLT):
// inverse factor W_S (nd x nd, row-major, lower, unit diagonal), D_S (nd), the Gram cache G (nd x nd, by
// constraint id: equality a -> a, inequality i -> n_eq + i), the slot list (nc ints: constraint id of the
// inequality slot j, -1 for a hole) and meta = {n_slots, n_c, ls_valid, ls_edited}, mus = {mu_eq, mu_in} the
// factor was built for.  The identity W_S (M_J + G_JJ) W_S^T = D_S can then be checked on the host
// (tests: accuracy of the rank-1 row appends / deletions on the real device).
int
pqp_batch_get_schur_factor(pqp_batch* h, int64_t idx, double* WS, double* dS, double* G, int32_t* slots,
                           int64_t* meta, double* mus)
{
  if (int rc = check_idx(h, idx))
    return rc;
  if (idx < 0)
    return fail(PQP_ERR_INVALID_ARGUMENT, "pqp_batch_get_schur_factor addresses one QP");
  if (int rc = settle(h))
    return rc;
  PQP_ON_DEVICE(h->device);
  const pqp::Dims& d = h->dev.d;
  const pqp::Batch& D = h->dev;
  const size_t nd = size_t(d.nd), nc = size_t(d.nc);
  int rc = 0;
  if ((rc = copy_out(WS, D.WS, idx, D.B, nd * nd)) || (rc = copy_out(dS, D.dS, idx, D.B, nd)) ||
      (rc = copy_out(G, D.G, idx, D.B, nd * nd)))
    return rc;
  if (slots && nc) {
    std::vector<int> raw(nc);
    HIP_TRY(hipMemcpy(raw.data(), D.act + size_t(idx) * nc, nc * sizeof(int), hipMemcpyDefault));
    for (size_t j = 0; j < nc; ++j)
      slots[j] = pqp::act_cid(raw[j]); // (bits 16-17 carry the persistent up / low flags)
  }
  pqp::State s;
  HIP_TRY(hipMemcpy(&s, D.state + idx, sizeof(s), hipMemcpyDefault));
  if (meta) {
    meta[0] = s.n_slots;
    meta[1] = s.n_c;
    meta[2] = s.ls_valid;
    meta[3] = s.ls_edited;
  }
  if (mus) {
    mus[0] = s.mu_eq_fact;
    mus[1] = s.mu_in_fact;
  }
  return PQP_OK;
}

int
pqp_batch_get_stats(pqp_batch* h, int64_t* stats)
{
  if (!h || !stats)
    return fail(PQP_ERR_INVALID_ARGUMENT, "null argument");
  static_assert(PQP_STATS_COUNT == pqp::ST_COUNT, "stats record size");
  static_assert(sizeof(long long) == sizeof(int64_t), "stats element size");
  if (int rc = settle(h))
    return rc;
  PQP_ON_DEVICE(h->device);
  HIP_TRY(hipMemcpy(stats, h->dev.stats, size_t(h->dev.B) * pqp::ST_COUNT * sizeof(int64_t), hipMemcpyDefault));
  return PQP_OK;
}

int
pqp_batch_get_trace(pqp_batch* h, int64_t idx, double* records, int64_t capacity, int64_t* n_records)
{
  if (!h || !n_records)
    return fail(PQP_ERR_INVALID_ARGUMENT, "null argument");
  if (idx < 0 || idx >= h->dev.B)
    return fail(PQP_ERR_INVALID_ARGUMENT, "QP index out of range");
  if (int rc = settle(h))
    return rc;
  *n_records = 0;
  if (h->trace_host.empty() || h->trace_slot.empty() || h->trace_slot[size_t(idx)] < 0)
    return PQP_OK; // the last launch did not trace this QP (settings.verbose was off)
  const double* slot = h->trace_host.data() + size_t(h->trace_slot[size_t(idx)]) * PQP_TRACE_RECORDS * 8;
  const int64_t n = int64_t(slot[0]);
  *n_records = n;
  if (records && capacity > 0)
    std::memcpy(records, slot + 8, size_t(std::min(n, capacity)) * 8 * sizeof(double));
  return PQP_OK;
}

double
pqp_batch_last_solve_ms(const pqp_batch* h)
{
  if (h && h->solve_in_flight) // the time of a solve in flight is known once it has finished
    (void)pqp_batch_wait(const_cast<pqp_batch*>(h));
  return h ? double(h->last_ms) : 0.0;
}

double
pqp_batch_last_prologue_ms(const pqp_batch* h)
{
  if (h && h->solve_in_flight)
    (void)pqp_batch_wait(const_cast<pqp_batch*>(h));
  return h ? double(h->last_prologue_ms) : 0.0;
}

int
pqp_batch_launch_config(const pqp_batch* h, int* threads, int64_t* lds_bytes)
{
  if (!h)
    return fail(PQP_ERR_INVALID_ARGUMENT, "null batch handle");
  const bool wave = pqp_diag_dispatch(h, true) == 1; // (known once the set-up kernel has run: pqp_batch_flush)
  // (dense QPs: the configuration of a whole-batch launch; a 256-thread factorisation prologue runs in front of the
  // one-wavefront iteration kernel)
  const bool common = h->dev.d.box == 0 && h->dev.d.hessian == PQP_HESSIAN_DENSE && h->dev.d.backend != PQP_BACKEND_PRIMAL_LDLT;
  const bool dwave = !wave && common && pqp_dense_wave_dispatch(h, long(h->dev.B)) == 1;
  if (threads)
    *threads = (wave || dwave) ? 64 : h->nt;
  if (lds_bytes)
    *lds_bytes = wave ? int64_t(pqp::diag_lds_bytes(pqp_diag_wave_slots(h->dev.d.n)))
                      : (dwave ? int64_t(pqp_dense_wave_lds_bytes()) : int64_t(h->lds_solve));
  return PQP_OK;
}

} // extern "C"
