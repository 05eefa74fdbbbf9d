// libproxqp_hip.so -- pqp_multi_*: one batch of QPs spread over several GPUs of the node from ONE process
// (include/proxqp_hip.h; reference parallel/qp_solve.hpp:41-59 -- solve_in_parallel uses every core of the host,
// here every listed device).  Host code only: a pqp_multi is G ordinary batch handles (one per listed device, each
// with its own stream and host-resident results) plus the index arithmetic that maps a QP of the whole batch to
// (shard, local index).  QPs are independent, so a solve is G launches issued back to back without any host
// synchronisation in between, then G waits; the only data that ever crosses devices is the optional final gather
// (pack kernel per shard + peer copies).
#include <hip/hip_runtime.h>

#include <string>
#include <thread>
#include <vector>

#include "pqp_host.hpp"

struct pqp_multi
{
  int64_t B = 0;
  pqp::Dims d{};
  std::vector<pqp_batch*> shard;
  std::vector<int64_t> first, count;
  std::vector<hipStream_t> stream;
  std::vector<double*> pack; // per shard: staging buffer of the gather on the shard's device (lazily allocated)
  std::vector<char> in_flight;
};

namespace {

int
check_multi_idx(const pqp_multi* m, int64_t idx)
{
  if (!m)
    return fail(PQP_ERR_INVALID_ARGUMENT, "null multi-device handle");
  if (idx < -1 || idx >= m->B)
    return fail(PQP_ERR_INVALID_ARGUMENT, "QP index out of range");
  return PQP_OK;
}

// shard of QP idx (the shards hold contiguous ranges: binary search would do; G is at most a handful)
int
shard_of(const pqp_multi* m, int64_t idx)
{
  for (size_t g = 0; g < m->shard.size(); ++g)
    if (idx >= m->first[g] && idx < m->first[g] + m->count[g])
      return int(g);
  return -1;
}

const double*
off(const double* p, int64_t first, size_t per_qp)
{
  return p ? p + size_t(first) * per_qp : nullptr;
}
double*
off(double* p, int64_t first, size_t per_qp)
{
  return p ? p + size_t(first) * per_qp : nullptr;
}

typedef int (*setup_fn)(pqp_batch*, int64_t, const double*, const double*, const double*, const double*, const double*,
                        const double*, const double*, const double*, const double*, int, double, double, double, double);

// Runs `work(s)` for every shard that holds QPs, all of them at once: one host thread per shard (the caller's takes the
// first).  A set-up is a blocking copy of the shard's slice of the model from pageable host memory plus, at flush time,
// the set-up kernel and a read-back of its flags -- walked shard after shard (rounds 3-4) the G devices of a node worked
// one at a time: 8 x (H2D + Ruiz) in front of an 8 ms solve (reference parallel/qp_solve.hpp:41-59 sets
// up on every core at once).  Every entry of the batch C-ABI switches to its handle's device for the calling thread only
// (DeviceGuard) and reports errors through a thread-local message: the first failure is re-raised on the caller's thread.
template<class F>
int
for_shards(pqp_multi* m, F&& work)
{
  std::vector<size_t> live;
  for (size_t s = 0; s < m->shard.size(); ++s)
    if (m->count[s] > 0)
      live.push_back(s);
  if (live.empty())
    return PQP_OK;
  std::vector<int> rc(live.size(), PQP_OK);
  std::vector<std::string> msg(live.size());
  auto run = [&](size_t k) {
    rc[k] = work(live[k]);
    if (rc[k])
      msg[k] = pqp_last_error();
  };
  std::vector<std::thread> helpers;
  for (size_t k = 1; k < live.size(); ++k) {
    try {
      helpers.emplace_back(run, k);
    } catch (...) { // (no thread to be had: this shard is set up on the caller's thread -- nothing crosses the C boundary)
      run(k);
    }
  }
  run(0);
  for (std::thread& t : helpers)
    t.join();
  for (size_t k = 0; k < live.size(); ++k)
    if (rc[k])
      return fail(rc[k], msg[k]);
  return PQP_OK;
}

int
multi_setup(pqp_multi* m, setup_fn fn, int64_t idx, const double* H, const double* g, const double* A, const double* b,
            const double* C, const double* l, const double* u, const double* l_box, const double* u_box, int flag,
            double rho, double mu_eq, double mu_in, double min_eig)
{
  if (int rc = check_multi_idx(m, idx))
    return rc;
  if (idx >= 0) {
    const int s = shard_of(m, idx);
    return fn(m->shard[size_t(s)], idx - m->first[size_t(s)], H, g, A, b, C, l, u, l_box, u_box, flag, rho, mu_eq, mu_in,
              min_eig);
  }
  const size_t n = size_t(m->d.n), ne = size_t(m->d.n_eq), ni = size_t(m->d.n_in);
  // every shard takes its slice of the model at the same time (see for_shards): the host-to-device copies of the G
  // devices overlap instead of queueing behind one another
  return for_shards(m, [&](size_t s) {
    const int64_t f = m->first[s];
    return fn(m->shard[s], -1, off(H, f, n * n), off(g, f, n), off(A, f, ne * n), off(b, f, ne), off(C, f, ni * n),
              off(l, f, ni), off(u, f, ni), off(l_box, f, n), off(u_box, f, n), flag, rho, mu_eq, mu_in, min_eig);
  });
}

} // namespace

extern "C" {

int
pqp_multi_create(int64_t batch_size, int64_t dim, int64_t n_eq, int64_t n_in, int box_constraints, int hessian_type,
                 int dense_backend, const int* devices, int n_devices, pqp_multi** out)
{
  if (!out)
    return fail(PQP_ERR_INVALID_ARGUMENT, "null output handle");
  *out = nullptr;
  if (n_devices <= 0 || !devices)
    return fail(PQP_ERR_INVALID_ARGUMENT, "pqp_multi_create: at least one device is required");
  if (batch_size < 0)
    return fail(PQP_ERR_INVALID_ARGUMENT, "negative size");
  pqp_multi* m = new pqp_multi();
  m->B = batch_size;
  const int64_t G = n_devices;
  int64_t next = 0;
  for (int64_t g = 0; g < G; ++g) {
    const int64_t cnt = batch_size / G + (g < batch_size % G ? 1 : 0);
    pqp_batch* h = nullptr;
    int rc = pqp_batch_create(cnt, dim, n_eq, n_in, box_constraints, hessian_type, dense_backend, devices[g], &h);
    hipStream_t st = nullptr;
    if (!rc) {
      DeviceGuard guard(devices[g]);
      // non-blocking: the shards of one device (logical shards, tests) must not serialise on the null stream
      if (!guard.ok() || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess)
        rc = fail(PQP_ERR_HIP, "pqp_multi_create: no stream on device " + std::to_string(devices[g]));
    }
    if (!rc)
      rc = pqp_batch_set_stream(h, st);
    if (!rc)
      rc = pqp_batch_enable_host_results(h, 1);
    if (rc) {
      const std::string msg = pqp_last_error();
      if (h)
        pqp_batch_destroy(h);
      if (st) { // (not yet owned by m)
        DeviceGuard guard(devices[g]);
        if (guard.ok())
          (void)hipStreamDestroy(st);
      }
      pqp_multi_destroy(m);
      return fail(rc, msg);
    }
    m->shard.push_back(h);
    m->stream.push_back(st);
    m->first.push_back(next);
    m->count.push_back(cnt);
    m->pack.push_back(nullptr);
    m->in_flight.push_back(0);
    next += cnt;
  }
  m->d = m->shard[0]->dev.d;
  *out = m;
  return PQP_OK;
}

void
pqp_multi_destroy(pqp_multi* m)
{
  if (!m)
    return;
  for (size_t g = 0; g < m->shard.size(); ++g) {
    pqp_batch* h = m->shard[g];
    const int dev = h->device;
    pqp_batch_destroy(h); // (waits for a solve in flight)
    DeviceGuard guard(dev);
    if (guard.ok()) {
      if (m->pack[g])
        (void)hipFree(m->pack[g]);
      if (m->stream[g])
        (void)hipStreamDestroy(m->stream[g]);
    }
  }
  delete m;
}

int64_t
pqp_multi_size(const pqp_multi* m)
{
  return m ? m->B : 0;
}

int
pqp_multi_shard_count(const pqp_multi* m)
{
  return m ? int(m->shard.size()) : 0;
}

int
pqp_multi_shard(pqp_multi* m, int g, pqp_batch** shard, int64_t* first, int64_t* count)
{
  if (!m || g < 0 || size_t(g) >= m->shard.size())
    return fail(PQP_ERR_INVALID_ARGUMENT, "shard index out of range");
  if (shard)
    *shard = m->shard[size_t(g)];
  if (first)
    *first = m->first[size_t(g)];
  if (count)
    *count = m->count[size_t(g)];
  return PQP_OK;
}

int
pqp_multi_locate(const pqp_multi* m, int64_t idx, int* shard, int64_t* local)
{
  if (!m || idx < 0 || idx >= m->B)
    return fail(PQP_ERR_INVALID_ARGUMENT, "QP index out of range");
  const int s = shard_of(m, idx);
  if (shard)
    *shard = s;
  if (local)
    *local = idx - m->first[size_t(s)];
  return PQP_OK;
}

pqp_settings*
pqp_multi_settings(pqp_multi* m, int64_t idx)
{
  if (!m || idx < 0 || idx >= m->B)
    return nullptr;
  const int s = shard_of(m, idx);
  return pqp_batch_settings(m->shard[size_t(s)], idx - m->first[size_t(s)]);
}

int
pqp_multi_init(pqp_multi* m, int64_t idx, const double* H, const double* g, const double* A, const double* b,
               const double* C, const double* l, const double* u, const double* l_box, const double* u_box,
               int compute_preconditioner, double rho, double mu_eq, double mu_in, double manual_minimal_H_eigenvalue)
{
  return multi_setup(m, pqp_batch_init, idx, H, g, A, b, C, l, u, l_box, u_box, compute_preconditioner, rho, mu_eq, mu_in,
                     manual_minimal_H_eigenvalue);
}

int
pqp_multi_update(pqp_multi* m, int64_t idx, const double* H, const double* g, const double* A, const double* b,
                 const double* C, const double* l, const double* u, const double* l_box, const double* u_box,
                 int update_preconditioner, double rho, double mu_eq, double mu_in, double manual_minimal_H_eigenvalue)
{
  return multi_setup(m, pqp_batch_update, idx, H, g, A, b, C, l, u, l_box, u_box, update_preconditioner, rho, mu_eq, mu_in,
                     manual_minimal_H_eigenvalue);
}

int
pqp_multi_warm_start(pqp_multi* m, int64_t idx, const double* x, const double* y, const double* z)
{
  if (int rc = check_multi_idx(m, idx))
    return rc;
  if (idx >= 0) {
    const int s = shard_of(m, idx);
    return pqp_batch_warm_start(m->shard[size_t(s)], idx - m->first[size_t(s)], x, y, z);
  }
  const size_t n = size_t(m->d.n), ne = size_t(m->d.n_eq), nc = size_t(m->d.nc);
  for (size_t s = 0; s < m->shard.size(); ++s)
    if (m->count[s] > 0)
      if (int rc = pqp_batch_warm_start(m->shard[s], -1, off(x, m->first[s], n), off(y, m->first[s], ne),
                                        off(z, m->first[s], nc)))
        return rc;
  return PQP_OK;
}

int
pqp_multi_cleanup(pqp_multi* m, int64_t idx)
{
  if (int rc = check_multi_idx(m, idx))
    return rc;
  if (idx >= 0) {
    const int s = shard_of(m, idx);
    return pqp_batch_cleanup(m->shard[size_t(s)], idx - m->first[size_t(s)]);
  }
  for (size_t s = 0; s < m->shard.size(); ++s)
    if (m->count[s] > 0)
      if (int rc = pqp_batch_cleanup(m->shard[s], -1))
        return rc;
  return PQP_OK;
}

int
pqp_multi_flush(pqp_multi* m)
{
  if (!m)
    return fail(PQP_ERR_INVALID_ARGUMENT, "null multi-device handle");
  // (set-up kernel + its synchronisation + the read-back of the structure flags, every shard at the same time)
  return for_shards(m, [&](size_t s) { return pqp_batch_flush(m->shard[s]); });
}

int
pqp_multi_solve_range_async(pqp_multi* m, int64_t first, int64_t count)
{
  if (!m)
    return fail(PQP_ERR_INVALID_ARGUMENT, "null multi-device handle");
  if (first < 0 || count < 0 || first + count > m->B)
    return fail(PQP_ERR_INVALID_ARGUMENT, "solve range outside the batch");
  // every shard's launch is enqueued before any of them is waited for
  for (size_t s = 0; s < m->shard.size(); ++s) {
    const int64_t lo = std::max(first, m->first[s]), hi = std::min(first + count, m->first[s] + m->count[s]);
    if (lo >= hi)
      continue;
    if (int rc = pqp_batch_solve_range_async(m->shard[s], lo - m->first[s], hi - lo)) {
      const std::string msg = pqp_last_error();
      (void)pqp_multi_wait(m);
      return fail(rc, msg);
    }
    m->in_flight[s] = 1;
  }
  return PQP_OK;
}

int
pqp_multi_wait(pqp_multi* m)
{
  if (!m)
    return fail(PQP_ERR_INVALID_ARGUMENT, "null multi-device handle");
  int rc_all = PQP_OK;
  std::string msg;
  for (size_t s = 0; s < m->shard.size(); ++s) {
    if (!m->in_flight[s])
      continue;
    m->in_flight[s] = 0;
    if (int rc = pqp_batch_wait(m->shard[s])) {
      if (rc_all == PQP_OK) {
        rc_all = rc;
        msg = pqp_last_error();
      }
    }
  }
  return rc_all == PQP_OK ? PQP_OK : fail(rc_all, msg);
}

int
pqp_multi_solve_async(pqp_multi* m)
{
  return pqp_multi_solve_range_async(m, 0, m ? m->B : 0);
}

int
pqp_multi_solve_range(pqp_multi* m, int64_t first, int64_t count)
{
  if (int rc = pqp_multi_solve_range_async(m, first, count))
    return rc;
  return pqp_multi_wait(m);
}

int
pqp_multi_solve(pqp_multi* m)
{
  return pqp_multi_solve_range(m, 0, m ? m->B : 0);
}

int
pqp_multi_get_results(pqp_multi* m, int64_t idx, double* x, double* y, double* z, double* se, double* si, pqp_info* info)
{
  if (int rc = check_multi_idx(m, idx))
    return rc;
  if (int rc = pqp_multi_wait(m))
    return rc;
  if (idx >= 0) {
    const int s = shard_of(m, idx);
    return pqp_batch_get_results(m->shard[size_t(s)], idx - m->first[size_t(s)], x, y, z, se, si, info);
  }
  const size_t n = size_t(m->d.n), ne = size_t(m->d.n_eq), nc = size_t(m->d.nc);
  for (size_t s = 0; s < m->shard.size(); ++s) {
    if (m->count[s] == 0)
      continue;
    const int64_t f = m->first[s];
    if (int rc = pqp_batch_get_results(m->shard[s], -1, off(x, f, n), off(y, f, ne), off(z, f, nc), off(se, f, ne),
                                       off(si, f, nc), info ? info + f : nullptr))
      return rc;
  }
  return PQP_OK;
}

int
pqp_multi_get_trace(pqp_multi* m, int64_t idx, double* records, int64_t capacity, int64_t* n_records)
{
  if (!m || !n_records)
    return fail(PQP_ERR_INVALID_ARGUMENT, "null argument");
  if (idx < 0)
    return fail(PQP_ERR_INVALID_ARGUMENT, "QP index out of range");
  if (int rc = check_multi_idx(m, idx))
    return rc;
  if (int rc = pqp_multi_wait(m))
    return rc;
  const int s = shard_of(m, idx);
  return pqp_batch_get_trace(m->shard[size_t(s)], idx - m->first[size_t(s)], records, capacity, n_records);
}

int
pqp_multi_gather_device(pqp_multi* m, int root_shard, double* out)
{
  if (!m || !out)
    return fail(PQP_ERR_INVALID_ARGUMENT, "null argument");
  if (root_shard < 0 || size_t(root_shard) >= m->shard.size())
    return fail(PQP_ERR_INVALID_ARGUMENT, "root shard out of range");
  const size_t width = size_t(m->d.n) + size_t(m->d.n_eq) + size_t(m->d.nc) + 2;
  const int root_dev = m->shard[size_t(root_shard)]->device;
  // pack on every shard's stream (ordered behind a solve in flight on that stream), then the copy into place on the
  // same stream: G independent chains, one synchronisation each at the end.  A failure on one shard does not return
  // before the chains already enqueued on the others have drained: they write into the caller's buffer.
  int rc = PQP_OK;
  std::string msg;
  auto bad = [&](int code, const std::string& what) {
    if (rc == PQP_OK) {
      rc = code;
      msg = what;
    }
  };
  for (size_t s = 0; s < m->shard.size() && rc == PQP_OK; ++s) {
    if (m->count[s] == 0)
      continue;
    pqp_batch* h = m->shard[s];
    DeviceGuard guard(h->device);
    if (!guard.ok()) {
      bad(PQP_ERR_HIP, "hipSetDevice failed in pqp_multi_gather_device");
      break;
    }
    double* dst = out + size_t(m->first[s]) * width;
    const size_t bytes = size_t(m->count[s]) * width * sizeof(double);
    if (h->device == root_dev) {
      // same device: the pack kernel writes straight into the caller's buffer
      if (int r = pqp_batch_pack_results(h, 0, m->count[s], dst, m->stream[s]))
        bad(r, pqp_last_error());
      continue;
    }
    hipError_t e = hipSuccess;
    if (!m->pack[s] && (e = hipMalloc(reinterpret_cast<void**>(&m->pack[s]), bytes)) != hipSuccess) {
      bad(PQP_ERR_HIP, std::string("hipMalloc of the pack buffer: ") + hipGetErrorString(e));
      break;
    }
    if (int r = pqp_batch_pack_results(h, 0, m->count[s], m->pack[s], m->stream[s])) {
      bad(r, pqp_last_error());
      break;
    }
    // direct xGMI copy where the two devices can map each other (enabled once per pair, from the source device: the
    // current one); otherwise hipMemcpyPeerAsync stages through the host by itself -- slower, still correct
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, h->device, root_dev) == hipSuccess && can) {
      e = hipDeviceEnablePeerAccess(root_dev, 0);
      if (e == hipErrorPeerAccessAlreadyEnabled)
        (void)hipGetLastError(); // (clear the sticky code: not an error)
      else if (e != hipSuccess) {
        bad(PQP_ERR_HIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
        break;
      }
    }
    if ((e = hipMemcpyPeerAsync(dst, root_dev, m->pack[s], h->device, bytes, m->stream[s])) != hipSuccess)
      bad(PQP_ERR_HIP, std::string("hipMemcpyPeerAsync: ") + hipGetErrorString(e));
  }
  for (size_t s = 0; s < m->shard.size(); ++s) {
    if (m->count[s] == 0)
      continue;
    DeviceGuard guard(m->shard[s]->device);
    if (!guard.ok()) {
      bad(PQP_ERR_HIP, "hipSetDevice failed in pqp_multi_gather_device");
      continue;
    }
    const hipError_t e = hipStreamSynchronize(m->stream[s]);
    if (e != hipSuccess)
      bad(PQP_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
  }
  // (the solves those streams carried have finished with them: run their host-side bookkeeping)
  const int wrc = pqp_multi_wait(m);
  if (rc != PQP_OK)
    return fail(rc, msg);
  return wrc;
}

double
pqp_multi_last_solve_ms(const pqp_multi* m)
{
  double worst = 0.0;
  if (m)
    for (pqp_batch* h : m->shard)
      worst = std::max(worst, pqp_batch_last_solve_ms(h));
  return worst;
}

} // extern "C"
