// proxsuite/proxqp/dense/wrapper.hpp -- dense::QP<T>, dense::BatchQP<T> and dense::solve of
// the ProxQP API on MI355X.
//
// Same class and member names, argument order, defaults and exceptions as the reference
// (include/proxsuite/proxqp/dense/wrapper.hpp: QP :114-963, free solve :1000-1236,
// BatchQP :1253-1311), so code such as benchmark/timings-parallel.cpp:178-220 recompiles
// against this header.  What differs is ownership: the numerical state of a QP lives on the
// GPU inside a *pool* -- one C-ABI batch handle (include/proxqp_hip.h) holding `capacity`
// QPs of one signature (dim, n_eq, n_in, box, Hessian type, backend).  A QP object is a
// (pool, slot) view plus host copies of `settings`, `results`, `model`.  Standalone QPs of one
// signature share pools too (a per-thread registry hands out and takes back slots), so that
// dense::solve_in_parallel(std::vector<QP>&) -- the reference's most common calling form -- is one
// kernel launch per pool exactly like the BatchQP overload.  Copying a QP copies its device state
// into a fresh slot (the reference's QP is a value type; `qps.push_back(qp)` must not alias).
//
// Host protocol of every call: settings -> device record, C-ABI call, device record ->
// settings (init/update/solve(x,y,z) change default_rho, compute_preconditioner,
// initial_guess, ... exactly where the reference does), then results <- device.
//
// Only T = double is supported (the device path computes in fp64).  Link with
// -lproxqp_hip (proxsuite_amd/csrc/libproxqp_hip.so).  No CPU fallback: without a HIP
// device the first constructor throws std::runtime_error.
#ifndef PROXSUITE_AMD_PROXQP_DENSE_WRAPPER_HPP
#define PROXSUITE_AMD_PROXQP_DENSE_WRAPPER_HPP

#include <algorithm>
#include <cstring>
#include <deque>
#include <limits>
#include <map>
#include <mutex>
#include <vector>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>

#include "proxqp_hip.h"
#include "proxsuite/proxqp/dense/model.hpp"
#include "proxsuite/proxqp/results.hpp"

namespace proxsuite {
namespace proxqp {
namespace dense {

namespace detail {

inline void
check(int rc)
{
  if (rc == PQP_OK)
    return;
  std::string msg = pqp_last_error();
  if (rc == PQP_ERR_INVALID_ARGUMENT)
    throw std::invalid_argument(msg);
  throw std::runtime_error("libproxqp_hip: " + msg);
}

[[noreturn]] inline void
bad_size(const char* what, isize got, isize expected)
{
  // counterpart of PROXSUITE_CHECK_ARGUMENT_SIZE (reference helpers/../macros.hpp:18-35)
  throw std::invalid_argument(std::string("wrong argument size: expected ") + std::to_string(expected) +
                              ", got " + std::to_string(got) + "\nhint: " + what);
}

// A batch spread over several devices (pqp_multi_*): its shards are ordinary batch handles that the pools of a
// BatchQP adopt; the owner lives as long as any of those pools.
struct MultiOwner
{
  pqp_multi* m = nullptr;
  explicit MultiOwner(pqp_multi* m_)
    : m(m_)
  {
  }
  MultiOwner(const MultiOwner&) = delete;
  MultiOwner& operator=(const MultiOwner&) = delete;
  ~MultiOwner() { pqp_multi_destroy(m); }
};

struct Pool
{
  pqp_batch* h = nullptr;
  std::shared_ptr<MultiOwner> multi; // set: `h` is shard `shard` of that multi-device batch (not owned here)
  int shard = 0;
  isize first = 0; // index of slot 0 in the whole multi-device batch
  isize capacity = 0, used = 0;
  isize dim = 0, n_eq = 0, n_in = 0, n_c = 0;
  std::vector<isize> free_slots; // registry pools only: slots given back by destroyed QPs
  // The handle's host state (queued commands, launch range, events, settings upload) is shared by every
  // QP of the pool.  With the reference each QP is an island, so `#pragma omp parallel for` over
  // qps[i].solve() is legal user code: every QP method that touches `h`, and the slot bookkeeping, runs
  // under this lock (recursive: solve() -> pull()).
  mutable std::recursive_mutex mtx;
  Pool(isize cap, isize dim_, isize n_eq_, isize n_in_, bool box, HessianType hessian, DenseBackend backend,
       int device)
    : capacity(cap)
    , dim(dim_)
    , n_eq(n_eq_)
    , n_in(n_in_)
    , n_c(n_in_ + (box ? dim_ : 0))
  {
    check(pqp_batch_create(cap, dim_, n_eq_, n_in_, box ? 1 : 0, int(hessian), int(backend), device, &h));
    // results are host members when a solve returns (reference parallel/qp_solve.hpp:33-37): the solve kernel
    // writes them into pinned host mirrors; a stream of its own lets this pool overlap with the others
    check(pqp_batch_enable_host_results(h, 1));
    check(pqp_batch_own_stream(h));
  }
  // shard `g` of a multi-device batch (already has its stream and its host mirrors)
  Pool(std::shared_ptr<MultiOwner> owner, int g, isize dim_, isize n_eq_, isize n_in_, bool box)
    : multi(std::move(owner))
    , shard(g)
    , dim(dim_)
    , n_eq(n_eq_)
    , n_in(n_in_)
    , n_c(n_in_ + (box ? dim_ : 0))
  {
    int64_t f = 0, c = 0;
    check(pqp_multi_shard(multi->m, g, &h, &f, &c));
    first = isize(f);
    capacity = isize(c);
  }
  Pool(const Pool&) = delete;
  Pool& operator=(const Pool&) = delete;
  ~Pool()
  {
    if (!multi)
      pqp_batch_destroy(h);
  }
};
using PoolLock = std::lock_guard<std::recursive_mutex>;

template<typename T>
inline T
opt_or_nan(const optional<T>& v)
{
  return v ? *v : std::numeric_limits<T>::quiet_NaN();
}

// Pools of the standalone QPs of this thread, by signature.  The registry only OBSERVES its pools
// (weak_ptr): a pool lives as long as a QP holds it and its device memory goes away with the last of
// them.  Pool sizes grow geometrically with the number of live QPs of the signature (1, 1, 2, 4, ...),
// capped so that one pool stays below ~256 MB of device memory: one QP costs one QP.
struct Registry
{
  using Key = std::tuple<isize, isize, isize, bool, int, int, int>;
  std::map<Key, std::vector<std::weak_ptr<Pool>>> pools;
  static Registry& instance()
  {
    static thread_local Registry r;
    return r;
  }
  static isize chunk_max(isize dim, isize n_eq, isize n_in, bool box)
  {
    const double n = double(dim), nd = double(n_eq + n_in + (box ? dim : 0));
    const double bytes = 8.0 * (5.0 * n * n + 6.0 * nd * n + 3.0 * nd * nd) + 4096.0;
    const double cap = 268435456.0 / bytes;
    return isize(std::max(1.0, std::min(256.0, cap)));
  }
  // -> (pool, slot, recycled): a recycled slot must be reset by its new owner (pqp_batch_reset_qp)
  std::tuple<std::shared_ptr<Pool>, isize, bool> acquire(isize dim, isize n_eq, isize n_in, bool box,
                                                         HessianType hessian, DenseBackend backend, int device)
  {
    auto& v = pools[Key{ dim, n_eq, n_in, box, int(hessian), int(backend), device }];
    isize live_capacity = 0;
    for (auto it = v.begin(); it != v.end();) {
      std::shared_ptr<Pool> p = it->lock();
      if (!p) {
        it = v.erase(it);
        continue;
      }
      PoolLock lock(p->mtx);
      if (!p->free_slots.empty()) {
        const isize s = p->free_slots.back();
        p->free_slots.pop_back();
        return { p, s, true };
      }
      if (p->used < p->capacity)
        return { p, p->used++, false };
      live_capacity += p->capacity;
      ++it;
    }
    const isize cap = std::max<isize>(1, std::min(chunk_max(dim, n_eq, n_in, box), live_capacity));
    auto p = std::make_shared<Pool>(cap, dim, n_eq, n_in, box, hessian, backend, device);
    v.push_back(p);
    p->used = 1;
    return { p, 0, false };
  }
};

} // namespace detail

template<typename T>
struct BatchQP;

template<typename T>
struct QP
{
  static_assert(std::is_same<T, double>::value, "the MI355X dense backend computes in fp64: use QP<double>");

private:
  std::shared_ptr<detail::Pool> pool_;
  isize slot_ = 0;
  bool owns_slot_ = false; // slot taken from the registry (standalone QP): given back on destruction
  int device_ = 0;
  DenseBackend dense_backend;
  bool box_constraints;
  HessianType hessian_type;

public:
  Results<T> results;
  Settings<T> settings;
  Model<T> model;

  // value semantics of the reference's QP: a copy owns its own device state
  QP(const QP& o)
    : owns_slot_(true)
    , device_(o.device_)
    , dense_backend(o.dense_backend)
    , box_constraints(o.box_constraints)
    , hessian_type(o.hessian_type)
    , results(o.results)
    , settings(o.settings)
    , model(o.model)
  {
    auto ps = detail::Registry::instance().acquire(model.dim, model.n_eq, model.n_in, box_constraints, hessian_type,
                                                   o.requested_backend_, device_);
    pool_ = std::get<0>(ps);
    slot_ = std::get<1>(ps);
    requested_backend_ = o.requested_backend_;
    // (every per-QP array, the settings and the flags are overwritten by the copy: no reset needed)
    if (pool_ == o.pool_) {
      detail::PoolLock lock(pool_->mtx);
      o.push_settings();
      detail::check(pqp_batch_copy_qp(pool_->h, slot_, o.pool_->h, o.slot_));
    } else {
      std::lock(pool_->mtx, o.pool_->mtx);
      detail::PoolLock la(pool_->mtx, std::adopt_lock), lb(o.pool_->mtx, std::adopt_lock);
      o.push_settings();
      detail::check(pqp_batch_copy_qp(pool_->h, slot_, o.pool_->h, o.slot_));
    }
  }
  QP& operator=(const QP& o)
  {
    if (this != &o) {
      QP tmp(o);
      swap(tmp);
    }
    return *this;
  }
  QP(QP&& o) noexcept
    : pool_(std::move(o.pool_))
    , slot_(o.slot_)
    , owns_slot_(o.owns_slot_)
    , device_(o.device_)
    , dense_backend(o.dense_backend)
    , box_constraints(o.box_constraints)
    , hessian_type(o.hessian_type)
    , results(std::move(o.results))
    , settings(std::move(o.settings))
    , model(std::move(o.model))
    , requested_backend_(o.requested_backend_)
  {
    o.owns_slot_ = false;
  }
  QP& operator=(QP&& o) noexcept
  {
    if (this != &o)
      swap(o);
    return *this;
  }
  ~QP() { release(); }
  void swap(QP& o) noexcept
  {
    std::swap(pool_, o.pool_);
    std::swap(slot_, o.slot_);
    std::swap(owns_slot_, o.owns_slot_);
    std::swap(device_, o.device_);
    std::swap(dense_backend, o.dense_backend);
    std::swap(box_constraints, o.box_constraints);
    std::swap(hessian_type, o.hessian_type);
    std::swap(results, o.results);
    std::swap(settings, o.settings);
    std::swap(model, o.model);
    std::swap(requested_backend_, o.requested_backend_);
  }

  // the 8 constructor overloads of the reference (wrapper.hpp:140-333)
  QP(isize dim, isize n_eq, isize n_in, bool box, HessianType hessian, DenseBackend backend)
    : QP(dim, n_eq, n_in, box, hessian, backend, nullptr, 0, 0)
  {
  }
  QP(isize dim, isize n_eq, isize n_in, bool box, DenseBackend backend, HessianType hessian)
    : QP(dim, n_eq, n_in, box, hessian, backend)
  {
  }
  QP(isize dim, isize n_eq, isize n_in, bool box, HessianType hessian)
    : QP(dim, n_eq, n_in, box, hessian, DenseBackend::Automatic)
  {
  }
  QP(isize dim, isize n_eq, isize n_in, bool box, DenseBackend backend)
    : QP(dim, n_eq, n_in, box, HessianType::Dense, backend)
  {
  }
  QP(isize dim, isize n_eq, isize n_in, bool box)
    : QP(dim, n_eq, n_in, box, HessianType::Dense, DenseBackend::Automatic)
  {
  }
  QP(isize dim, isize n_eq, isize n_in, HessianType hessian)
    : QP(dim, n_eq, n_in, false, hessian, DenseBackend::Automatic)
  {
  }
  QP(isize dim, isize n_eq, isize n_in, DenseBackend backend)
    : QP(dim, n_eq, n_in, false, HessianType::Dense, backend)
  {
  }
  QP(isize dim, isize n_eq, isize n_in)
    : QP(dim, n_eq, n_in, false, HessianType::Dense, DenseBackend::Automatic)
  {
  }

  bool is_box_constrained() const { return box_constraints; }
  DenseBackend which_dense_backend() const { return dense_backend; }
  HessianType which_hessian_type() const { return hessian_type; }

  // QP::init without box constraints (reference wrapper.hpp:354-498)
  void init(optional<MatRef<T>> H, optional<VecRef<T>> g, optional<MatRef<T>> A, optional<VecRef<T>> b,
            optional<MatRef<T>> C, optional<VecRef<T>> l, optional<VecRef<T>> u,
            bool compute_preconditioner = true, optional<T> rho = nullopt, optional<T> mu_eq = nullopt,
            optional<T> mu_in = nullopt, optional<T> manual_minimal_H_eigenvalue = nullopt)
  {
    if (box_constraints)
      throw std::invalid_argument("wrong model setup: the QP object is designed with box constraints, but is "
                                  "initialized without lower or upper box inequalities.");
    setup(true, H, g, A, b, C, l, u, nullopt, nullopt, compute_preconditioner, rho, mu_eq, mu_in,
          manual_minimal_H_eigenvalue);
  }
  // QP::init with box constraints (reference wrapper.hpp:520-703)
  void init(optional<MatRef<T>> H, optional<VecRef<T>> g, optional<MatRef<T>> A, optional<VecRef<T>> b,
            optional<MatRef<T>> C, optional<VecRef<T>> l, optional<VecRef<T>> u, optional<VecRef<T>> l_box,
            optional<VecRef<T>> u_box, bool compute_preconditioner = true, optional<T> rho = nullopt,
            optional<T> mu_eq = nullopt, optional<T> mu_in = nullopt,
            optional<T> manual_minimal_H_eigenvalue = nullopt)
  {
    require_box(l_box, u_box);
    setup(true, H, g, A, b, C, l, u, l_box, u_box, compute_preconditioner, rho, mu_eq, mu_in,
          manual_minimal_H_eigenvalue);
  }
  // QP::update (reference wrapper.hpp:723-807 and, with boxes, :831-918)
  void update(optional<MatRef<T>> H, optional<VecRef<T>> g, optional<MatRef<T>> A, optional<VecRef<T>> b,
              optional<MatRef<T>> C, optional<VecRef<T>> l, optional<VecRef<T>> u,
              bool update_preconditioner = false, optional<T> rho = nullopt, optional<T> mu_eq = nullopt,
              optional<T> mu_in = nullopt, optional<T> manual_minimal_H_eigenvalue = nullopt)
  {
    if (box_constraints)
      throw std::invalid_argument("wrong model setup: the QP object is designed with box constraints, but the "
                                  "update does not take into account lower or upper box inequalities.");
    setup(false, H, g, A, b, C, l, u, nullopt, nullopt, update_preconditioner, rho, mu_eq, mu_in,
          manual_minimal_H_eigenvalue);
  }
  void update(optional<MatRef<T>> H, optional<VecRef<T>> g, optional<MatRef<T>> A, optional<VecRef<T>> b,
              optional<MatRef<T>> C, optional<VecRef<T>> l, optional<VecRef<T>> u, optional<VecRef<T>> l_box,
              optional<VecRef<T>> u_box, bool update_preconditioner = false, optional<T> rho = nullopt,
              optional<T> mu_eq = nullopt, optional<T> mu_in = nullopt,
              optional<T> manual_minimal_H_eigenvalue = nullopt)
  {
    require_box(l_box, u_box);
    setup(false, H, g, A, b, C, l, u, l_box, u_box, update_preconditioner, rho, mu_eq, mu_in,
          manual_minimal_H_eigenvalue);
  }

  // QP::solve() (reference wrapper.hpp:922-939)
  void solve()
  {
    detail::PoolLock lock(pool_->mtx);
    push_settings();
    detail::check(pqp_batch_solve_range(pool_->h, slot_, 1));
    pull_from_mirrors();
  }
  // QP::solve(x, y, z) (reference wrapper.hpp:940-957; warm_start helpers.hpp:715-763)
  void solve(optional<VecRef<T>> x, optional<VecRef<T>> y, optional<VecRef<T>> z)
  {
    detail::PoolLock lock(pool_->mtx);
    push_settings();
    std::vector<T> tx, ty, tz;
    const T* px = pack_vec(x, model.dim, tx, "the dimension wrt primal variable x for warm start is not valid.");
    const T* py =
      pack_vec(y, model.n_eq, ty, "the dimension wrt equality constrained variables for warm start is not valid.");
    const T* pz = pack_vec(
      z, pool_->n_c, tz, "the dimension wrt inequality constrained variables for warm start is not valid.");
    detail::check(pqp_batch_warm_start(pool_->h, slot_, px, py, pz));
    pull_settings();
    solve();
  }
  // QP::cleanup (reference wrapper.hpp:958-962)
  void cleanup()
  {
    detail::PoolLock lock(pool_->mtx);
    detail::check(pqp_batch_cleanup(pool_->h, slot_));
    pull();
  }

  // -- used by BatchQP / solve_in_parallel ------------------------------------------------
  void push_settings() const { settings.to_c(*pqp_batch_settings(pool_->h, slot_)); }
  void pull_settings() { settings.from_c(*pqp_batch_settings(pool_->h, slot_)); }
  void pull()
  {
    detail::PoolLock lock(pool_->mtx);
    pull_settings();
    pqp_info info;
    detail::check(pqp_batch_get_results(pool_->h, slot_, results.x.data(), results.y.data(), results.z.data(),
                                        results.se.data(), results.si.data(), &info));
    results.info.from_c(info);
  }
  // results of this QP straight from the pool's host mirrors (valid after a finished solve of the slot: the solve
  // kernel's epilogue wrote them; no device-to-host copy).  The caller holds the pool lock.
  void pull_from_mirrors()
  {
    pull_settings();
    const double *mx, *my, *mz, *mse, *msi;
    const pqp_info* mi;
    detail::check(pqp_batch_host_results(pool_->h, &mx, &my, &mz, &mse, &msi, &mi));
    if (!pqp_batch_host_results_fresh(pool_->h, slot_)) { // (cannot happen behind a solve of this slot; stay correct anyway)
      pull();
      return;
    }
    const usize s = usize(slot_);
    auto put = [s](Vec<T>& dst, const double* src) {
      if (dst.size())
        std::memcpy(dst.data(), src + s * usize(dst.size()), usize(dst.size()) * sizeof(T));
    };
    put(results.x, mx);
    put(results.y, my);
    put(results.z, mz);
    put(results.se, mse);
    put(results.si, msi);
    results.info.from_c(mi[s]);
  }
  const std::shared_ptr<detail::Pool>& pool() const { return pool_; }
  isize slot() const { return slot_; }

private:
  friend struct BatchQP<T>;
  DenseBackend requested_backend_ = DenseBackend::Automatic; // as passed to the constructor (pool key)
  void release() noexcept
  {
    if (owns_slot_ && pool_) {
      // the slot goes back to its pool; its next owner resets it (pqp_batch_reset_qp in the constructor)
      detail::PoolLock lock(pool_->mtx);
      pool_->free_slots.push_back(slot_);
    }
    owns_slot_ = false;
    pool_.reset();
  }
  QP(isize dim, isize n_eq, isize n_in, bool box, HessianType hessian, DenseBackend backend,
     std::shared_ptr<detail::Pool> pool, isize slot, int device)
    : pool_(std::move(pool))
    , slot_(slot)
    , device_(device)
    , dense_backend(backend)
    , box_constraints(box)
    , hessian_type(hessian)
    , results(dim, n_eq, n_in, box)
    , settings(DenseBackend::PrimalDualLDLT)
    , model(dim, n_eq, n_in, box)
    , requested_backend_(backend)
  {
    if (!pool_) {
      if (dim <= 0) // reference dense/model.hpp:65-68 (checked here before a registry pool is created)
        throw std::invalid_argument(
          "wrong argument size: the dimension wrt the primal variable x should be strictly positive.");
      auto ps = detail::Registry::instance().acquire(dim, n_eq, n_in, box, hessian, backend, device);
      pool_ = std::get<0>(ps);
      slot_ = std::get<1>(ps);
      owns_slot_ = true;
      if (std::get<2>(ps)) {
        // a slot another QP object used before: a new QP starts from Settings / Results / Model defaults
        // (reference wrapper.hpp:140-333), not from what its predecessor left there
        detail::PoolLock lock(pool_->mtx);
        detail::check(pqp_batch_reset_qp(pool_->h, slot_));
      }
    }
    detail::PoolLock lock(pool_->mtx);
    dense_backend = DenseBackend(pqp_batch_dense_backend(pool_->h)); // Automatic resolved (wrapper.hpp:81-113)
    pull_settings();                                                  // defaults of that backend
    pqp_info info;
    detail::check(pqp_batch_get_results(pool_->h, slot_, nullptr, nullptr, nullptr, nullptr, nullptr, &info));
    results.info.from_c(info);
  }

  void require_box(const optional<VecRef<T>>& l_box, const optional<VecRef<T>>& u_box) const
  {
    const bool given = (l_box && l_box->size() != 0) || (u_box && u_box->size() != 0);
    if (!box_constraints && given)
      throw std::invalid_argument("wrong model setup: the QP object is designed without box constraints, but is "
                                  "used with lower or upper box inequalities.");
  }

  static const T* pack_vec(const optional<VecRef<T>>& v, isize expected, std::vector<T>& tmp, const char* hint)
  {
    if (!v || v->size() == 0)
      return nullptr;
    if (v->size() != expected)
      detail::bad_size(hint, v->size(), expected);
    if (v->stride == 1)
      return v->ptr;
    tmp.resize(usize(expected));
    for (isize i = 0; i < expected; ++i)
      tmp[usize(i)] = (*v)[i];
    return tmp.data();
  }
  static const T* pack_mat(const optional<MatRef<T>>& m, isize rows, isize cols, std::vector<T>& tmp,
                           const char* row_hint, const char* col_hint)
  {
    if (!m || m->size() == 0)
      return nullptr;
    if (m->rows() != rows)
      detail::bad_size(row_hint, m->rows(), rows);
    if (m->cols() != cols)
      detail::bad_size(col_hint, m->cols(), cols);
    if (m->is_packed_row_major())
      return m->ptr;
    tmp.resize(usize(rows * cols));
    for (isize i = 0; i < rows; ++i)
      for (isize j = 0; j < cols; ++j)
        tmp[usize(i * cols + j)] = (*m)(i, j);
    return tmp.data();
  }
  template<typename Dst>
  static void remember(Dst& dst, const T* src, T lo, T hi)
  {
    if (!src)
      return;
    T* d = dst.data();
    for (isize i = 0; i < dst.size(); ++i)
      d[i] = std::min(std::max(src[i], lo), hi);
  }

  void setup(bool is_init, const optional<MatRef<T>>& H, const optional<VecRef<T>>& g,
             const optional<MatRef<T>>& A, const optional<VecRef<T>>& b, const optional<MatRef<T>>& C,
             const optional<VecRef<T>>& l, const optional<VecRef<T>>& u, const optional<VecRef<T>>& l_box,
             const optional<VecRef<T>>& u_box, bool preconditioner_flag, const optional<T>& rho,
             const optional<T>& mu_eq, const optional<T>& mu_in, const optional<T>& min_eig)
  {
    detail::PoolLock lock(pool_->mtx);
    const isize n = model.dim, ne = model.n_eq, ni = model.n_in;
    std::vector<T> tH, tg, tA, tb, tC, tl, tu, tlb, tub;
    // the checks of the reference (wrapper.hpp:380-451 / :744-797), in its order
    const T* pg = pack_vec(g, n, tg, "the dimension wrt the primal variable x variable for g is not valid.");
    const T* pb = pack_vec(b, ne, tb, "the dimension wrt equality constrained variables for b is not valid.");
    const T* pu = pack_vec(u, ni, tu, "the dimension wrt inequality constrained variables for u is not valid.");
    const T* pl = pack_vec(l, ni, tl, "the dimension wrt inequality constrained variables for l is not valid.");
    const T* plb = pack_vec(l_box, n, tlb, "the dimension wrt box inequality constrained variables for l_box is not valid.");
    const T* pub = pack_vec(u_box, n, tub, "the dimension wrt box inequality constrained variables for u_box is not valid.");
    const T* pH = pack_mat(H, n, n, tH, "the row dimension for H is not valid.", "the column dimension for H is not valid.");
    const T* pA = pack_mat(A, ne, n, tA, "the row dimension for A is not valid.", "the column dimension for A is not valid.");
    const T* pC = pack_mat(C, ni, n, tC, "the row dimension for C is not valid.", "the column dimension for C is not valid.");
    push_settings();
    auto fn = is_init ? &pqp_batch_init : &pqp_batch_update;
    detail::check(fn(pool_->h, slot_, pH, pg, pA, pb, pC, pl, pu, plb, pub, preconditioner_flag ? 1 : 0,
                     detail::opt_or_nan(rho), detail::opt_or_nan(mu_eq), detail::opt_or_nan(mu_in),
                     detail::opt_or_nan(min_eig)));
    const T inf = std::numeric_limits<T>::infinity();
    remember(model.H, pH, -inf, inf);
    remember(model.g, pg, -inf, inf);
    remember(model.A, pA, -inf, inf);
    remember(model.b, pb, -inf, inf);
    remember(model.C, pC, -inf, inf);
    // bounds are clamped to +-1e20 (reference helpers.hpp:588-612)
    remember(model.l, pl, T(-1e20), inf);
    remember(model.u, pu, -inf, T(1e20));
    if (box_constraints) {
      remember(model.l_box, plb, T(-1e20), inf);
      remember(model.u_box, pub, -inf, T(1e20));
    }
    pull();
  }
};

// One-shot solve without box constraints (reference wrapper.hpp:1000-1092)
template<typename T>
Results<T>
solve(optional<MatRef<T>> H, optional<VecRef<T>> g, optional<MatRef<T>> A, optional<VecRef<T>> b,
      optional<MatRef<T>> C, optional<VecRef<T>> l, optional<VecRef<T>> u, optional<VecRef<T>> x = nullopt,
      optional<VecRef<T>> y = nullopt, optional<VecRef<T>> z = nullopt, optional<T> eps_abs = nullopt,
      optional<T> eps_rel = nullopt, optional<T> rho = nullopt, optional<T> mu_eq = nullopt,
      optional<T> mu_in = nullopt, optional<bool> verbose = nullopt, bool compute_preconditioner = true,
      bool compute_timings = false, optional<isize> max_iter = nullopt,
      InitialGuessStatus initial_guess = InitialGuessStatus::EQUALITY_CONSTRAINED_INITIAL_GUESS,
      bool check_duality_gap = false, optional<T> eps_duality_gap_abs = nullopt,
      optional<T> eps_duality_gap_rel = nullopt, bool primal_infeasibility_solving = false,
      optional<T> manual_minimal_H_eigenvalue = nullopt)
{
  isize n = H ? H->rows() : 0, n_eq = A ? A->rows() : 0, n_in = C ? C->rows() : 0;
  QP<T> Qp(n, n_eq, n_in, false, DenseBackend::PrimalDualLDLT);
  Qp.settings.initial_guess = initial_guess;
  Qp.settings.check_duality_gap = check_duality_gap;
  if (eps_abs)
    Qp.settings.eps_abs = *eps_abs;
  if (eps_rel)
    Qp.settings.eps_rel = *eps_rel;
  if (verbose)
    Qp.settings.verbose = *verbose;
  if (max_iter)
    Qp.settings.max_iter = *max_iter;
  if (eps_duality_gap_abs)
    Qp.settings.eps_duality_gap_abs = *eps_duality_gap_abs;
  if (eps_duality_gap_rel)
    Qp.settings.eps_duality_gap_rel = *eps_duality_gap_rel;
  Qp.settings.compute_timings = compute_timings;
  Qp.settings.primal_infeasibility_solving = primal_infeasibility_solving;
  Qp.init(H, g, A, b, C, l, u, compute_preconditioner, rho, mu_eq, mu_in, manual_minimal_H_eigenvalue);
  Qp.solve(x, y, z);
  return Qp.results;
}

// One-shot solve with box constraints (reference wrapper.hpp:1133-1236)
template<typename T>
Results<T>
solve(optional<MatRef<T>> H, optional<VecRef<T>> g, optional<MatRef<T>> A, optional<VecRef<T>> b,
      optional<MatRef<T>> C, optional<VecRef<T>> l, optional<VecRef<T>> u, optional<VecRef<T>> l_box,
      optional<VecRef<T>> u_box, optional<VecRef<T>> x = nullopt, optional<VecRef<T>> y = nullopt,
      optional<VecRef<T>> z = nullopt, optional<T> eps_abs = nullopt, optional<T> eps_rel = nullopt,
      optional<T> rho = nullopt, optional<T> mu_eq = nullopt, optional<T> mu_in = nullopt,
      optional<bool> verbose = nullopt, bool compute_preconditioner = true, bool compute_timings = false,
      optional<isize> max_iter = nullopt,
      InitialGuessStatus initial_guess = InitialGuessStatus::EQUALITY_CONSTRAINED_INITIAL_GUESS,
      bool check_duality_gap = false, optional<T> eps_duality_gap_abs = nullopt,
      optional<T> eps_duality_gap_rel = nullopt, bool primal_infeasibility_solving = false,
      optional<T> manual_minimal_H_eigenvalue = nullopt)
{
  isize n = H ? H->rows() : 0, n_eq = A ? A->rows() : 0, n_in = C ? C->rows() : 0;
  QP<T> Qp(n, n_eq, n_in, true, DenseBackend::PrimalDualLDLT);
  Qp.settings.initial_guess = initial_guess;
  Qp.settings.check_duality_gap = check_duality_gap;
  if (eps_abs)
    Qp.settings.eps_abs = *eps_abs;
  if (eps_rel)
    Qp.settings.eps_rel = *eps_rel;
  if (verbose)
    Qp.settings.verbose = *verbose;
  if (max_iter)
    Qp.settings.max_iter = *max_iter;
  if (eps_duality_gap_abs)
    Qp.settings.eps_duality_gap_abs = *eps_duality_gap_abs;
  if (eps_duality_gap_rel)
    Qp.settings.eps_duality_gap_rel = *eps_duality_gap_rel;
  Qp.settings.compute_timings = compute_timings;
  Qp.settings.primal_infeasibility_solving = primal_infeasibility_solving;
  Qp.init(H, g, A, b, C, l, u, l_box, u_box, compute_preconditioner, rho, mu_eq, mu_in,
          manual_minimal_H_eigenvalue);
  Qp.solve(x, y, z);
  return Qp.results;
}

// dense::BatchQP<T> (reference wrapper.hpp:1253-1311).  `batch_size` is the capacity of each
// device pool (the reference reserves a std::vector of that many QPs); QPs of different sizes
// get different pools.  References returned by init_qp_in_place stay valid for the life of the
// BatchQP (the reference only guarantees it up to `batch_size` QPs).
template<typename T>
struct BatchQP
{
  // `device`: a HIP device ordinal, or all_devices (the default): with more than one device visible the batch is
  // spread over all of them -- every signature gets a multi-device batch (pqp_multi_create) of `batch_size` QPs whose
  // shards hold contiguous ranges, and solve_in_parallel launches every shard before it waits for any (the
  // reference's solve_in_parallel uses every core of the host, parallel/qp_solve.hpp:41-59).  With one device
  // visible this is BatchQP(batch_size, 0).
  static constexpr int all_devices = -1;
  explicit BatchQP(usize batch_size = 0, int device = all_devices)
    : capacity_(isize(batch_size) > 0 ? isize(batch_size) : 1)
    , device_(device)
  {
    if (device_ == all_devices) {
      const int nd = pqp_device_count();
      if (nd > 1)
        for (int g = 0; g < nd; ++g)
          devices_.push_back(g);
      else
        device_ = 0;
    }
  }
  // the batch spread over exactly these devices (an ordinal may repeat: logical shards on one GPU)
  BatchQP(usize batch_size, const std::vector<int>& devices)
    : capacity_(isize(batch_size) > 0 ? isize(batch_size) : 1)
    , device_(devices.empty() ? 0 : devices[0])
    , devices_(devices.size() > 1 ? devices : std::vector<int>())
  {
  }

  QP<T>& init_qp_in_place(isize dim, isize n_eq, isize n_in) { return emplace(dim, n_eq, n_in, false, HessianType::Dense, DenseBackend::Automatic); }
  QP<T>& init_qp_in_place(isize dim, isize n_eq, isize n_in, bool box, HessianType hessian = HessianType::Dense,
                          DenseBackend backend = DenseBackend::Automatic)
  {
    return emplace(dim, n_eq, n_in, box, hessian, backend);
  }

  // Copies `qp` (model, settings) into a new slot.  The reference's insert forgets to bump
  // m_size (wrapper.hpp:1288), hiding the QP from size() and solve_in_parallel; here it counts.
  void insert(const QP<T>& qp)
  {
    QP<T>& q = emplace(qp.model.dim, qp.model.n_eq, qp.model.n_in, qp.is_box_constrained(),
                       qp.which_hessian_type(), qp.which_dense_backend());
    q.settings = qp.settings;
    const Model<T>& m = qp.model;
    if (qp.is_box_constrained())
      q.init(m.H, m.g, m.A, m.b, m.C, m.l, m.u, m.l_box, m.u_box, qp.settings.compute_preconditioner);
    else
      q.init(m.H, m.g, m.A, m.b, m.C, m.l, m.u, qp.settings.compute_preconditioner);
  }

  // (the QPs of a BatchQP are views of its pools: the container is movable, not copyable)
  BatchQP(const BatchQP&) = delete;
  BatchQP& operator=(const BatchQP&) = delete;
  BatchQP(BatchQP&&) = default;
  BatchQP& operator=(BatchQP&&) = default;

  QP<T>& get(isize i) { return qps_[usize(i)]; }
  const QP<T>& get(isize i) const { return qps_[usize(i)]; }
  QP<T>& operator[](isize i) { return qps_[usize(i)]; }
  const QP<T>& operator[](isize i) const { return qps_[usize(i)]; }
  isize size() const { return isize(qps_.size()); }

  // pools in creation order, each with the indices (into this BatchQP) of the QPs it holds
  struct PoolEntry
  {
    std::shared_ptr<detail::Pool> pool;
    std::vector<isize> members; // members[slot] = index of the QP in that slot
  };
  const std::vector<PoolEntry>& pools() const { return pools_; }
  const std::vector<int>& devices() const { return devices_; } // empty: one device (device())
  int device() const { return device_; }

private:
  using Key = std::tuple<isize, isize, isize, bool, int, int>;
  QP<T>& emplace(isize dim, isize n_eq, isize n_in, bool box, HessianType hessian, DenseBackend backend)
  {
    if (dim <= 0)
      throw std::invalid_argument(
        "wrong argument size: the dimension wrt the primal variable x should be strictly positive.");
    Key key{ dim, n_eq, n_in, box, int(hessian), int(backend) };
    auto it = open_.find(key);
    // (the shards of a multi-device batch are consecutive pools: a full shard hands over to the next one)
    while (it != open_.end() && pools_[it->second].pool->used == pools_[it->second].pool->capacity &&
           pools_[it->second].pool->multi && it->second + 1 < pools_.size() &&
           pools_[it->second + 1].pool->multi == pools_[it->second].pool->multi)
      ++it->second;
    if (it == open_.end() || pools_[it->second].pool->used == pools_[it->second].pool->capacity) {
      const usize at = pools_.size();
      if (devices_.empty()) {
        pools_.push_back({ std::make_shared<detail::Pool>(capacity_, dim, n_eq, n_in, box, hessian, backend, device_), {} });
      } else {
        pqp_multi* m = nullptr;
        detail::check(pqp_multi_create(capacity_, dim, n_eq, n_in, box ? 1 : 0, int(hessian), int(backend), devices_.data(),
                                       int(devices_.size()), &m));
        auto owner = std::make_shared<detail::MultiOwner>(m);
        for (int g = 0; g < pqp_multi_shard_count(m); ++g) {
          auto sp = std::make_shared<detail::Pool>(owner, g, dim, n_eq, n_in, box);
          if (sp->capacity > 0)
            pools_.push_back({ sp, {} });
        }
      }
      open_[key] = at;
      it = open_.find(key);
    }
    PoolEntry& e = pools_[it->second];
    const isize slot = e.pool->used++;
    e.members.push_back(isize(qps_.size()));
    qps_.push_back(QP<T>(dim, n_eq, n_in, box, hessian, backend, e.pool, slot, e.pool->multi ? devices_[usize(e.pool->shard)] : device_));
    return qps_.back();
  }

  isize capacity_;
  int device_;
  std::vector<int> devices_; // more than one entry: every signature lives in a multi-device batch over these
  std::deque<QP<T>> qps_;
  std::vector<PoolEntry> pools_;
  std::map<Key, usize> open_;
};

} // namespace dense
} // namespace proxqp
} // namespace proxsuite

#endif
