// proxsuite/proxqp/parallel/qp_solve.hpp -- dense::solve_in_parallel on MI355X.
//
// The reference runs `qp.solve()` for every QP under `#pragma omp parallel for
// schedule(dynamic)` (include/proxsuite/proxqp/parallel/qp_solve.hpp:17-59).  Here the unit of
// parallelism is one GPU workgroup per QP: a BatchQP is solved with ONE kernel launch per
// device pool (pqp_batch_solve_range_async, include/proxqp_hip.h): all pools -- on one device or, for a BatchQP
// spread over the node's GPUs, on several -- are launched before any is waited for, and the results are read from
// the pools' pinned host mirrors, which the solve kernel itself wrote.  `num_threads` is accepted for source
// compatibility and ignored.
#ifndef PROXSUITE_AMD_PROXQP_PARALLEL_QPSOLVE_HPP
#define PROXSUITE_AMD_PROXQP_PARALLEL_QPSOLVE_HPP

#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "proxsuite/proxqp/dense/compute_ECJ.hpp"

namespace proxsuite {
namespace proxqp {
namespace dense {

template<typename T>
void
solve_in_parallel(BatchQP<T>& qps, const optional<usize> /*num_threads*/ = nullopt)
{
  // every pool is launched before any of them is waited for (pqp_batch_solve_range_async: the pools have streams of
  // their own, the shards of a multi-device batch their own devices); the results are then in the pools' host
  // mirrors -- the solve kernel wrote them -- and only the slots that were solved are scattered
  std::vector<std::unique_lock<std::recursive_mutex>> locks; // (another thread may be driving a standalone QP of a pool)
  std::vector<const typename BatchQP<T>::PoolEntry*> launched;
  for (const auto& e : qps.pools()) {
    const detail::Pool& p = *e.pool;
    if (p.used == 0)
      continue;
    locks.emplace_back(p.mtx);
    for (isize idx : e.members)
      qps[idx].push_settings();
    detail::check(pqp_batch_solve_range_async(p.h, 0, p.used));
    launched.push_back(&e);
  }
  for (const auto* e : launched)
    detail::check(pqp_batch_wait(e->pool->h));
  for (const auto* e : launched)
    for (isize idx : e->members)
      qps[idx].pull_from_mirrors();
}

// std::vector of QPs (reference qp_solve.hpp:17-39, the most common calling form): standalone QPs
// of one signature share device pools (dense/wrapper.hpp, detail::Registry), so the QPs of the
// vector are grouped by pool and every group is ONE launch -- pqp_batch_solve_subset_async, workgroup i
// solves slot idx[i]; all groups are launched, then waited for, then read from the pools' host mirrors.
template<typename T>
void
solve_in_parallel(std::vector<QP<T>>& qps, const optional<usize> /*num_threads*/ = nullopt)
{
  std::map<const detail::Pool*, std::vector<usize>> groups;
  for (usize i = 0; i < qps.size(); ++i)
    groups[qps[i].pool().get()].push_back(i);
  std::vector<std::unique_lock<std::recursive_mutex>> locks;
  for (auto& kv : groups) {
    const detail::Pool& p = *kv.first;
    locks.emplace_back(p.mtx);
    std::vector<int64_t> idx;
    idx.reserve(kv.second.size());
    for (usize i : kv.second) {
      qps[i].push_settings();
      idx.push_back(int64_t(qps[i].slot()));
    }
    detail::check(pqp_batch_solve_subset_async(p.h, idx.data(), int64_t(idx.size())));
  }
  for (auto& kv : groups)
    detail::check(pqp_batch_wait(kv.first->h));
  for (auto& kv : groups)
    for (usize i : kv.second)
      qps[i].pull_from_mirrors();
}

// dense::qp_solve_backward_in_parallel (reference parallel/qp_solve.hpp:83-137): compute_backward
// for every QP of the batch -- one pqp_batch_backward_range launch per device pool.
template<typename T>
void
qp_solve_backward_in_parallel(optional<const usize> /*num_threads*/, BatchQP<T>& qps,
                              std::vector<Vec<T>>& loss_derivatives, T eps = 1.E-4, T rho_new = 1.E-6,
                              T mu_new = 1.E-6)
{
  if (isize(loss_derivatives.size()) != qps.size())
    throw std::invalid_argument("wrong argument size: one loss derivative per QP is expected");
  for (const auto& e : qps.pools()) {
    const detail::Pool& p = *e.pool;
    if (p.used == 0)
      continue;
    detail::PoolLock lock(p.mtx);
    const usize ntot = usize(p.dim + p.n_eq + p.n_in);
    std::vector<T> ld(usize(p.used) * ntot);
    for (usize s = 0; s < e.members.size(); ++s) {
      const Vec<T>& v = loss_derivatives[usize(e.members[s])];
      if (usize(v.size()) != ntot)
        detail::bad_size("the loss derivative has dim + n_eq + n_in entries.", v.size(), isize(ntot));
      std::memcpy(ld.data() + s * ntot, v.data(), ntot * sizeof(T));
      qps[e.members[s]].push_settings();
    }
    detail::check(pqp_batch_backward_range(p.h, 0, p.used, ld.data(), eps, rho_new, mu_new));
    for (usize s = 0; s < e.members.size(); ++s) {
      detail::pull_backward(qps[e.members[s]]);
      qps[e.members[s]].pull();
    }
  }
}

template<typename T>
void
qp_solve_backward_in_parallel(optional<const usize> num_threads, std::vector<QP<T>>& qps,
                              std::vector<Vec<T>>& loss_derivatives, T eps = 1.E-4, T rho_new = 1.E-6,
                              T mu_new = 1.E-6)
{
  (void)num_threads;
  if (loss_derivatives.size() != qps.size())
    throw std::invalid_argument("wrong argument size: one loss derivative per QP is expected");
  // grouped by pool like the forward form: ONE pqp_batch_backward_subset launch per pool
  std::map<const detail::Pool*, std::vector<usize>> groups;
  for (usize i = 0; i < qps.size(); ++i)
    groups[qps[i].pool().get()].push_back(i);
  for (auto& kv : groups) {
    const detail::Pool& p = *kv.first;
    detail::PoolLock lock(p.mtx);
    const usize ntot = usize(p.dim + p.n_eq + p.n_in);
    std::vector<T> ld(kv.second.size() * ntot);
    std::vector<int64_t> idx;
    idx.reserve(kv.second.size());
    for (usize k = 0; k < kv.second.size(); ++k) {
      const usize i = kv.second[k];
      const Vec<T>& v = loss_derivatives[i];
      if (usize(v.size()) != ntot)
        detail::bad_size("the loss derivative has dim + n_eq + n_in entries.", v.size(), isize(ntot));
      std::memcpy(ld.data() + k * ntot, v.data(), ntot * sizeof(T));
      qps[i].push_settings();
      idx.push_back(int64_t(qps[i].slot()));
    }
    detail::check(pqp_batch_backward_subset(p.h, idx.data(), int64_t(idx.size()), ld.data(), eps, rho_new, mu_new));
    for (usize i : kv.second) {
      detail::pull_backward(qps[i]);
      qps[i].pull();
    }
  }
}

} // namespace dense
} // namespace proxqp
} // namespace proxsuite

#endif
