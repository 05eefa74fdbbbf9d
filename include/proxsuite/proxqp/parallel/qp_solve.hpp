// proxsuite/proxqp/parallel/qp_solve.hpp -- dense::solve_in_parallel on MI355X.
//
// The reference runs `qp.solve()` for every QP under `#pragma omp parallel for
// schedule(dynamic)` (include/proxsuite/proxqp/parallel/qp_solve.hpp:17-59).  Here the unit of
// parallelism is one GPU workgroup per QP: a BatchQP is solved with ONE kernel launch per
// device pool (pqp_batch_solve_range, include/proxqp_hip.h), followed by one bulk copy of the
// results.  `num_threads` is accepted for source compatibility and ignored.
#ifndef PROXSUITE_AMD_PROXQP_PARALLEL_QPSOLVE_HPP
#define PROXSUITE_AMD_PROXQP_PARALLEL_QPSOLVE_HPP

#include <cstring>
#include <map>
#include <vector>

#include "proxsuite/proxqp/dense/compute_ECJ.hpp"

namespace proxsuite {
namespace proxqp {
namespace dense {

template<typename T>
void
solve_in_parallel(BatchQP<T>& qps, const optional<usize> /*num_threads*/ = nullopt)
{
  for (const auto& e : qps.pools()) {
    const detail::Pool& p = *e.pool;
    if (p.used == 0)
      continue;
    detail::PoolLock lock(p.mtx); // (another thread may be driving a standalone QP of this pool)
    for (isize idx : e.members)
      qps[idx].push_settings();
    detail::check(pqp_batch_solve_range(p.h, 0, p.used));
    // one device-to-host copy per array for the whole pool, then scatter
    const usize B = usize(p.capacity);
    std::vector<T> x(B * usize(p.dim)), y(B * usize(p.n_eq)), z(B * usize(p.n_c)), se(B * usize(p.n_eq)),
      si(B * usize(p.n_c));
    std::vector<pqp_info> info(B);
    detail::check(pqp_batch_get_results(p.h, -1, x.data(), y.data(), z.data(), se.data(), si.data(), info.data()));
    for (usize s = 0; s < e.members.size(); ++s) {
      QP<T>& q = qps[e.members[s]];
      auto put = [s](Vec<T>& dst, const std::vector<T>& src) {
        if (dst.size())
          std::memcpy(dst.data(), src.data() + s * usize(dst.size()), usize(dst.size()) * sizeof(T));
      };
      put(q.results.x, x);
      put(q.results.y, y);
      put(q.results.z, z);
      put(q.results.se, se);
      put(q.results.si, si);
      q.results.info.from_c(info[s]);
      q.pull_settings();
    }
  }
}

// std::vector of QPs (reference qp_solve.hpp:17-39, the most common calling form): standalone QPs
// of one signature share device pools (dense/wrapper.hpp, detail::Registry), so the QPs of the
// vector are grouped by pool and every group is ONE launch -- pqp_batch_solve_subset, workgroup i
// solves slot idx[i] -- followed by one bulk copy of that pool's results.
template<typename T>
void
solve_in_parallel(std::vector<QP<T>>& qps, const optional<usize> /*num_threads*/ = nullopt)
{
  std::map<const detail::Pool*, std::vector<usize>> groups;
  for (usize i = 0; i < qps.size(); ++i)
    groups[qps[i].pool().get()].push_back(i);
  for (auto& kv : groups) {
    const detail::Pool& p = *kv.first;
    detail::PoolLock lock(p.mtx);
    std::vector<int64_t> idx;
    idx.reserve(kv.second.size());
    for (usize i : kv.second) {
      qps[i].push_settings();
      idx.push_back(int64_t(qps[i].slot()));
    }
    detail::check(pqp_batch_solve_subset(p.h, idx.data(), int64_t(idx.size())));
    const usize B = usize(p.capacity);
    std::vector<T> x(B * usize(p.dim)), y(B * usize(p.n_eq)), z(B * usize(p.n_c)), se(B * usize(p.n_eq)),
      si(B * usize(p.n_c));
    std::vector<pqp_info> info(B);
    detail::check(pqp_batch_get_results(p.h, -1, x.data(), y.data(), z.data(), se.data(), si.data(), info.data()));
    for (usize i : kv.second) {
      QP<T>& q = qps[i];
      const usize s = usize(q.slot());
      auto put = [s](Vec<T>& dst, const std::vector<T>& src) {
        if (dst.size())
          std::memcpy(dst.data(), src.data() + s * usize(dst.size()), usize(dst.size()) * sizeof(T));
      };
      put(q.results.x, x);
      put(q.results.y, y);
      put(q.results.z, z);
      put(q.results.se, se);
      put(q.results.si, si);
      q.results.info.from_c(info[s]);
      q.pull_settings();
    }
  }
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
  for (usize i = 0; i < qps.size(); ++i)
    compute_backward<T>(qps[i], loss_derivatives[i], eps, rho_new, mu_new);
}

} // namespace dense
} // namespace proxqp
} // namespace proxsuite

#endif
