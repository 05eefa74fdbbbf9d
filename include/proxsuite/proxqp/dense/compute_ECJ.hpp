// proxsuite/proxqp/dense/compute_ECJ.hpp -- dense::compute_backward on MI355X: derivatives of a
// loss wrt (H, g, A, b, C, u, l) of a SOLVED QP given dL/d(x, y, z), written to
// qp.model.backward_data.  Same signature and defaults as the reference
// (include/proxsuite/proxqp/dense/compute_ECJ.hpp:29-132); the work is one launch of
// pqp_backward_kernel through pqp_batch_backward_range (include/proxqp_hip.h).
#ifndef PROXSUITE_AMD_PROXQP_DENSE_COMPUTE_ECJ_HPP
#define PROXSUITE_AMD_PROXQP_DENSE_COMPUTE_ECJ_HPP

#include "proxsuite/proxqp/dense/wrapper.hpp"

namespace proxsuite {
namespace proxqp {
namespace dense {

namespace detail {
template<typename T>
inline void
pull_backward(QP<T>& qp)
{
  BackwardData<T>& bd = qp.model.backward_data;
  bd.initialize(qp.model.dim, qp.model.n_eq, qp.model.n_in);
  check(pqp_batch_get_backward(qp.pool()->h, qp.slot(), bd.dL_dH.data(), bd.dL_dg.data(), bd.dL_dA.data(),
                               bd.dL_db.data(), bd.dL_dC.data(), bd.dL_du.data(), bd.dL_dl.data()));
}
} // namespace detail

template<typename T>
void
compute_backward(QP<T>& solved_qp, VecRef<T> loss_derivative, T eps = 1.E-4, T rho_new = 1.E-6, T mu_new = 1.E-6)
{
  const isize ntot = solved_qp.model.dim + solved_qp.model.n_eq + solved_qp.model.n_in;
  if (loss_derivative.size() != ntot)
    detail::bad_size("the loss derivative has dim + n_eq + n_in entries.", loss_derivative.size(), ntot);
  std::vector<T> tmp;
  const T* p = loss_derivative.ptr;
  if (loss_derivative.stride != 1) {
    tmp.resize(usize(ntot));
    for (isize i = 0; i < ntot; ++i)
      tmp[usize(i)] = loss_derivative[i];
    p = tmp.data();
  }
  detail::PoolLock lock(solved_qp.pool()->mtx); // (the pool's handle is shared with the other QPs of the pool)
  solved_qp.push_settings();
  detail::check(pqp_batch_backward_range(solved_qp.pool()->h, solved_qp.slot(), 1, p, eps, rho_new, mu_new));
  detail::pull_backward(solved_qp);
  solved_qp.pull(); // results.info carries the backward proximal parameters, as in the reference
}

} // namespace dense
} // namespace proxqp
} // namespace proxsuite

#endif
