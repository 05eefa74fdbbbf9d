// proxsuite/proxqp/results.hpp -- Info<T> / Results<T> of the dense ProxQP API, MI355X build
// (reference include/proxsuite/proxqp/results.hpp:27-74 and :76-204).  Results are host copies
// refreshed from the device after every init / update / solve / cleanup of the owning QP.
#ifndef PROXSUITE_AMD_PROXQP_RESULTS_HPP
#define PROXSUITE_AMD_PROXQP_RESULTS_HPP

#include "proxsuite/proxqp/dense/views.hpp"

namespace proxsuite {
namespace proxqp {

template<typename T>
struct Info
{
  T mu_eq = 1e-3, mu_eq_inv = 1e3, mu_in = 1e-1, mu_in_inv = 1e1, rho = 1e-6, nu = 1;
  isize iter = 0, iter_ext = 0, mu_updates = 0, rho_updates = 0;
  QPSolverOutput status = QPSolverOutput::PROXQP_NOT_RUN;
  T setup_time = 0, solve_time = 0, run_time = 0;
  T objValue = 0, pri_res = 0, dua_res = 0, duality_gap = 0, iterative_residual = 0;
  T minimal_H_eigenvalue_estimate = 0;

  void from_c(const pqp_info& i)
  {
    mu_eq = i.mu_eq;
    mu_eq_inv = i.mu_eq_inv;
    mu_in = i.mu_in;
    mu_in_inv = i.mu_in_inv;
    rho = i.rho;
    nu = i.nu;
    iter = isize(i.iter);
    iter_ext = isize(i.iter_ext);
    mu_updates = isize(i.mu_updates);
    rho_updates = isize(i.rho_updates);
    status = QPSolverOutput(i.status);
    setup_time = i.setup_time;
    solve_time = i.solve_time;
    run_time = i.run_time;
    objValue = i.objValue;
    pri_res = i.pri_res;
    dua_res = i.dua_res;
    duality_gap = i.duality_gap;
    iterative_residual = i.iterative_residual;
    minimal_H_eigenvalue_estimate = i.minimal_H_eigenvalue_estimate;
  }
};

template<typename T>
struct Results
{
  dense::Vec<T> x, y, z, se, si;
  Info<T> info;

  Results() = default;
  // reference results.hpp:90-131: z (and si) carry n_in (+ dim with box constraints) entries
  Results(isize dim, isize n_eq, isize n_in, bool box_constraints = false)
    : x(dim)
    , y(n_eq)
    , z(n_in + (box_constraints ? dim : 0))
    , se(n_eq)
    , si(n_in + (box_constraints ? dim : 0))
  {
  }
};

} // namespace proxqp
} // namespace proxsuite

#endif
