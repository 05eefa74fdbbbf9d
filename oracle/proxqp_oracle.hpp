// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or executed from
// the product path (proxsuite_amd/, include/proxsuite/, libproxqp_hip.so).
//
// CPU restatement (plain fp64 loops, no Eigen) of the reference's dense ProxQP
// solver, PrimalDualLDLT backend:
//   reference include/proxsuite/proxqp/dense/wrapper.hpp   (QP state machine)
//   reference include/proxsuite/proxqp/dense/helpers.hpp   (setup, update, ...)
//   reference include/proxsuite/proxqp/dense/preconditioner/ruiz.hpp
//   reference include/proxsuite/proxqp/dense/solver.hpp    (qp_solve & friends)
//   reference include/proxsuite/proxqp/dense/linesearch.hpp
//   reference include/proxsuite/proxqp/dense/utils.hpp     (global residuals)
//   reference include/proxsuite/proxqp/dense/workspace.hpp / results.hpp
// Member and local names follow the reference so the two can be read side by
// side.  Matrices are row-major (reference dense/fwd.hpp:16-19).
//
// PARITY: Eigen is absent from the authoring container, so this restatement
// cannot be diffed against the reference binary -> "parity unpinned" for random
// QPs; it is pinned on the reference's known answers and fixtures (see
// oracle/README.md and tests/test_oracle_*.py).
#ifndef PQP_ORACLE_HPP
#define PQP_ORACLE_HPP

#include "../include/pqp_types.h"
#include "ldlt_oracle.hpp"

#include <limits>
#include <stdexcept>
#include <string>

namespace pqo {

using Vec = std::vector<double>;

inline double
infty_norm(const double* v, isize n)
{
  double r = 0;
  for (isize i = 0; i < n; ++i) {
    double a = std::fabs(v[i]);
    if (a > r)
      r = a;
  }
  return r;
}
inline double
dot(const double* a, const double* b, isize n)
{
  double r = 0;
  for (isize i = 0; i < n; ++i)
    r += a[i] * b[i];
  return r;
}
// reference include/proxsuite/helpers/common.hpp:17-25
inline double
infinite_bound()
{
  return std::sqrt(std::numeric_limits<double>::max());
}

struct Model
{
  isize dim = 0, n_eq = 0, n_in = 0;
  Vec H, g, A, b, C, u, l, u_box, l_box;
};

struct Results
{
  Vec x, y, z, se, si;
  pqp_info info;
};

// reference dense/backward_data.hpp:27-133: jacobians of the loss wrt the model (row-major)
struct BackwardData
{
  Vec dL_dH, dL_dg, dL_dA, dL_db, dL_dC, dL_du, dL_dl;
};

// reference preconditioner/ruiz.hpp:316-358
struct Ruiz
{
  Vec delta;
  double c = 1;
  isize dim = 0, n_eq = 0, n_in = 0;
};

// reference workspace.hpp:24-100
struct Workspace
{
  Ldlt ldl;
  Vec H_scaled, g_scaled, A_scaled, C_scaled, b_scaled, u_scaled, l_scaled;
  Vec u_box_scaled, l_box_scaled, i_scaled;
  Vec x_prev, y_prev, z_prev;
  Vec kkt;
  std::vector<isize> current_bijection_map, new_bijection_map;
  std::vector<char> active_set_up, active_set_low, active_inequalities;
  Vec Hdx, Cdx, Adx, active_part_z;
  Vec alphas;
  Vec dw_aug, rhs, err;
  double dual_feasibility_rhs_2 = 0, correction_guess_rhs_g = 0, correction_guess_rhs_b = 0;
  double alpha = 1;
  Vec dual_residual_scaled, primal_residual_in_scaled_up;
  Vec primal_residual_in_scaled_up_plus_alphaCdx, primal_residual_in_scaled_low_plus_alphaCdx;
  Vec CTz;
  bool constraints_changed = false, dirty = false, refactorize = false;
  bool proximal_parameter_update = false, is_initialized = false;
  isize n_c = 0;
  Vec solve_work;
};

struct QP
{
  int dense_backend;
  bool box_constraints;
  int hessian_type;
  pqp_settings settings;
  Results results;
  Model model;
  Workspace work;
  Ruiz ruiz;
  OpCounters counters;
  // extra event counters
  isize n_solves = 0, n_residuals = 0, n_ls_evals = 0, n_inserted = 0, n_deleted = 0,
        n_refactorize = 0;

  // settings.verbose: called once per outer iteration with the unscaled-iterate statistics the reference
  // prints (solver.hpp:1469-1499); null = the round trip of the iterates happens, nothing is printed
  void (*verbose_sink)(long long outer_iteration, const pqp_info& info) = nullptr;
  // settings.verbose: the lines the reference prints per iteration, as records of 8 doubles
  //   outer (solver.hpp:1478-1485): { 1, iter + 1, pri_res, dua_res, duality_gap, mu_in, rho, 0 }
  //   inner (solver.hpp:1021-1027): { 2, iter + 1, inner residual, alpha, 0, 0, 0, 0 }
  // cleared at the start of every solve (test infrastructure: the device's per-iteration trace is compared with it)
  std::vector<double> trace;

  QP(isize dim, isize n_eq, isize n_in, bool box, int hessian, int backend);

  // wrapper.hpp:354-498 / 520-703.  Null pointer == nullopt; NaN == nullopt.
  void init(const double* H, const double* g, const double* A, const double* b,
            const double* C, const double* l, const double* u, const double* l_box,
            const double* u_box, bool compute_preconditioner, double rho, double mu_eq,
            double mu_in, double manual_minimal_H_eigenvalue);
  // wrapper.hpp:723-807 / 831-918
  void update(const double* H, const double* g, const double* A, const double* b,
              const double* C, const double* l, const double* u, const double* l_box,
              const double* u_box, bool update_preconditioner, double rho, double mu_eq,
              double mu_in, double manual_minimal_H_eigenvalue);
  // wrapper.hpp:922-957
  void solve(const double* x, const double* y, const double* z);
  // wrapper.hpp:958-962
  void cleanup();
  // dense/compute_ECJ.hpp:29-189 (compute_backward + compute_backward_loss_ESG): derivatives of a
  // loss wrt (H, g, A, b, C, u, l) given dL/d(x, y, z) at the solution of a solved QP
  BackwardData backward_data;
  void compute_backward(const double* loss_derivative, double eps, double rho_new, double mu_new);

  isize n_constraints() const { return model.n_in + (box_constraints ? model.dim : 0); }
};

} // namespace pqo

#endif
