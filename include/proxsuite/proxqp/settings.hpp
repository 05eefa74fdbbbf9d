// proxsuite/proxqp/settings.hpp -- Settings<T> of the dense ProxQP API, MI355X build.
// Member names, types and defaults follow the reference (include/proxsuite/proxqp/
// settings.hpp:28-47 for the enums, :95-315 for the struct); the record that travels to the
// device is the POD pqp_settings of include/pqp_types.h, converted by to_c() / from_c().
#ifndef PROXSUITE_AMD_PROXQP_SETTINGS_HPP
#define PROXSUITE_AMD_PROXQP_SETTINGS_HPP

#include <cstddef>
#include <optional>

#include "proxsuite/proxqp/status.hpp"

namespace proxsuite {
namespace proxqp {

using isize = std::ptrdiff_t;
using usize = std::size_t;
template<typename T>
using optional = std::optional<T>;
using std::nullopt;

enum struct DenseBackend
{
  Automatic = PQP_BACKEND_AUTOMATIC,
  PrimalDualLDLT = PQP_BACKEND_PRIMAL_DUAL_LDLT,
  PrimalLDLT = PQP_BACKEND_PRIMAL_LDLT
};
enum struct HessianType
{
  Zero = PQP_HESSIAN_ZERO,
  Dense = PQP_HESSIAN_DENSE,
  Diagonal = PQP_HESSIAN_DIAGONAL
};
enum struct MeritFunctionType
{
  GPDAL = PQP_MERIT_GPDAL,
  PDAL = PQP_MERIT_PDAL
};

template<typename T>
struct Settings
{
  T default_rho, default_mu_eq, default_mu_in;
  T alpha_bcl, beta_bcl;
  T refactor_dual_feasibility_threshold, refactor_rho_threshold;
  T mu_min_eq, mu_min_in, mu_max_eq_inv, mu_max_in_inv;
  T mu_update_factor, mu_update_inv_factor;
  T cold_reset_mu_eq, cold_reset_mu_in, cold_reset_mu_eq_inv, cold_reset_mu_in_inv;
  T eps_abs, eps_rel;
  isize max_iter, max_iter_in, safe_guard, nb_iterative_refinement;
  T eps_refact;
  bool verbose;
  InitialGuessStatus initial_guess;
  bool update_preconditioner, compute_preconditioner, compute_timings;
  bool check_duality_gap;
  T eps_duality_gap_abs, eps_duality_gap_rel;
  isize preconditioner_max_iter;
  T preconditioner_accuracy;
  T eps_primal_inf, eps_dual_inf;
  bool bcl_update;
  MeritFunctionType merit_function_type;
  T alpha_gpdal;
  bool primal_infeasibility_solving;
  isize frequence_infeasibility_check;
  T default_H_eigenvalue_estimate;

  explicit Settings(DenseBackend dense_backend = DenseBackend::PrimalDualLDLT)
  {
    pqp_settings s;
    pqp_settings_default(&s, int(dense_backend));
    from_c(s);
  }

  void from_c(const pqp_settings& s)
  {
    default_rho = s.default_rho;
    default_mu_eq = s.default_mu_eq;
    default_mu_in = s.default_mu_in;
    alpha_bcl = s.alpha_bcl;
    beta_bcl = s.beta_bcl;
    refactor_dual_feasibility_threshold = s.refactor_dual_feasibility_threshold;
    refactor_rho_threshold = s.refactor_rho_threshold;
    mu_min_eq = s.mu_min_eq;
    mu_min_in = s.mu_min_in;
    mu_max_eq_inv = s.mu_max_eq_inv;
    mu_max_in_inv = s.mu_max_in_inv;
    mu_update_factor = s.mu_update_factor;
    mu_update_inv_factor = s.mu_update_inv_factor;
    cold_reset_mu_eq = s.cold_reset_mu_eq;
    cold_reset_mu_in = s.cold_reset_mu_in;
    cold_reset_mu_eq_inv = s.cold_reset_mu_eq_inv;
    cold_reset_mu_in_inv = s.cold_reset_mu_in_inv;
    eps_abs = s.eps_abs;
    eps_rel = s.eps_rel;
    max_iter = isize(s.max_iter);
    max_iter_in = isize(s.max_iter_in);
    safe_guard = isize(s.safe_guard);
    nb_iterative_refinement = isize(s.nb_iterative_refinement);
    eps_refact = s.eps_refact;
    verbose = s.verbose != 0;
    initial_guess = InitialGuessStatus(s.initial_guess);
    update_preconditioner = s.update_preconditioner != 0;
    compute_preconditioner = s.compute_preconditioner != 0;
    compute_timings = s.compute_timings != 0;
    check_duality_gap = s.check_duality_gap != 0;
    eps_duality_gap_abs = s.eps_duality_gap_abs;
    eps_duality_gap_rel = s.eps_duality_gap_rel;
    preconditioner_max_iter = isize(s.preconditioner_max_iter);
    preconditioner_accuracy = s.preconditioner_accuracy;
    eps_primal_inf = s.eps_primal_inf;
    eps_dual_inf = s.eps_dual_inf;
    bcl_update = s.bcl_update != 0;
    merit_function_type = MeritFunctionType(s.merit_function_type);
    alpha_gpdal = s.alpha_gpdal;
    primal_infeasibility_solving = s.primal_infeasibility_solving != 0;
    frequence_infeasibility_check = isize(s.frequence_infeasibility_check);
    default_H_eigenvalue_estimate = s.default_H_eigenvalue_estimate;
  }

  void to_c(pqp_settings& s) const
  {
    s.default_rho = default_rho;
    s.default_mu_eq = default_mu_eq;
    s.default_mu_in = default_mu_in;
    s.alpha_bcl = alpha_bcl;
    s.beta_bcl = beta_bcl;
    s.refactor_dual_feasibility_threshold = refactor_dual_feasibility_threshold;
    s.refactor_rho_threshold = refactor_rho_threshold;
    s.mu_min_eq = mu_min_eq;
    s.mu_min_in = mu_min_in;
    s.mu_max_eq_inv = mu_max_eq_inv;
    s.mu_max_in_inv = mu_max_in_inv;
    s.mu_update_factor = mu_update_factor;
    s.mu_update_inv_factor = mu_update_inv_factor;
    s.cold_reset_mu_eq = cold_reset_mu_eq;
    s.cold_reset_mu_in = cold_reset_mu_in;
    s.cold_reset_mu_eq_inv = cold_reset_mu_eq_inv;
    s.cold_reset_mu_in_inv = cold_reset_mu_in_inv;
    s.eps_abs = eps_abs;
    s.eps_rel = eps_rel;
    s.max_iter = max_iter;
    s.max_iter_in = max_iter_in;
    s.safe_guard = safe_guard;
    s.nb_iterative_refinement = nb_iterative_refinement;
    s.eps_refact = eps_refact;
    s.verbose = verbose;
    s.initial_guess = int32_t(initial_guess);
    s.update_preconditioner = update_preconditioner;
    s.compute_preconditioner = compute_preconditioner;
    s.compute_timings = compute_timings;
    s.check_duality_gap = check_duality_gap;
    s.eps_duality_gap_abs = eps_duality_gap_abs;
    s.eps_duality_gap_rel = eps_duality_gap_rel;
    s.preconditioner_max_iter = preconditioner_max_iter;
    s.preconditioner_accuracy = preconditioner_accuracy;
    s.eps_primal_inf = eps_primal_inf;
    s.eps_dual_inf = eps_dual_inf;
    s.bcl_update = bcl_update;
    s.merit_function_type = int32_t(merit_function_type);
    s.alpha_gpdal = alpha_gpdal;
    s.primal_infeasibility_solving = primal_infeasibility_solving;
    s.frequence_infeasibility_check = frequence_infeasibility_check;
    s.default_H_eigenvalue_estimate = default_H_eigenvalue_estimate;
    s._pad = 0;
  }
};

} // namespace proxqp
} // namespace proxsuite

#endif
