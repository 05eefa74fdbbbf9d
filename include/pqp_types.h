/*
 * pqp_types.h -- plain-C data contract shared by the C-ABI (proxqp_hip.h), the
 * host-side C++ facade and the test oracle.
 *
 * Every field mirrors a member of the reference's public structs so that the
 * parity tests can compare counters one to one:
 *   pqp_settings  <->  proxsuite::proxqp::Settings<T>
 *                      (reference include/proxsuite/proxqp/settings.hpp:96-315)
 *   pqp_info      <->  proxsuite::proxqp::Info<T>
 *                      (reference include/proxsuite/proxqp/results.hpp:27-58)
 *   enums         <->  include/proxsuite/proxqp/status.hpp:17-43 and
 *                      include/proxsuite/proxqp/settings.hpp:19-53
 *
 * All floating point is fp64 (the reference only instantiates f64 for the dense
 * backend: bindings/python/src/expose-all.cpp:94).
 */
#ifndef PQP_TYPES_H
#define PQP_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* QPSolverOutput, reference status.hpp:17-26 (same numeric order). */
enum pqp_status
{
  PQP_SOLVED = 0,
  PQP_MAX_ITER_REACHED = 1,
  PQP_PRIMAL_INFEASIBLE = 2,
  PQP_SOLVED_CLOSEST_PRIMAL_FEASIBLE = 3,
  PQP_DUAL_INFEASIBLE = 4,
  PQP_NOT_RUN = 5
};

/* InitialGuessStatus, reference status.hpp:28-35 (same numeric order). */
enum pqp_initial_guess
{
  PQP_NO_INITIAL_GUESS = 0,
  PQP_EQUALITY_CONSTRAINED_INITIAL_GUESS = 1,
  PQP_WARM_START_WITH_PREVIOUS_RESULT = 2,
  PQP_WARM_START = 3,
  PQP_COLD_START_WITH_PREVIOUS_RESULT = 4
};

/* DenseBackend, reference settings.hpp:28-34. */
enum pqp_dense_backend
{
  PQP_BACKEND_AUTOMATIC = 0,
  PQP_BACKEND_PRIMAL_DUAL_LDLT = 1,
  PQP_BACKEND_PRIMAL_LDLT = 2
};

/* HessianType, reference settings.hpp:42-47. */
enum pqp_hessian_type
{
  PQP_HESSIAN_ZERO = 0,
  PQP_HESSIAN_DENSE = 1,
  PQP_HESSIAN_DIAGONAL = 2
};

/* MeritFunctionType, reference settings.hpp:36-40. */
enum pqp_merit_function
{
  PQP_MERIT_GPDAL = 0,
  PQP_MERIT_PDAL = 1
};

/* Settings<T>; defaults are set by pqp_settings_default() exactly as the
 * reference constructor does (settings.hpp:213-315). */
typedef struct pqp_settings
{
  double default_rho;
  double default_mu_eq;
  double default_mu_in;
  double alpha_bcl;
  double beta_bcl;
  double refactor_dual_feasibility_threshold;
  double refactor_rho_threshold;
  double mu_min_eq;
  double mu_min_in;
  double mu_max_eq_inv;
  double mu_max_in_inv;
  double mu_update_factor;
  double mu_update_inv_factor;
  double cold_reset_mu_eq;
  double cold_reset_mu_in;
  double cold_reset_mu_eq_inv;
  double cold_reset_mu_in_inv;
  double eps_abs;
  double eps_rel;
  double eps_refact;
  double eps_duality_gap_abs;
  double eps_duality_gap_rel;
  double preconditioner_accuracy;
  double eps_primal_inf;
  double eps_dual_inf;
  double alpha_gpdal;
  double default_H_eigenvalue_estimate;
  int64_t max_iter;
  int64_t max_iter_in;
  int64_t safe_guard;
  int64_t nb_iterative_refinement;
  int64_t preconditioner_max_iter;
  int64_t frequence_infeasibility_check;
  int32_t initial_guess; /* enum pqp_initial_guess */
  int32_t merit_function_type;
  int32_t verbose;
  int32_t update_preconditioner;
  int32_t compute_preconditioner;
  int32_t compute_timings;
  int32_t check_duality_gap;
  int32_t bcl_update;
  int32_t primal_infeasibility_solving;
  int32_t _pad;
} pqp_settings;

/* Info<T>. */
typedef struct pqp_info
{
  double mu_eq;
  double mu_eq_inv;
  double mu_in;
  double mu_in_inv;
  double rho;
  double nu;
  int64_t iter;
  int64_t iter_ext;
  int64_t mu_updates;
  int64_t rho_updates;
  double setup_time;
  double solve_time;
  double run_time;
  double objValue;
  double pri_res;
  double dua_res;
  double duality_gap;
  double iterative_residual;
  double minimal_H_eigenvalue_estimate;
  int32_t status; /* enum pqp_status */
  int32_t _pad;
} pqp_info;

/* Fill `s` with the reference defaults for the given dense backend
 * (settings.hpp:213-315: default_rho 1e-6, or 1e-5 for PrimalLDLT). */
static inline void
pqp_settings_default(pqp_settings* s, int dense_backend)
{
  s->default_rho = (dense_backend == PQP_BACKEND_PRIMAL_LDLT) ? 1.E-5 : 1.E-6;
  s->default_mu_eq = 1.E-3;
  s->default_mu_in = 1.E-1;
  s->alpha_bcl = 0.1;
  s->beta_bcl = 0.9;
  s->refactor_dual_feasibility_threshold = 1e-2;
  s->refactor_rho_threshold = 1e-7;
  s->mu_min_eq = 1e-9;
  s->mu_min_in = 1e-8;
  s->mu_max_eq_inv = 1e9;
  s->mu_max_in_inv = 1e8;
  s->mu_update_factor = 0.1;
  s->mu_update_inv_factor = 10;
  s->cold_reset_mu_eq = 1. / 1.1;
  s->cold_reset_mu_in = 1. / 1.1;
  s->cold_reset_mu_eq_inv = 1.1;
  s->cold_reset_mu_in_inv = 1.1;
  s->eps_abs = 1.e-5;
  s->eps_rel = 0;
  s->eps_refact = 1.e-6;
  s->eps_duality_gap_abs = 1.e-4;
  s->eps_duality_gap_rel = 0;
  s->preconditioner_accuracy = 1.e-3;
  s->eps_primal_inf = 1.E-4;
  s->eps_dual_inf = 1.E-4;
  s->alpha_gpdal = 0.95;
  s->default_H_eigenvalue_estimate = 0.;
  s->max_iter = 10000;
  s->max_iter_in = 1500;
  s->safe_guard = 10000;
  s->nb_iterative_refinement = 10;
  s->preconditioner_max_iter = 10;
  s->frequence_infeasibility_check = 1;
  s->initial_guess = PQP_EQUALITY_CONSTRAINED_INITIAL_GUESS;
  s->merit_function_type = PQP_MERIT_GPDAL;
  s->verbose = 0;
  s->update_preconditioner = 0;
  s->compute_preconditioner = 1;
  s->compute_timings = 0;
  s->check_duality_gap = 0;
  s->bcl_update = 1;
  s->primal_infeasibility_solving = 0;
  s->_pad = 0;
}

/* Info defaults of the Results constructor (results.hpp:90-144). */
static inline void
pqp_info_default(pqp_info* i, int dense_backend)
{
  i->rho = (dense_backend == PQP_BACKEND_PRIMAL_LDLT) ? 1.E-5 : 1.E-6;
  i->mu_eq_inv = 1e3;
  i->mu_eq = 1e-3;
  i->mu_in_inv = 1e1;
  i->mu_in = 1e-1;
  i->nu = 1.;
  i->iter = 0;
  i->iter_ext = 0;
  i->mu_updates = 0;
  i->rho_updates = 0;
  i->run_time = 0;
  i->setup_time = 0;
  i->solve_time = 0;
  i->objValue = 0.;
  i->pri_res = 0.;
  i->dua_res = 0.;
  i->duality_gap = 0.;
  i->iterative_residual = 0.;
  i->status = PQP_NOT_RUN;
  i->minimal_H_eigenvalue_estimate = 0.;
  i->_pad = 0;
}

#ifdef __cplusplus
}
#endif

#endif /* PQP_TYPES_H */
