// proxsuite/proxqp/status.hpp -- solver outcome and initial-guess enums of the dense ProxQP
// API, MI355X build.  Same enumerators and values as the reference
// (include/proxsuite/proxqp/status.hpp:17-45); the values are those of include/pqp_types.h.
#ifndef PROXSUITE_AMD_PROXQP_STATUS_HPP
#define PROXSUITE_AMD_PROXQP_STATUS_HPP

#include "pqp_types.h"

namespace proxsuite {
namespace proxqp {

enum struct QPSolverOutput
{
  PROXQP_SOLVED = PQP_SOLVED,
  PROXQP_MAX_ITER_REACHED = PQP_MAX_ITER_REACHED,
  PROXQP_PRIMAL_INFEASIBLE = PQP_PRIMAL_INFEASIBLE,
  PROXQP_SOLVED_CLOSEST_PRIMAL_FEASIBLE = PQP_SOLVED_CLOSEST_PRIMAL_FEASIBLE,
  PROXQP_DUAL_INFEASIBLE = PQP_DUAL_INFEASIBLE,
  PROXQP_NOT_RUN = PQP_NOT_RUN
};

enum struct InitialGuessStatus
{
  NO_INITIAL_GUESS = PQP_NO_INITIAL_GUESS,
  EQUALITY_CONSTRAINED_INITIAL_GUESS = PQP_EQUALITY_CONSTRAINED_INITIAL_GUESS,
  WARM_START_WITH_PREVIOUS_RESULT = PQP_WARM_START_WITH_PREVIOUS_RESULT,
  WARM_START = PQP_WARM_START,
  COLD_START_WITH_PREVIOUS_RESULT = PQP_COLD_START_WITH_PREVIOUS_RESULT
};

} // namespace proxqp
} // namespace proxsuite

#endif
