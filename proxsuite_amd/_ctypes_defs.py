"""ctypes mirrors of include/pqp_types.h (pqp_settings, pqp_info) and the enums."""
from __future__ import annotations

import ctypes as C
import enum


class QPSolverOutput(enum.IntEnum):
    """reference include/proxsuite/proxqp/status.hpp:17-26"""
    PROXQP_SOLVED = 0
    PROXQP_MAX_ITER_REACHED = 1
    PROXQP_PRIMAL_INFEASIBLE = 2
    PROXQP_SOLVED_CLOSEST_PRIMAL_FEASIBLE = 3
    PROXQP_DUAL_INFEASIBLE = 4
    PROXQP_NOT_RUN = 5


class InitialGuess(enum.IntEnum):
    """reference status.hpp:28-35 (python name: proxsuite.proxqp.InitialGuess)"""
    NO_INITIAL_GUESS = 0
    EQUALITY_CONSTRAINED_INITIAL_GUESS = 1
    WARM_START_WITH_PREVIOUS_RESULT = 2
    WARM_START = 3
    COLD_START_WITH_PREVIOUS_RESULT = 4


class DenseBackend(enum.IntEnum):
    """reference settings.hpp:28-34"""
    Automatic = 0
    PrimalDualLDLT = 1
    PrimalLDLT = 2


class HessianType(enum.IntEnum):
    """reference settings.hpp:42-47"""
    Zero = 0
    Dense = 1
    Diagonal = 2


class EigenValueEstimateMethodOption(enum.IntEnum):
    """reference settings.hpp:49-53"""
    PowerIteration = 0
    ExactMethod = 1


class MeritFunctionType(enum.IntEnum):
    """reference settings.hpp:36-40"""
    GPDAL = 0
    PDAL = 1


_D = C.c_double
_I64 = C.c_int64
_I32 = C.c_int32


class pqp_settings(C.Structure):
    _fields_ = [(n, _D) for n in (
        "default_rho", "default_mu_eq", "default_mu_in", "alpha_bcl", "beta_bcl",
        "refactor_dual_feasibility_threshold", "refactor_rho_threshold", "mu_min_eq", "mu_min_in",
        "mu_max_eq_inv", "mu_max_in_inv", "mu_update_factor", "mu_update_inv_factor",
        "cold_reset_mu_eq", "cold_reset_mu_in", "cold_reset_mu_eq_inv", "cold_reset_mu_in_inv",
        "eps_abs", "eps_rel", "eps_refact", "eps_duality_gap_abs", "eps_duality_gap_rel",
        "preconditioner_accuracy", "eps_primal_inf", "eps_dual_inf", "alpha_gpdal",
        "default_H_eigenvalue_estimate")] + [(n, _I64) for n in (
        "max_iter", "max_iter_in", "safe_guard", "nb_iterative_refinement",
        "preconditioner_max_iter", "frequence_infeasibility_check")] + [(n, _I32) for n in (
        "initial_guess", "merit_function_type", "verbose", "update_preconditioner",
        "compute_preconditioner", "compute_timings", "check_duality_gap", "bcl_update",
        "primal_infeasibility_solving", "_pad")]


class pqp_info(C.Structure):
    _fields_ = [(n, _D) for n in ("mu_eq", "mu_eq_inv", "mu_in", "mu_in_inv", "rho", "nu")] + [
        (n, _I64) for n in ("iter", "iter_ext", "mu_updates", "rho_updates")] + [
        (n, _D) for n in ("setup_time", "solve_time", "run_time", "objValue", "pri_res", "dua_res",
                          "duality_gap", "iterative_residual", "minimal_H_eigenvalue_estimate")] + [
        ("status", _I32), ("_pad", _I32)]


def default_settings(dense_backend: int = DenseBackend.PrimalDualLDLT) -> pqp_settings:
    """pqp_settings_default() of include/pqp_types.h (reference settings.hpp:213-315)."""
    s = pqp_settings()
    s.default_rho = 1.0e-5 if dense_backend == DenseBackend.PrimalLDLT else 1.0e-6
    s.default_mu_eq = 1.0e-3
    s.default_mu_in = 1.0e-1
    s.alpha_bcl = 0.1
    s.beta_bcl = 0.9
    s.refactor_dual_feasibility_threshold = 1e-2
    s.refactor_rho_threshold = 1e-7
    s.mu_min_eq = 1e-9
    s.mu_min_in = 1e-8
    s.mu_max_eq_inv = 1e9
    s.mu_max_in_inv = 1e8
    s.mu_update_factor = 0.1
    s.mu_update_inv_factor = 10
    s.cold_reset_mu_eq = 1.0 / 1.1
    s.cold_reset_mu_in = 1.0 / 1.1
    s.cold_reset_mu_eq_inv = 1.1
    s.cold_reset_mu_in_inv = 1.1
    s.eps_abs = 1.0e-5
    s.eps_rel = 0
    s.eps_refact = 1.0e-6
    s.eps_duality_gap_abs = 1.0e-4
    s.eps_duality_gap_rel = 0
    s.preconditioner_accuracy = 1.0e-3
    s.eps_primal_inf = 1.0e-4
    s.eps_dual_inf = 1.0e-4
    s.alpha_gpdal = 0.95
    s.default_H_eigenvalue_estimate = 0.0
    s.max_iter = 10000
    s.max_iter_in = 1500
    s.safe_guard = 10000
    s.nb_iterative_refinement = 10
    s.preconditioner_max_iter = 10
    s.frequence_infeasibility_check = 1
    s.initial_guess = InitialGuess.EQUALITY_CONSTRAINED_INITIAL_GUESS
    s.merit_function_type = MeritFunctionType.GPDAL
    s.verbose = 0
    s.update_preconditioner = 0
    s.compute_preconditioner = 1
    s.compute_timings = 0
    s.check_duality_gap = 0
    s.bcl_update = 1
    s.primal_infeasibility_solving = 0
    return s


def dense_backend_choice(backend: int, dim: int, n_eq: int, n_in: int, box: bool) -> int:
    """reference dense/wrapper.hpp:81-113"""
    if backend != DenseBackend.Automatic:
        return int(backend)
    n_constraints = n_in + (dim if box else 0)
    threshold, frequence = 1.5, 0.2
    d = float(dim)
    pd = (0.5 * (n_eq / d) ** 2 + 0.17 * ((n_eq / d) ** 3 + (n_constraints / d) ** 3)
          + frequence * ((n_eq + n_constraints) / d) ** 2 / d)
    p = threshold * ((0.5 * n_eq + n_constraints) / d + frequence / d)
    return int(DenseBackend.PrimalLDLT if pd > p else DenseBackend.PrimalDualLDLT)
