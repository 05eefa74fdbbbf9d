"""`proxsuite.proxqp` namespace of the reference, dense backend only (MI355X build)."""
from .._ctypes_defs import (DenseBackend, EigenValueEstimateMethodOption, HessianType, InitialGuess,
                            MeritFunctionType, QPSolverOutput)
from . import dense

__all__ = ["dense", "DenseBackend", "EigenValueEstimateMethodOption", "HessianType", "InitialGuess", "MeritFunctionType", "QPSolverOutput",
           "omp_get_max_threads"]

# the reference's bindings export the members of these enums into the module scope as well (nanobind
# `.export_values()`, bindings/python/src/expose-results.hpp:23-31, expose-settings.hpp:22-47):
# proxsuite.proxqp.PROXQP_SOLVED, proxsuite.proxqp.NO_INITIAL_GUESS, ...
for _e in (QPSolverOutput, InitialGuess, MeritFunctionType, EigenValueEstimateMethodOption):
    globals().update(_e.__members__)
    __all__ += list(_e.__members__)
del _e


def omp_get_max_threads() -> int:
    """reference bindings/python/src/expose-all.cpp:26-28: the size of the OpenMP pool `solve_in_parallel` may use.
    There is no OpenMP pool here -- a batch is ONE kernel launch whatever `num_threads` says (one workgroup or one
    wavefront per QP; `dense.BatchQP` / `pqp_batch_launch_config` report that configuration) -- so what callers do with
    the number is size HOST thread pools and sweeps (benchmark/timings-parallel.py:96-99, :131: `range(1, n, 2)`,
    `ThreadPoolExecutor(max_workers=n)` around `dense.solve_no_gil`): it is the number of host threads that can drive
    the device at once, i.e. the CPUs this process may run on."""
    import os
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        return os.cpu_count() or 1
