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
    """reference bindings/python/src/expose-all.cpp:26-28.  There is no OpenMP pool here: the
    unit of parallelism is one workgroup per QP, so this reports the number of compute units
    of HIP device 0 times the workgroups resident per CU (an upper bound on QPs in flight)."""
    try:
        import torch
        if torch.cuda.is_available():
            p = torch.cuda.get_device_properties(0)
            return int(p.multi_processor_count) * 4
    except Exception:  # (no torch: the count below)
        pass
    import os
    return os.cpu_count() or 1  # (no device visible, e.g. the CPU emulator of the test-suite: callers only pass it back as num_threads)
