"""`proxsuite.proxqp` namespace of the reference, dense backend only (MI355X build)."""
from .._ctypes_defs import (DenseBackend, EigenValueEstimateMethodOption, HessianType, InitialGuess,
                            MeritFunctionType, QPSolverOutput)
from . import dense

__all__ = ["dense", "DenseBackend", "EigenValueEstimateMethodOption", "HessianType", "InitialGuess", "MeritFunctionType", "QPSolverOutput",
           "omp_get_max_threads"]


def omp_get_max_threads() -> int:
    """reference bindings/python/src/expose-all.cpp:26-28.  There is no OpenMP pool here: the
    unit of parallelism is one workgroup per QP, so this reports the number of compute units
    of HIP device 0 times the workgroups resident per CU (an upper bound on QPs in flight)."""
    import torch
    p = torch.cuda.get_device_properties(0)
    return int(p.multi_processor_count) * 4
