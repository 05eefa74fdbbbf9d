"""alias of proxsuite_amd.proxqp (the `proxsuite.proxqp` namespace of the reference binding)"""
import sys as _sys

import proxsuite_amd.proxqp as _impl
from proxsuite_amd.proxqp import *  # noqa: F401,F403

globals().update({_k: _v for _k, _v in vars(_impl).items() if not _k.startswith("_")})  # (the exported enum members too)
from proxsuite_amd.proxqp import dense  # noqa: F401

# `import proxsuite.proxqp.dense` and `from proxsuite.proxqp.dense import QP` resolve to the same module object
_sys.modules[__name__ + ".dense"] = dense
__all__ = list(_impl.__all__)
