"""`proxsuite` import name of the reference's Python package (bindings/python/proxsuite/__init__.py),
served by the MI355X batched dense backend of this repository:

    import proxsuite
    qp = proxsuite.proxqp.dense.QP(n, n_eq, n_in)          # bindings/python/src/expose-qpobject.hpp
    from proxsuite.torch.qplayer import QPFunction         # bindings/python/proxsuite/torch/qplayer.py:12-20

Only the dense path exists here (proxsuite.proxqp.sparse raises ImportError: it stays on the CPU
reference).  Put the repository root on PYTHONPATH *instead of* an installed ProxSuite to switch a
program over; nothing else in it has to change.  Thin aliases only: the implementation is `proxsuite_amd`.
"""
import importlib as _importlib

from proxsuite_amd import proxqp  # noqa: F401

__all__ = ["proxqp", "torch"]


def __getattr__(name):
    if name == "torch":  # lazy: importing proxsuite must not import PyTorch
        return _importlib.import_module(".torch", __name__)
    raise AttributeError(name)
