"""proxsuite_amd -- the batched dense ProxQP path of Simple-Robotics/proxsuite on MI355X.

    from proxsuite_amd import proxqp                # proxsuite.proxqp.dense surface
    from proxsuite_amd.torch import QPFunction      # proxsuite.torch.qplayer forward

Importing the package is free of side effects; the gfx950 library (csrc/libproxqp_hip.so)
is loaded the first time a QP is constructed and its absence is an error, not a fallback.
"""
__all__ = ["proxqp"]


def __getattr__(name):
    if name == "proxqp":
        import importlib
        return importlib.import_module(".proxqp", __name__)
    raise AttributeError(name)
