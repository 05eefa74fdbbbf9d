"""alias of proxsuite_amd.torch (reference bindings/python/proxsuite/torch/__init__.py)"""
import sys as _sys

from proxsuite_amd.torch import qplayer  # noqa: F401
from proxsuite_amd.torch.qplayer import QPFunction  # noqa: F401

_sys.modules[__name__ + ".qplayer"] = qplayer
__all__ = ["QPFunction", "qplayer"]
