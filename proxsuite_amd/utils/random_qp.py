"""Synthetic random QP families of the reference's test/benchmark utilities
(reference include/proxsuite/proxqp/utils/random_qp_problems.hpp), backed by the
host-only native generator proxsuite_amd/csrc/random_qp.cpp so that the CPU baseline
and the MI355X path are fed the very same Lehmer-seeded problems
(reference benchmark/timings-parallel.cpp:43-63)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from .. import _build

_lib = None


def _load():
    global _lib
    if _lib is None:
        path = _build.build_randqp()
        lib = C.CDLL(str(path))
        dp = C.POINTER(C.c_double)
        lib.pqp_rand_set_seed.argtypes = [C.c_uint64]
        lib.pqp_rand_uniform.restype = C.c_double
        lib.pqp_rand_normal.restype = C.c_double
        lib.pqp_dense_strongly_convex_qp.argtypes = [C.c_int64] * 3 + [C.c_double] * 2 + [dp] * 7
        lib.pqp_dense_strongly_convex_qp_batch.argtypes = (
            [C.c_int64, C.c_uint64] + [C.c_int64] * 3 + [C.c_double] * 2 + [dp] * 7)
        lib.pqp_dense_not_strongly_convex_qp.argtypes = [C.c_int64] * 3 + [C.c_double] + [dp] * 7
        lib.pqp_dense_degenerate_qp.argtypes = [C.c_int64] * 3 + [C.c_double] * 2 + [dp] * 7
        lib.pqp_dense_box_constrained_qp.argtypes = [C.c_int64] * 3 + [C.c_double] * 2 + [dp] * 7
        lib.pqp_dense_unconstrained_qp.argtypes = [C.c_int64] + [C.c_double] * 2 + [dp] * 2
        _lib = lib
    return _lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


@dataclass
class Model:
    """Counterpart of proxsuite::proxqp::dense::Model<T> (row-major numpy arrays)."""
    H: np.ndarray
    g: np.ndarray
    A: np.ndarray
    b: np.ndarray
    C: np.ndarray
    u: np.ndarray
    l: np.ndarray

    @property
    def dim(self):
        return self.H.shape[-1]

    @property
    def n_eq(self):
        return self.A.shape[-2]

    @property
    def n_in(self):
        return self.C.shape[-2]


def min_eigenvalue_symmetric(H) -> float:
    """Smallest eigenvalue of a symmetric matrix (Householder tridiagonalisation + Sturm bisection)."""
    H = np.ascontiguousarray(np.asarray(H, dtype=np.float64))
    L = _load()
    L.pqp_min_eigenvalue_symmetric.restype = C.c_double
    L.pqp_min_eigenvalue_symmetric.argtypes = [C.c_int64, C.POINTER(C.c_double)]
    return float(L.pqp_min_eigenvalue_symmetric(H.shape[0], _p(H)))


def set_seed(seed: int) -> None:
    """reference random_qp_problems.hpp:121-127"""
    _load().pqp_rand_set_seed(int(seed))


def uniform_rand() -> float:
    return _load().pqp_rand_uniform()


def normal_rand() -> float:
    return _load().pqp_rand_normal()


def _alloc(n, n_eq, n_in, batch=None):
    pre = () if batch is None else (batch,)
    z = lambda *s: np.zeros(pre + s, dtype=np.float64)
    return z(n, n), z(n), z(n_eq, n), z(n_eq), z(n_in, n), z(n_in), z(n_in)


def dense_strongly_convex_qp(dim, n_eq, n_in, sparsity_factor, strong_convexity_factor=1e-2) -> Model:
    """reference random_qp_problems.hpp:462-502 (uses the global RNG state: call set_seed first)"""
    H, g, A, b, Cm, u, l = _alloc(dim, n_eq, n_in)
    _load().pqp_dense_strongly_convex_qp(dim, n_eq, n_in, sparsity_factor, strong_convexity_factor,
                                         _p(H), _p(g), _p(A), _p(b), _p(Cm), _p(u), _p(l))
    return Model(H, g, A, b, Cm, u, l)


def dense_strongly_convex_qp_batch(batch, dim, n_eq, n_in, sparsity_factor=0.15,
                                   strong_convexity_factor=1e-2, seed0=0) -> Model:
    """The benchmark's generation loop (reference benchmark/timings-parallel.cpp:43-63):
    QP i is `set_seed(seed0 + i); dense_strongly_convex_qp(...)`.  Arrays are [B, ...]."""
    H, g, A, b, Cm, u, l = _alloc(dim, n_eq, n_in, batch)
    _load().pqp_dense_strongly_convex_qp_batch(batch, seed0, dim, n_eq, n_in, sparsity_factor,
                                               strong_convexity_factor, _p(H), _p(g), _p(A), _p(b),
                                               _p(Cm), _p(u), _p(l))
    return Model(H, g, A, b, Cm, u, l)


def dense_not_strongly_convex_qp(dim, n_eq, n_in, sparsity_factor) -> Model:
    """reference random_qp_problems.hpp:504-543"""
    H, g, A, b, Cm, u, l = _alloc(dim, n_eq, n_in)
    _load().pqp_dense_not_strongly_convex_qp(dim, n_eq, n_in, sparsity_factor, _p(H), _p(g), _p(A),
                                             _p(b), _p(Cm), _p(u), _p(l))
    return Model(H, g, A, b, Cm, u, l)


def dense_degenerate_qp(dim, n_eq, n_in, sparsity_factor, strong_convexity_factor=1e-2) -> Model:
    """reference random_qp_problems.hpp:545-589 (C has 2*n_in rows)"""
    H, g, A, b, Cm, u, l = _alloc(dim, n_eq, 2 * n_in)
    _load().pqp_dense_degenerate_qp(dim, n_eq, n_in, sparsity_factor, strong_convexity_factor,
                                    _p(H), _p(g), _p(A), _p(b), _p(Cm), _p(u), _p(l))
    return Model(H, g, A, b, Cm, u, l)


def dense_box_constrained_qp(dim, n_eq, n_in, sparsity_factor, strong_convexity_factor=1e-2) -> Model:
    """reference random_qp_problems.hpp:591-628 (C = I)"""
    H, g, A, b, Cm, u, l = _alloc(dim, n_eq, n_in)
    _load().pqp_dense_box_constrained_qp(dim, n_eq, n_in, sparsity_factor, strong_convexity_factor,
                                         _p(H), _p(g), _p(A), _p(b), _p(Cm), _p(u), _p(l))
    return Model(H, g, A, b, Cm, u, l)


def dense_unconstrained_qp(dim, sparsity_factor, strong_convexity_factor=1e-2) -> Model:
    """reference random_qp_problems.hpp:438-460"""
    H, g, A, b, Cm, u, l = _alloc(dim, 0, 0)
    _load().pqp_dense_unconstrained_qp(dim, sparsity_factor, strong_convexity_factor, _p(H), _p(g))
    return Model(H, g, A, b, Cm, u, l)
