"""Pins the CPU oracle on the reference's hardest real-world fixtures:
reference test/src/dense_maros_meszaros.cpp:85-169 (eps_abs=2e-8, dual < 2 eps, primal within
eps, warm re-solve must take iter == 0)."""
import glob
import os

import numpy as np
import pytest

from conftest import split_maros
from proxsuite_amd._ctypes_defs import InitialGuess

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "maros_meszaros_small.npz")
EPS = 2e-8


def _check(oracle, P, q, A, l, u):
    H, g, Aeq, b, C, lin, uin = split_maros(P, q, A, l, u)
    n, n_eq, n_in = H.shape[0], Aeq.shape[0], C.shape[0]
    qp = oracle.QP(n, n_eq, n_in)
    qp.init(H, g, Aeq, b, C, lin, uin)
    qp.settings.eps_abs = EPS
    qp.settings.eps_rel = 0
    qp.settings.eps_primal_inf = 1e-12
    qp.settings.eps_dual_inf = 1e-12
    for it in range(2):
        if it > 0:
            qp.settings.initial_guess = InitialGuess.WARM_START_WITH_PREVIOUS_RESULT
        qp.solve()
        x, y, z = qp.results.x, qp.results.y, qp.results.z
        dua = H @ x + g
        if n_eq:
            dua = dua + Aeq.T @ y
        if n_in:
            dua = dua + C.T @ z
            assert (C @ x - lin).min() > -EPS
            assert (C @ x - uin).max() < EPS
        assert np.max(np.abs(dua)) < 2 * EPS
        if n_eq:
            assert np.max(np.abs(Aeq @ x - b)) < EPS * 1.0001
        if it > 0:
            assert qp.results.info.iter == 0


def _names():
    d = np.load(GOLD)
    return [str(s) for s in d["names"]]


@pytest.mark.parametrize("name", _names())
def test_maros_meszaros_small(oracle, name):
    d = np.load(GOLD)
    _check(oracle, *(d["%s/%s" % (name, k)] for k in "PqAlu"))


def _medium():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_maros_meszaros_fixtures as mm
    return mm


def _medium_names():
    d = np.load(os.path.join(os.path.dirname(GOLD), "maros_meszaros_medium.npz"))
    return [str(s) for s in d["names"]]


@pytest.mark.parametrize("name", _medium_names())
def test_maros_meszaros_medium(oracle, name):
    """the other 29 problems the reference test runs (n <= 1000 and n_eq + n_in <= 1000,
    test/src/dense_maros_meszaros.cpp:97), from the committed triplet fixture"""
    _check(oracle, *_medium().load_medium(only=name)[name])


@pytest.mark.parametrize("name", ["GOULDQP2", "CVXQP2_M"])
def test_maros_meszaros_above_1024_rows(oracle, name):
    """two problems beyond the reference test's selection (1048 and 1250 constraint rows: more rows than the widest
    workgroup has threads), to the same acceptance lines; the GPU suite runs them through the chunked 1024-thread
    kernel (tests/test_gpu_parity.py::test_maros_meszaros_above_1024_rows)"""
    mm = _medium()
    _check(oracle, *mm.load_medium(mm.OUT_LARGE, only=name)[name])
