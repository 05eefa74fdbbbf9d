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


@pytest.mark.slow
@pytest.mark.skipif(not os.path.isdir("/root/reference/test/data/maros_meszaros_data"),
                    reason="reference fixtures not on this box")
def test_maros_meszaros_medium_from_reference():
    """A few of the larger problems straight from the reference tree (authoring box only)."""
    import scipy.io as sio
    from oracle import oracle as O
    for name in ("PRIMAL1", "QRECIPE", "QSC205", "QPCBOEI2", "QE226"):
        m = sio.loadmat("/root/reference/test/data/maros_meszaros_data/%s.mat" % name)
        dense = lambda a: a.toarray() if hasattr(a, "toarray") else np.asarray(a)
        _check(O, dense(m["P"]).astype(float), np.asarray(m["q"], float).ravel(), dense(m["A"]).astype(float),
               np.asarray(m["l"], float).ravel(), np.asarray(m["u"], float).ravel())
