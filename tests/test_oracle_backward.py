"""Oracle `compute_backward` (restating reference dense/compute_ECJ.hpp:29-189) pinned the way the
reference pins it: against central finite differences of the solution map
(reference test/src/dense_backward.cpp:16-221: dx/dg and dx/db on feasible QPs with and without
inequalities, |difference| < 1e-5)."""
import numpy as np
import pytest

from proxsuite_amd.utils import random_qp as R

EPS_FD = 1e-5
TOL = 1e-5


def _solve(oracle, m, n, ne, ni, g=None, b=None):
    q = oracle.QP(n, ne, ni)
    q.settings.eps_abs = 1e-9
    q.settings.eps_rel = 0
    q.init(m.H, m.g if g is None else g, m.A if ne else None, (m.b if b is None else b) if ne else None,
           m.C if ni else None, m.l if ni else None, m.u if ni else None)
    q.solve()
    return q


@pytest.mark.parametrize("n_eq,n_in", [(5, 0), (5, 2), (3, 6)])
def test_dx_dg_matches_finite_differences(oracle, n_eq, n_in):
    n = 10
    R.set_seed(1)
    m = R.dense_strongly_convex_qp(n, n_eq, n_in, 0.85, 1e-1)
    qp = _solve(oracle, m, n, n_eq, n_in)
    dx_dg = np.zeros((n, n))
    for i in range(n):
        ld = np.zeros(n + n_eq + n_in)
        ld[i] = 1.0
        dx_dg[i] = qp.compute_backward(ld, 1e-5, 1e-7, 1e-7)["dL_dg"]
    fd = np.zeros((n, n))
    for i in range(n):
        gp, gm = m.g.copy(), m.g.copy()
        gp[i] += EPS_FD
        gm[i] -= EPS_FD
        fd[:, i] = (_solve(oracle, m, n, n_eq, n_in, g=gp).results.x
                    - _solve(oracle, m, n, n_eq, n_in, g=gm).results.x) / (2 * EPS_FD)
    assert np.max(np.abs(fd - dx_dg)) < TOL


def test_dx_db_matches_finite_differences(oracle):
    n, n_eq, n_in = 10, 5, 0
    R.set_seed(1)
    m = R.dense_strongly_convex_qp(n, n_eq, n_in, 0.85, 1e-1)
    qp = _solve(oracle, m, n, n_eq, n_in)
    dx_db = np.zeros((n, n_eq))
    for i in range(n):
        ld = np.zeros(n + n_eq + n_in)
        ld[i] = 1.0
        dx_db[i] = qp.compute_backward(ld, 1e-5, 1e-7, 1e-7)["dL_db"]
    fd = np.zeros((n, n_eq))
    for i in range(n_eq):
        bp, bm = m.b.copy(), m.b.copy()
        bp[i] += EPS_FD
        bm[i] -= EPS_FD
        fd[:, i] = (_solve(oracle, m, n, n_eq, n_in, b=bp).results.x
                    - _solve(oracle, m, n, n_eq, n_in, b=bm).results.x) / (2 * EPS_FD)
    assert np.max(np.abs(fd - dx_db)) < TOL


def test_loss_gradients_wrt_matrices(oracle):
    """dL/dH, dL/dA, dL/dC, dL/du for L = w . x*, against finite differences of L."""
    n, n_eq, n_in = 8, 3, 5
    R.set_seed(3)
    m = R.dense_strongly_convex_qp(n, n_eq, n_in, 0.85, 1e-1)
    rng = np.random.default_rng(0)
    wx = rng.standard_normal(n)
    qp = _solve(oracle, m, n, n_eq, n_in)
    ld = np.concatenate([wx, np.zeros(n_eq + n_in)])
    bd = qp.compute_backward(ld, 1e-9, 1e-9, 1e-9)

    def loss(H=None, A=None, C=None, u=None):
        q = oracle.QP(n, n_eq, n_in)
        q.settings.eps_abs = 1e-11
        q.settings.eps_rel = 0
        q.init(m.H if H is None else H, m.g, m.A if A is None else A, m.b, m.C if C is None else C, m.l,
               m.u if u is None else u)
        q.solve()
        return float(wx @ q.results.x)

    h = 1e-6
    for (i, j) in [(0, 0), (1, 4), (5, 2)]:
        Hp, Hm = m.H.copy(), m.H.copy()
        Hp[i, j] += h; Hp[j, i] += h if i != j else 0
        Hm[i, j] -= h; Hm[j, i] -= h if i != j else 0
        fd = (loss(H=Hp) - loss(H=Hm)) / (2 * h)
        ref = bd["dL_dH"][i, j] + (bd["dL_dH"][j, i] if i != j else 0)
        assert abs(fd - ref) < 1e-5, (i, j, fd, ref)
    for (i, j) in [(0, 1), (2, 7)]:
        Ap, Am = m.A.copy(), m.A.copy()
        Ap[i, j] += h; Am[i, j] -= h
        assert abs((loss(A=Ap) - loss(A=Am)) / (2 * h) - bd["dL_dA"][i, j]) < 1e-5
    for i in range(n_in):
        up, um = m.u.copy(), m.u.copy()
        up[i] += h; um[i] -= h
        assert abs((loss(u=up) - loss(u=um)) / (2 * h) - bd["dL_du"][i]) < 1e-5
        Cp, Cm = m.C.copy(), m.C.copy()
        Cp[i, 3] += h; Cm[i, 3] -= h
        assert abs((loss(C=Cp) - loss(C=Cm)) / (2 * h) - bd["dL_dC"][i, 3]) < 1e-5
