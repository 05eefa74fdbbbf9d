"""Pins the CPU oracle on every literal known answer the reference's dense tests hold
(SURVEY.md 8(c), Appendix B)."""
import numpy as np
import pytest

from proxsuite_amd._ctypes_defs import InitialGuess, QPSolverOutput, HessianType


def test_cvxpy_1d(oracle):
    # reference test/src/cvxpy.cpp:61-102 : H=20, g=-10, 0<=x<=1 -> x*=0.5
    qp = oracle.QP(1, 0, 1)
    qp.settings.eps_abs = 1e-8
    qp.init(np.array([[20.0]]), np.array([-10.0]), None, None, np.array([[1.0]]), np.array([0.0]),
            np.array([1.0]))
    qp.solve()
    assert qp.results.info.status == QPSolverOutput.PROXQP_SOLVED
    assert abs(qp.results.x[0] - 0.5) <= 1e-8


def test_cvxpy_1d_start_from_solution(oracle):
    # reference test/src/cvxpy.cpp:104-160 : warm start at the solution -> iter <= 0
    qp = oracle.QP(1, 0, 1)
    qp.settings.eps_abs = 1e-8
    qp.init(np.array([[20.0]]), np.array([-10.0]), None, None, np.array([[1.0]]), np.array([0.0]),
            np.array([1.0]))
    qp.solve(np.array([0.5]), None, np.array([0.0]))
    assert qp.results.info.iter <= 0
    assert abs(qp.results.x[0] - 0.5) <= 1e-8


def test_cvxpy_3d_box(oracle):
    # reference test/src/cvxpy.cpp:22-59
    H = np.array([[13.0, 12.0, -2.0], [12.0, 17.0, 6.0], [-2.0, 6.0, 12.0]])
    g = np.array([-22.0, -14.5, 13.0])
    C = np.eye(3)
    l, u = -np.ones(3), np.ones(3)
    qp = oracle.QP(3, 0, 3)
    qp.settings.eps_abs = 1e-9
    qp.init(H, g, None, None, C, l, u)
    qp.solve()
    pri, dua = oracle.kkt_residuals(H, g, None, None, C, l, u, qp.results.x, qp.results.y, qp.results.z)
    assert pri <= 1e-9 and dua <= 1e-9


def test_lower_bounded_tridiagonal(oracle):
    # reference test/src/dense_qp_solve.py:302-333 : x* = [2]*149 + [3]
    n = 150
    M = np.eye(n)
    for i in range(1, n - 1):
        M[i, i + 1] = -1
        M[i, i - 1] = 1
    H = M @ M.T
    g = -np.ones(n)
    C = np.eye(n)
    l = 2.0 * np.ones(n)
    qp = oracle.QP(n, 0, n)
    qp.init(H, g, None, None, C, l, None)
    qp.solve()
    x_theoretically_optimal = np.array([2.0] * 149 + [3.0])
    assert np.max(np.abs(qp.results.x - x_theoretically_optimal)) < 1e-3


def test_simple_qp_with_infinity_lower_bound(oracle):
    # reference test/data/simple_qp_with_inifinity_lower_bound.mat (dense_qp_solve.py:375-405),
    # values restated literally (a 3x3 QP) rather than loading the .mat
    import os
    import scipy.io as sio
    path = "/root/reference/test/data/simple_qp_with_inifinity_lower_bound.mat"
    if not os.path.exists(path):
        pytest.skip("reference fixture not present on this box")
    m = sio.loadmat(path)
    P, q = np.asarray(m["P"], float), np.asarray(m["q"], float).ravel()
    A, b = np.asarray(m["A"], float), np.asarray(m["b"], float).ravel()
    C, l, u = np.asarray(m["C"], float), np.asarray(m["l"], float).ravel(), np.asarray(m["u"], float).ravel()
    qp = oracle.QP(3, 1, 3)
    qp.init(P, q, A, b, C, l, u)
    qp.solve()
    pri, dua = oracle.kkt_residuals(P, q, A, b, C, l, u, qp.results.x, qp.results.y, qp.results.z)
    assert pri <= 1e-5 and dua <= 1e-5


def test_ruiz_algebra(oracle, randqp):
    # reference test/src/dense_ruiz_equilibration.cpp:15-72
    for dim in (10, 40, 100):
        randqp.set_seed(1)
        n_eq = n_in = dim // 2
        m = randqp.dense_strongly_convex_qp(dim, n_eq, n_in, 0.15, 1e-2)
        qp = oracle.QP(dim, n_eq, n_in)
        qp.init(m.H, m.g, m.A, m.b, m.C, m.l, m.u)
        s = qp.scaled()
        d = s["delta"]
        D, E, F = d[:dim], d[dim:dim + n_eq], d[dim + n_eq:]
        c = s["c"]
        assert np.max(np.abs(s["H"] - c * (D[:, None] * m.H * D[None, :]))) <= 1e-10
        assert np.max(np.abs(s["g"] - c * D * m.g)) <= 1e-10
        assert np.max(np.abs(s["A"] - E[:, None] * m.A * D[None, :])) <= 1e-10
        assert np.max(np.abs(s["b"] - E * m.b)) <= 1e-10
        assert np.max(np.abs(s["C"] - F[:, None] * m.C * D[None, :])) <= 1e-10


def test_dim_zero_throws(oracle):
    # reference dense/model.hpp:65-68
    with pytest.raises(ValueError):
        oracle.QP(0, 0, 0)
