"""The reference's universal acceptance test (KKT residuals <= eps_abs on the unscaled
model, eps_abs=1e-9, eps_rel=0) applied to the CPU oracle on every random QP family of
reference test/src/dense_qp_with_eq_and_in.cpp, dense_qp_eq.cpp, dense_unconstrained_qp.cpp,
dense_qp_wrapper.cpp (box ordering)."""
import numpy as np
import pytest

from proxsuite_amd._ctypes_defs import InitialGuess, QPSolverOutput, HessianType, DenseBackend

EPS = 1e-9
DIMS = (10, 60, 110, 210)


def _solve(oracle, m, n, n_eq, n_in, guess=None, **kw):
    qp = oracle.QP(n, n_eq, n_in, **kw)
    qp.settings.eps_abs = EPS
    qp.settings.eps_rel = 0
    if guess is not None:
        qp.settings.initial_guess = guess
    qp.init(m.H, m.g, m.A, m.b, m.C, m.l, m.u)
    qp.solve()
    return qp


@pytest.mark.parametrize("dim", DIMS)
def test_strongly_convex_eq_and_in(oracle, randqp, dim):
    # dense_qp_with_eq_and_in.cpp:14-65
    randqp.set_seed(1)
    n_eq = n_in = dim // 4
    m = randqp.dense_strongly_convex_qp(dim, n_eq, n_in, 0.15, 1e-2)
    qp = _solve(oracle, m, dim, n_eq, n_in)
    pri, dua = oracle.kkt_residuals(m.H, m.g, m.A, m.b, m.C, m.l, m.u, qp.results.x, qp.results.y, qp.results.z)
    assert qp.results.info.status == QPSolverOutput.PROXQP_SOLVED
    assert pri <= EPS and dua <= EPS


@pytest.mark.parametrize("dim", DIMS)
def test_box_as_general_inequalities(oracle, randqp, dim):
    # dense_qp_with_eq_and_in.cpp:67-115
    randqp.set_seed(1)
    m = randqp.dense_box_constrained_qp(dim, 0, dim, 0.15, 1e-2)
    qp = _solve(oracle, m, dim, 0, dim)
    pri, dua = oracle.kkt_residuals(m.H, m.g, m.A, m.b, m.C, m.l, m.u, qp.results.x, qp.results.y, qp.results.z)
    assert pri <= EPS and dua <= EPS


@pytest.mark.parametrize("dim", DIMS)
def test_not_strongly_convex(oracle, randqp, dim):
    # dense_qp_with_eq_and_in.cpp:117-165
    randqp.set_seed(1)
    n_eq = n_in = dim // 2
    m = randqp.dense_not_strongly_convex_qp(dim, n_eq, n_in, 0.15)
    qp = _solve(oracle, m, dim, n_eq, n_in)
    pri, dua = oracle.kkt_residuals(m.H, m.g, m.A, m.b, m.C, m.l, m.u, qp.results.x, qp.results.y, qp.results.z)
    assert pri <= EPS and dua <= EPS


@pytest.mark.parametrize("dim", DIMS)
def test_degenerate(oracle, randqp, dim):
    # dense_qp_with_eq_and_in.cpp:167-221
    randqp.set_seed(1)
    m_ = dim // 4
    m = randqp.dense_degenerate_qp(dim, m_, m_, 0.15, 1e-2)
    qp = _solve(oracle, m, dim, m_, 2 * m_)
    pri, dua = oracle.kkt_residuals(m.H, m.g, m.A, m.b, m.C, m.l, m.u, qp.results.x, qp.results.y, qp.results.z)
    assert pri <= EPS and dua <= EPS


@pytest.mark.parametrize("dim", DIMS)
def test_linear_program(oracle, randqp, dim):
    # dense_qp_with_eq_and_in.cpp:223+ : H = 0, g chosen so the LP is bounded
    randqp.set_seed(1)
    n_eq = n_in = dim // 2
    m = randqp.dense_not_strongly_convex_qp(dim, n_eq, n_in, 0.15)
    x_sol = np.array([randqp.normal_rand() for _ in range(dim)])
    y_sol = np.array([randqp.normal_rand() for _ in range(n_eq)])
    z_sol = np.array([randqp.normal_rand() for _ in range(n_in)])
    m.H[:] = 0
    m.g[:] = -(m.A.T @ y_sol + m.C.T @ z_sol)
    del x_sol
    qp = _solve(oracle, m, dim, n_eq, n_in)
    pri, dua = oracle.kkt_residuals(m.H, m.g, m.A, m.b, m.C, m.l, m.u, qp.results.x, qp.results.y, qp.results.z)
    assert pri <= EPS and dua <= EPS


@pytest.mark.parametrize("dim", DIMS)
def test_equality_only(oracle, randqp, dim):
    # dense_qp_eq.cpp:14-60
    randqp.set_seed(1)
    n_eq = dim // 2
    m = randqp.dense_strongly_convex_qp(dim, n_eq, 0, 0.15, 1e-2)
    qp = _solve(oracle, m, dim, n_eq, 0)
    pri, dua = oracle.kkt_residuals(m.H, m.g, m.A, m.b, m.C, m.l, m.u, qp.results.x, qp.results.y, qp.results.z)
    assert pri <= EPS and dua <= EPS


@pytest.mark.parametrize("dim", DIMS)
def test_unconstrained(oracle, randqp, dim):
    # dense_unconstrained_qp.cpp:15-60
    randqp.set_seed(1)
    m = randqp.dense_unconstrained_qp(dim, 0.15, 1e-2)
    qp = _solve(oracle, m, dim, 0, 0)
    pri, dua = oracle.kkt_residuals(m.H, m.g, m.A, m.b, m.C, m.l, m.u, qp.results.x, qp.results.y, qp.results.z)
    assert dua <= EPS


@pytest.mark.parametrize("guess", list(InitialGuess))
def test_every_initial_guess_multi_solve(oracle, randqp, guess):
    # dense_qp_wrapper.cpp:1539-3927 (condensed): solve, re-solve, update g, re-solve
    randqp.set_seed(1)
    dim, n_eq, n_in = 30, 7, 9
    m = randqp.dense_strongly_convex_qp(dim, n_eq, n_in, 0.15, 1e-2)
    qp = _solve(oracle, m, dim, n_eq, n_in, guess=guess)
    for _ in range(2):
        pri, dua = oracle.kkt_residuals(m.H, m.g, m.A, m.b, m.C, m.l, m.u, qp.results.x, qp.results.y, qp.results.z)
        assert pri <= EPS and dua <= EPS
        qp.solve()
    g2 = m.g + 0.5
    qp.update(g=g2)
    qp.solve()
    pri, dua = oracle.kkt_residuals(m.H, g2, m.A, m.b, m.C, m.l, m.u, qp.results.x, qp.results.y, qp.results.z)
    assert pri <= EPS and dua <= EPS
    H2 = m.H + np.eye(dim)
    qp.update(H=H2, update_preconditioner=True)
    qp.solve()
    pri, dua = oracle.kkt_residuals(H2, g2, m.A, m.b, m.C, m.l, m.u, qp.results.x, qp.results.y, qp.results.z)
    assert pri <= EPS and dua <= EPS


def test_box_constraints_z_ordering(oracle, randqp):
    # dense_qp_wrapper.cpp:6803-6900 : z = [z_C ; z_box], 100 of the 1000 seeds, dim 15
    dim, n_eq, n_in = 15, 3, 4
    for seed in range(100):
        randqp.set_seed(seed)
        m = randqp.dense_strongly_convex_qp(dim, n_eq, n_in, 1.0, 1e-2)
        x_sol = np.array([randqp.normal_rand() for _ in range(dim)])
        delta = np.array([randqp.uniform_rand() for _ in range(n_in)])
        m.u[:] = m.C @ x_sol + delta
        m.b[:] = m.A @ x_sol
        shift = np.array([randqp.uniform_rand() for _ in range(dim)])
        u_box, l_box = x_sol + shift, x_sol - shift
        qp = oracle.QP(dim, n_eq, n_in, box_constraints=True)
        qp.settings.eps_abs = EPS
        qp.settings.eps_rel = 0
        qp.init(m.H, m.g, m.A, m.b, m.C, m.l, m.u, l_box, u_box)
        qp.solve()
        pri, dua = oracle.kkt_residuals(m.H, m.g, m.A, m.b, m.C, m.l, m.u, qp.results.x, qp.results.y,
                                        qp.results.z, l_box, u_box)
        assert pri <= EPS and dua <= EPS, (seed, pri, dua)


def test_diagonal_hessian_box(oracle, randqp):
    # benchmark/timings-diagonal-hessian.cpp:43-92 shape (smaller)
    dim = 40
    randqp.set_seed(1)
    m = randqp.dense_box_constrained_qp(dim, 0, dim, 0.15, 1e-2)
    H = np.diag(np.arange(1, dim + 1, dtype=float))
    qp = oracle.QP(dim, 0, 0, box_constraints=True, hessian_type=HessianType.Diagonal)
    qp.settings.eps_abs = EPS
    qp.settings.eps_rel = 0
    qp.init(H, m.g, None, None, None, None, None, m.l, m.u)
    qp.solve()
    z0 = np.zeros(0)
    pri, dua = oracle.kkt_residuals(H, m.g, None, None, None, z0, z0, qp.results.x, qp.results.y, qp.results.z,
                                    m.l, m.u)
    assert pri <= EPS and dua <= EPS


def test_parallel_equals_serial(oracle, randqp):
    # reference test/src/parallel_qp_solve.cpp:19-77 (bitwise on x), smaller dims
    dim, n_eq, n_in, B = 50, 5, 5, 16
    def make():
        qps = []
        for i in range(B):
            randqp.set_seed(i)
            m = randqp.dense_strongly_convex_qp(dim, n_eq, n_in, 0.15, 1e-2)
            qp = oracle.QP(dim, n_eq, n_in)
            qp.settings.eps_abs = EPS
            qp.settings.eps_rel = 0
            qp.init(m.H, m.g, m.A, m.b, m.C, m.l, m.u)
            qps.append(qp)
        return qps
    a, b = make(), make()
    for q in a:
        q.solve()
    oracle.solve_in_parallel(b, 4)
    for qa, qb in zip(a, b):
        assert np.array_equal(qa.results.x, qb.results.x)
