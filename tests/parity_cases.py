"""Parity cases shared by the CPU-emulated run (tests/test_emu_parity.py, `-m "not gpu"`) and
the real MI355X run (tests/test_gpu_parity.py, `-m gpu`).  Every case drives the C-ABI of
include/proxqp_hip.h and compares with the CPU oracle on the same seeded inputs.

Tolerances: unscaled KKT residuals <= eps_abs = 1e-9 is the hard gate (the
reference's own acceptance test, test/src/dense_qp_with_eq_and_in.cpp:46-56); (x, y, z) must
agree with the oracle to XYZ_TOL * (1 + |ref|_inf) = 1e-10 (observed: 1e-13 on random QPs); the
counters of Results::info (iter, iter_ext, mu_updates, rho_updates, status) must be EQUAL and
objValue / pri_res / dua_res / mu / rho close (INFO_*): the device runs the reference's iteration,
only its linear algebra engine differs, so the iterates follow the oracle's to rounding.  Cases on
ill-conditioned data (Maros-Meszaros, degenerate families) state their own, looser, numbers.
"""
import numpy as np

from proxsuite_amd import _native as N
from proxsuite_amd._ctypes_defs import HessianType, InitialGuess, QPSolverOutput

EPS = 1e-9
XYZ_TOL = 1e-10
INFO_REL = 1e-9   # objValue, mu_eq, mu_in, rho: relative
INFO_RES = 1e-11  # pri_res, dua_res, duality_gap: absolute, plus 1e-6 relative (they are differences at the rounding floor)


def kkt_numpy(H, g, A, b, Cm, l, u, x, y, z, l_box=None, u_box=None):
    """The reference's universal acceptance test (test/src/dense_qp_with_eq_and_in.cpp:46-56):
    primal and dual residual of (x, y, z) on the UNSCALED model, in plain numpy -- independent of
    both the device code and the oracle library."""
    n_in = Cm.shape[0] if Cm is not None and np.size(Cm) else 0
    pri = 0.0
    dua = H @ x + g
    if A is not None and np.size(A):
        pri = max(pri, float(np.max(np.abs(A @ x - b))))
        dua = dua + A.T @ y
    if n_in:
        Cx = Cm @ x
        pri = max(pri, float(np.max(np.abs(np.maximum(Cx - u, 0) + np.minimum(Cx - l, 0)))))
        dua = dua + Cm.T @ z[:n_in]
    if l_box is not None:
        pri = max(pri, float(np.max(np.abs(np.maximum(x - u_box, 0) + np.minimum(x - l_box, 0)))))
        dua = dua + z[n_in:]
    return pri, float(np.max(np.abs(dua)))


def kkt(oracle, m, i, x, y, z, l_box=None, u_box=None):
    pick = (lambda a: a[i]) if i is not None else (lambda a: a)
    return kkt_numpy(pick(m.H), pick(m.g), pick(m.A), pick(m.b), pick(m.C), pick(m.l), pick(m.u),
                     x, y, z, l_box, u_box)


def close(a, ref, tol=None):
    if ref.size == 0:
        return True
    return float(np.max(np.abs(a - ref))) <= (XYZ_TOL if tol is None else tol) * (1 + float(np.max(np.abs(ref))))


def info_close(dev, ref, counters=True, residuals=True):
    """Results::info of the device against the oracle's (reference results.hpp:28-76): the integer
    counters and the status are equal, the proximal parameters and the objective agree to INFO_REL, the
    residual norms to INFO_RES.  Returns a description of the first mismatch, or None."""
    if dev.status != ref.status:
        return "status %d != %d" % (dev.status, ref.status)
    if counters:
        for k in ("iter", "iter_ext", "mu_updates", "rho_updates"):
            if getattr(dev, k) != getattr(ref, k):
                return "%s %d != %d" % (k, getattr(dev, k), getattr(ref, k))
    for k in ("mu_eq", "mu_in", "rho", "objValue"):
        a, r = getattr(dev, k), getattr(ref, k)
        if abs(a - r) > INFO_REL * (1 + abs(r)):
            return "%s %.17g != %.17g" % (k, a, r)
    if residuals:
        for k in ("pri_res", "dua_res", "duality_gap"):
            a, r = getattr(dev, k), getattr(ref, k)
            if abs(a - r) > INFO_RES + 1e-6 * abs(r):
                return "%s %.6e != %.6e" % (k, a, r)
    return None


def settings_all(b, **kw):
    for i in range(b.B):
        s = b.settings(i)
        for k, v in kw.items():
            setattr(s, k, v)


def oracle_solve(oracle, m, i, n, ne, ni, guess, **qpkw):
    q = oracle.QP(n, ne, ni, **qpkw)
    q.settings.eps_abs = EPS
    q.settings.eps_rel = 0
    q.settings.initial_guess = guess
    q.init(m.H[i], m.g[i], m.A[i], m.b[i], m.C[i], m.l[i], m.u[i])
    q.solve()
    return q


def case_random_batch(lib, oracle, randqp, n, ne, ni, B, guess=InitialGuess.NO_INITIAL_GUESS, sparsity=0.15,
                      compare=True, info_residuals=True):
    """benchmark/timings-parallel.cpp:43-63 workload at arbitrary size."""
    m = randqp.dense_strongly_convex_qp_batch(B, n, ne, ni, sparsity, 1e-2)
    b = N.Batch(B, n, ne, ni, lib=lib)
    settings_all(b, eps_abs=EPS, eps_rel=0, initial_guess=int(guess))
    b.init(-1, m.H, m.g, m.A, m.b, m.C, m.l, m.u)
    b.solve()
    x, y, z, se, si, info = b.results()
    for i in range(B):
        assert info[i].status == QPSolverOutput.PROXQP_SOLVED, (i, info[i].status)
        pri, dua = kkt(oracle, m, i, x[i], y[i], z[i])
        assert pri <= EPS and dua <= EPS, (i, pri, dua)
    if compare:
        # EVERY QP of the batch, the oracle running under its solve_in_parallel (reference
        # parallel/qp_solve.hpp:17-39); an integer `compare` limits the comparison to the first ones
        idx = range(B) if compare is True or compare == "all" else range(min(B, int(compare)))
        qs = oracle_solve_many(oracle, [(m.H[i], m.g[i], m.A[i], m.b[i], m.C[i], m.l[i], m.u[i]) for i in idx],
                               n, ne, ni, guess)
        for i, q in zip(idx, qs):
            assert close(x[i], q.results.x) and close(y[i], q.results.y) and close(z[i], q.results.z), i
            # (info_residuals=False: shapes of thousands of rows, whose duality gap is a sum of thousands of terms of
            # size 1 that cancel to 1e-9 -- the two summation orders differ by 1e-11 there; everything else is compared)
            bad = info_close(info[i], q.results.info, residuals=info_residuals)
            assert bad is None, (i, bad)
    b.close()
    return x, y, z, info


def case_verbose_round_trip(lib, oracle, randqp, n=20, ne=5, ni=8, B=3):
    """settings.verbose is not only printing in the reference: its report block unscales x, y, z and scales them back
    (dense/solver.hpp:1469-1510), which perturbs the iterates in their last bits at every outer iteration.  Oracle and
    device both restate that round trip: a verbose run must agree like any other (solutions to 1e-10, Info counters)."""
    m = randqp.dense_strongly_convex_qp_batch(B, n, ne, ni, 0.15, 1e-2, seed0=40)
    b = N.Batch(B, n, ne, ni, lib=lib)
    settings_all(b, eps_abs=EPS, eps_rel=0, initial_guess=int(InitialGuess.NO_INITIAL_GUESS), verbose=1)
    b.init(-1, m.H, m.g, m.A, m.b, m.C, m.l, m.u)
    b.solve()
    x, y, z, se, si, info = b.results()
    for i in range(B):
        q = oracle.QP(n, ne, ni)
        q.settings.eps_abs, q.settings.eps_rel = EPS, 0
        q.settings.initial_guess = InitialGuess.NO_INITIAL_GUESS
        q.settings.verbose = 1
        q.init(m.H[i], m.g[i], m.A[i], m.b[i], m.C[i], m.l[i], m.u[i])
        q.solve()
        assert info[i].status == QPSolverOutput.PROXQP_SOLVED
        assert close(x[i], q.results.x) and close(y[i], q.results.y) and close(z[i], q.results.z), i
        bad = info_close(info[i], q.results.info)
        assert bad is None, (i, bad)
    b.close()


def case_verbose_trace(lib, oracle, randqp, capfd=None, n=20, ne=5, ni=8, B=4, box=False):
    """settings.verbose: the per-iteration lines of the reference's report (dense/solver.hpp:1478-1485 outer,
    :1021-1027 inner).  The device records them inside the solve kernel (pqp_batch_get_trace) and the library prints
    them when the launch has finished; the oracle records the same lines where the reference prints them.  Line by
    line: same sequence of outer / inner iterations with the same numbers k, residuals within the gate of the final Info
    (INFO_RES + 1e-6 relative), step lengths to 1e-6, mu_in and rho EQUAL -- the device follows the reference's path at every iteration, not only
    to the same end.  QP 1 is left non-verbose: no trace, no lines.  A second, warm-started solve replaces the trace."""
    m = randqp.dense_strongly_convex_qp_batch(B, n, ne, ni, 0.15, 1e-2, seed0=77)
    lb = ub = None
    if box:
        # a box around the solution of the QP without it: feasible by construction, for the second solve (other g) too,
        # where some of its sides become active.  (On an infeasible instance the two sides stagnate at the rounding
        # floor and leave their inner loops after different numbers of iterations: nothing to compare line by line.)
        rng = np.random.default_rng(5)
        xs = np.stack([q.results.x for q in oracle_solve_many(oracle, [(m.H[i], m.g[i], m.A[i], m.b[i], m.C[i], m.l[i], m.u[i])
                                                                        for i in range(B)], n, ne, ni)])
        sh = rng.uniform(0.02, 0.5, (B, n))
        lb, ub = xs - sh, xs + sh
    b = N.Batch(B, n, ne, ni, box_constraints=box, lib=lib)
    settings_all(b, eps_abs=EPS, eps_rel=0, initial_guess=int(InitialGuess.NO_INITIAL_GUESS), verbose=1)
    b.settings(1).verbose = 0
    kw = dict(l_box=lb, u_box=ub) if box else {}
    b.init(-1, m.H, m.g, m.A, m.b, m.C, m.l, m.u, **kw)
    qs = []
    for i in range(B):
        q = oracle.QP(n, ne, ni, box_constraints=box)
        q.settings.eps_abs, q.settings.eps_rel = EPS, 0
        q.settings.initial_guess = InitialGuess.NO_INITIAL_GUESS
        q.settings.verbose = 0 if i == 1 else 1
        q.init(m.H[i], m.g[i], m.A[i], m.b[i], m.C[i], m.l[i], m.u[i], **(dict(l_box=lb[i], u_box=ub[i]) if box else {}))
        qs.append(q)
    lines = truncated = 0
    for phase in (0, 1):
        if phase == 1:
            g2 = m.g * 1.25
            for i in range(B):
                b.settings(i).initial_guess = int(InitialGuess.WARM_START_WITH_PREVIOUS_RESULT)
                qs[i].settings.initial_guess = InitialGuess.WARM_START_WITH_PREVIOUS_RESULT
                qs[i].update(g=g2[i])
            b.update(-1, g=g2)
        if capfd is not None:
            capfd.readouterr()
        b.solve()
        out = capfd.readouterr().out if capfd is not None else None
        x, y, z, se, si, info = b.results()
        for i, q in enumerate(qs):
            q.solve()
            t, to = b.trace(i), q.trace()
            if i == 1:
                assert t.shape[0] == 0 and to.shape[0] == 0
                continue
            if t.shape[0] == 4095 and to.shape[0] > 4095:
                to = to[:4095]  # (an instance that runs long: the device keeps the first 4095 lines and counts the rest)
                truncated += 1
            assert t.shape == to.shape and t.shape[0] > 0, (phase, i, t.shape, to.shape)
            assert q.results.info.status == QPSolverOutput.PROXQP_SOLVED, (phase, i)
            assert np.array_equal(t[:, :2], to[:, :2]), (phase, i)  # same kinds, same iteration numbers
            outer = t[:, 0] == 1
            assert np.array_equal(t[outer, 5:7], to[outer, 5:7]), (phase, i)  # mu_in, rho
            tol = INFO_RES + 1e-6 * np.abs(to[:, 2:5])  # (the gate of the final Info residuals, info_close)
            # (the step length of an inner iteration is a ratio of two sums that both vanish at the solution: on the
            # last steps, with |dw| ~ 1e-9, two orders of summation give alpha = 1 +- 1e-8)
            tol[~outer, 1] = 1e-6 * (1 + np.abs(to[~outer, 3]))
            # (the inner residual of the last steps of a Newton loop sits at the rounding floor of sums of magnitude
            # |H| |x|: 1.3e-8 against 2.6e-9 on a C2-sized QP whose first inner residual is ~1e1; the floor scales with it)
            tol[~outer, 0] += 1e-8 * max(1.0, float(np.max(to[~outer, 2], initial=0.0)))
            err = np.abs(t[:, 2:5] - to[:, 2:5]) / tol
            k = np.unravel_index(int(np.argmax(err)), err.shape)
            assert np.max(err) <= 1.0, (phase, i, k, t[k[0]].tolist(), to[k[0]].tolist())
            # the trace is the solve: an outer line per outer iteration Info counts (+ the one that found the solution),
            # an inner line per inner iteration -- less the iterations the reference leaves before its print (the
            # |alpha dw| < 1e-11 exit of solver.hpp:944-951, an infeasibility exit)
            n_out, n_inn = int(outer.sum()), int((~outer).sum())
            assert 1 <= n_out <= info[i].iter_ext + 1 and n_inn <= info[i].iter, (phase, i, n_out, n_inn, info[i].iter_ext, info[i].iter)
            if info[i].status == QPSolverOutput.PROXQP_SOLVED:
                assert n_out == info[i].iter_ext + 1, (phase, i)
            ri = q.results.info
            assert (info[i].status, info[i].iter, info[i].iter_ext, info[i].mu_updates) == (ri.status, ri.iter, ri.iter_ext, ri.mu_updates)
            assert abs(info[i].objValue - ri.objValue) <= 1e-7 * (1 + abs(ri.objValue)), (phase, i)
            lines += t.shape[0]
        if out is not None:
            # the library printed exactly these lines, in the reference's wording, for the three verbose QPs
            n_outer = sum(int((b.trace(i)[:, 0] == 1).sum()) for i in range(B))
            n_inner = sum(int((b.trace(i)[:, 0] == 2).sum()) for i in range(B))
            assert out.count("[outer iteration ") == n_outer and out.count("| primal residual=") == n_outer
            assert out.count("[inner iteration ") == n_inner and out.count("| inner residual=") == n_inner
            assert out.count("SOLVER STATISTICS") == B - 1
            assert out.count("more iteration lines not recorded") == sum(1 for i in range(B) if b.trace(i).shape[0] == 4095)
            t0 = b.trace(0)
            first = "| primal residual=%.2e | dual residual=%.2e | duality gap=%.2e | mu_in=%.2e | rho=%.2e" % tuple(t0[0, 2:7])
            assert first in out, first
    b.close()
    return lines


def oracle_solve_many(oracle, models, n, ne, ni, guess=InitialGuess.NO_INITIAL_GUESS, eps=EPS, **qpkw):
    """init + solve_in_parallel of the oracle on a list of (H, g, A, b, C, l, u[, l_box, u_box])."""
    qs = []
    for mod in models:
        q = oracle.QP(n, ne, ni, **qpkw)
        q.settings.eps_abs = eps
        q.settings.eps_rel = 0
        q.settings.initial_guess = guess
        q.init(*mod)
        qs.append(q)
    oracle.solve_in_parallel(qs)
    return qs


def case_ruiz(lib, oracle, randqp, n=40, ne=20, ni=20):
    """reference test/src/dense_ruiz_equilibration.cpp:15-72 + agreement with the oracle."""
    m = randqp.dense_strongly_convex_qp_batch(2, n, ne, ni, 0.15, 1e-2)
    b = N.Batch(2, n, ne, ni, lib=lib)
    b.init(-1, m.H, m.g, m.A, m.b, m.C, m.l, m.u)
    b.flush()
    for i in range(2):
        s = b.scaled(i)
        d = s["delta"]
        D, E, F = d[:n], d[n:n + ne], d[n + ne:]
        c = s["c"]
        assert np.max(np.abs(s["H"] - c * (D[:, None] * m.H[i] * D[None, :]))) <= 1e-10
        assert np.max(np.abs(s["g"] - c * D * m.g[i])) <= 1e-10
        assert np.max(np.abs(s["A"] - E[:, None] * m.A[i] * D[None, :])) <= 1e-10
        assert np.max(np.abs(s["b"] - E * m.b[i])) <= 1e-10
        assert np.max(np.abs(s["C"] - F[:, None] * m.C[i] * D[None, :])) <= 1e-10
        q = oracle.QP(n, ne, ni)
        q.init(m.H[i], m.g[i], m.A[i], m.b[i], m.C[i], m.l[i], m.u[i])
        so = q.scaled()
        for k in ("H", "g", "A", "b", "C", "u", "delta"):
            assert np.max(np.abs(so[k] - s[k])) <= 1e-12 * (1 + np.max(np.abs(so[k]))), k
        assert so["c"] == s["c"]
    b.close()


def case_state_machine(lib, oracle, randqp, guess):
    """reference test/src/dense_qp_wrapper.cpp:1539-3927 condensed: solve, re-solve (dirty path),
    update g, update H with a new preconditioner, warm start -- mirrored call by call on the oracle."""
    n, ne, ni = 30, 7, 9
    B = 3
    m = randqp.dense_strongly_convex_qp_batch(B, n, ne, ni, 0.15, 1e-2)
    b = N.Batch(B, n, ne, ni, lib=lib)
    settings_all(b, eps_abs=EPS, eps_rel=0, initial_guess=int(guess))
    qs = []
    for i in range(B):
        q = oracle.QP(n, ne, ni)
        q.settings.eps_abs = EPS
        q.settings.eps_rel = 0
        q.settings.initial_guess = guess
        qs.append(q)

    def check(H, g):
        x, y, z, se, si, info = b.results()
        for i in range(B):
            pri, dua = kkt_numpy(H[i], g[i], m.A[i], m.b[i], m.C[i], m.l[i], m.u[i], x[i], y[i], z[i])
            assert pri <= EPS and dua <= EPS, (i, pri, dua)
            r = qs[i].results
            assert close(x[i], r.x) and close(y[i], r.y) and close(z[i], r.z), i
            bad = info_close(info[i], r.info)
            assert bad is None, (i, bad)

    b.init(-1, m.H, m.g, m.A, m.b, m.C, m.l, m.u)
    for i, q in enumerate(qs):
        q.init(m.H[i], m.g[i], m.A[i], m.b[i], m.C[i], m.l[i], m.u[i])
    for _ in range(2):  # second pass = dirty re-solve
        b.solve()
        for q in qs:
            q.solve()
        check(m.H, m.g)
    g2 = m.g + 0.5
    b.update(-1, g=g2)
    for i, q in enumerate(qs):
        q.update(g=g2[i])
    b.solve()
    for q in qs:
        q.solve()
    check(m.H, g2)
    H2 = m.H + np.eye(n)[None]
    b.update(-1, H=H2, update_preconditioner=True)
    for i, q in enumerate(qs):
        q.update(H=H2[i], update_preconditioner=True)
    b.solve()
    for q in qs:
        q.solve()
    check(H2, g2)
    # mu / rho update
    b.update(-1, rho=1e-7, mu_eq=1e-4)
    for q in qs:
        q.update(rho=1e-7, mu_eq=1e-4)
    b.solve()
    for q in qs:
        q.solve()
    check(H2, g2)
    # explicit warm start from a perturbed solution (solve(x, y, z))
    x, y, z, *_ = b.results()
    b.warm_start(-1, x + 1e-3, y, z)
    for i, q in enumerate(qs):
        q.solve(x[i] + 1e-3, y[i], z[i])
    b.solve()
    check(H2, g2)
    for i in range(B):
        assert b.settings(i).initial_guess == InitialGuess.WARM_START
    b.close()


def case_known_answers(lib):
    """reference test/src/cvxpy.cpp:61-160 through the C-ABI."""
    b = N.Batch(1, 1, 0, 1, lib=lib)
    b.settings(0).eps_abs = 1e-8
    H, g, C, l, u = np.array([[20.0]]), np.array([-10.0]), np.array([[1.0]]), np.array([0.0]), np.array([1.0])
    b.init(0, H, g, None, None, C, l, u)
    b.solve()
    x, y, z, se, si, info = b.results(0)
    assert info.status == QPSolverOutput.PROXQP_SOLVED
    assert abs(x[0] - 0.5) <= 1e-8
    # start from the solution: no iteration needed
    b2 = N.Batch(1, 1, 0, 1, lib=lib)
    b2.settings(0).eps_abs = 1e-8
    b2.init(0, H, g, None, None, C, l, u)
    b2.warm_start(0, np.array([0.5]), None, np.array([0.0]))
    b2.solve()
    x, y, z, se, si, info = b2.results(0)
    assert info.iter <= 0 and abs(x[0] - 0.5) <= 1e-8
    b.close()
    b2.close()


def case_box_constraints(lib, oracle, randqp, seeds=20, hessian=HessianType.Dense):
    """reference test/src/dense_qp_wrapper.cpp:6803-6900: z = [z_C; z_box]."""
    dim, n_eq, n_in = 15, 3, 4
    H = np.zeros((seeds, dim, dim))
    g = np.zeros((seeds, dim))
    A = np.zeros((seeds, n_eq, dim))
    bb = np.zeros((seeds, n_eq))
    Cm = np.zeros((seeds, n_in, dim))
    l = np.zeros((seeds, n_in))
    u = np.zeros((seeds, n_in))
    lb = np.zeros((seeds, dim))
    ub = np.zeros((seeds, dim))
    for s in range(seeds):
        randqp.set_seed(s)
        m = randqp.dense_strongly_convex_qp(dim, n_eq, n_in, 1.0, 1e-2)
        x_sol = np.array([randqp.normal_rand() for _ in range(dim)])
        delta = np.array([randqp.uniform_rand() for _ in range(n_in)])
        shift = np.array([randqp.uniform_rand() for _ in range(dim)])
        Hs = m.H if hessian == HessianType.Dense else np.diag(np.diag(m.H))
        H[s], g[s], A[s], Cm[s], l[s] = Hs, m.g, m.A, m.C, m.l
        u[s] = m.C @ x_sol + delta
        bb[s] = m.A @ x_sol
        ub[s], lb[s] = x_sol + shift, x_sol - shift
    b = N.Batch(seeds, dim, n_eq, n_in, box_constraints=True, hessian_type=int(hessian), lib=lib)
    settings_all(b, eps_abs=EPS, eps_rel=0)
    b.init(-1, H, g, A, bb, Cm, l, u, lb, ub)
    b.solve()
    x, y, z, se, si, info = b.results()
    for s in range(seeds):
        pri, dua = kkt_numpy(H[s], g[s], A[s], bb[s], Cm[s], l[s], u[s], x[s], y[s], z[s], lb[s], ub[s])
        assert pri <= EPS and dua <= EPS, (s, pri, dua)
        q = oracle.QP(dim, n_eq, n_in, box_constraints=True, hessian_type=hessian)
        q.settings.eps_abs = EPS
        q.settings.eps_rel = 0
        q.init(H[s], g[s], A[s], bb[s], Cm[s], l[s], u[s], lb[s], ub[s])
        q.solve()
        assert close(x[s], q.results.x) and close(y[s], q.results.y) and close(z[s], q.results.z), s
        bad = info_close(info[s], q.results.info)
        assert bad is None, (s, bad)
    b.close()


def case_families(lib, oracle, randqp, dim):
    """reference test/src/dense_qp_with_eq_and_in.cpp: not strongly convex, degenerate, LP;
    dense_qp_eq.cpp (equality only); dense_unconstrained_qp.cpp."""
    def solve1(m, n, ne, ni, hessian=HessianType.Dense):
        b = N.Batch(1, n, ne, ni, hessian_type=int(hessian), lib=lib)
        settings_all(b, eps_abs=EPS, eps_rel=0)
        b.init(0, m.H, m.g, m.A if ne else None, m.b if ne else None, m.C if ni else None,
               m.l if ni else None, m.u if ni else None)
        b.solve()
        x, y, z, se, si, info = b.results(0)
        pri, dua = kkt_numpy(m.H, m.g, m.A, m.b, m.C, m.l, m.u, x, y, z)
        assert pri <= EPS and dua <= EPS, (pri, dua, info.status)
        # against the oracle: x and the Info counters (the multipliers of the degenerate / not strongly
        # convex families are not unique in exact arithmetic, but the iteration is the same one)
        q = oracle.QP(n, ne, ni, hessian_type=hessian)
        q.settings.eps_abs, q.settings.eps_rel = EPS, 0
        q.init(m.H, m.g, m.A if ne else None, m.b if ne else None, m.C if ni else None,
               m.l if ni else None, m.u if ni else None)
        q.solve()
        assert close(x, q.results.x, 1e-8), np.max(np.abs(x - q.results.x))
        # (the residual norms at the solution of these rank-deficient families are what the linear solves leave
        # behind -- 1e-10 vs 4e-10 between two engines, both below eps and KKT-gated above: not compared)
        bad = info_close(info, q.results.info, residuals=False)
        assert bad is None, bad
        b.close()

    randqp.set_seed(1)
    solve1(randqp.dense_not_strongly_convex_qp(dim, dim // 2, dim // 2, 0.15), dim, dim // 2, dim // 2)
    randqp.set_seed(1)
    solve1(randqp.dense_degenerate_qp(dim, dim // 4, dim // 4, 0.15, 1e-2), dim, dim // 4, 2 * (dim // 4))
    randqp.set_seed(1)
    solve1(randqp.dense_box_constrained_qp(dim, 0, dim, 0.15, 1e-2), dim, 0, dim)
    randqp.set_seed(1)
    solve1(randqp.dense_strongly_convex_qp(dim, dim // 2, 0, 0.15, 1e-2), dim, dim // 2, 0)
    randqp.set_seed(1)
    solve1(randqp.dense_unconstrained_qp(dim, 0.15, 1e-2), dim, 0, 0)
    # LP (HessianType::Zero)
    randqp.set_seed(1)
    m = randqp.dense_not_strongly_convex_qp(dim, dim // 2, dim // 2, 0.15)
    y_sol = np.array([randqp.normal_rand() for _ in range(dim // 2)])
    z_sol = np.array([randqp.normal_rand() for _ in range(dim // 2)])
    m.H[:] = 0
    m.g[:] = -(m.A.T @ y_sol + m.C.T @ z_sol)
    solve1(m, dim, dim // 2, dim // 2)
    solve1(m, dim, dim // 2, dim // 2, hessian=HessianType.Zero)


def case_maros_meszaros(lib, P, q, A, l, u):
    """reference test/src/dense_maros_meszaros.cpp:85-169."""
    from conftest import split_maros
    H, g, Aeq, b, C, lin, uin = split_maros(P, q, A, l, u)
    n, n_eq, n_in = H.shape[0], Aeq.shape[0], C.shape[0]
    eps = 2e-8
    bt = N.Batch(1, n, n_eq, n_in, lib=lib)
    bt.init(0, H, g, Aeq, b, C, lin, uin)
    s = bt.settings(0)
    s.eps_abs = eps
    s.eps_rel = 0
    s.eps_primal_inf = 1e-12
    s.eps_dual_inf = 1e-12
    s.max_iter = 1000  # the reference converges in < 30 outer iterations; bounds a bad run on the GPU
    for it in range(2):
        if it > 0:
            s.initial_guess = InitialGuess.WARM_START_WITH_PREVIOUS_RESULT
        bt.solve()
        x, y, z, se, si, info = bt.results(0)
        dua = H @ x + g
        mag = np.abs(H) @ np.abs(x) + np.abs(g)  # size of the terms that cancel in `dua`
        if n_eq:
            dua = dua + Aeq.T @ y
            mag = mag + np.abs(Aeq.T) @ np.abs(y)
            assert np.max(np.abs(Aeq @ x - b)) < eps * 1.0001
        if n_in:
            dua = dua + C.T @ z
            mag = mag + np.abs(C.T) @ np.abs(z)
            assert (C @ x - lin).min() > -eps
            assert (C @ x - uin).max() < eps
        # the reference's acceptance line (2 * eps) plus the fp64 floor of evaluating the residual:
        # QPCBOEI2 has multipliers of 1e8, its terms reach 2.5e8 and cancel to 1e-8 -- one ulp of
        # them is 5.6e-8, so the recomputed residual moves by a few 1e-8 with the summation order
        # while the solver's own criterion is met (info.dua_res = 2e-10 on that problem)
        assert np.max(np.abs(dua)) < 2 * eps + 4 * np.finfo(float).eps * np.max(mag)
        assert info.dua_res <= eps and info.pri_res <= eps
        if it > 0:
            assert info.iter == 0
    bt.close()


def case_maros_meszaros_path(lib, oracle, problems):
    """The PATH on the Maros-Meszaros problems (settings of reference test/src/dense_maros_meszaros.cpp:85-169, plus
    verbose): the device's per-iteration trace against the oracle's.  Returns (names whose traces have the same
    sequence of outer / inner iterations with equal mu_in / rho on every outer line, {name: (first differing line,
    lines)} of the others).  Every problem must end SOLVED on both sides with the same number of mu updates; where
    the paths fork (the degenerate LP-like Q* problems: their first KKT system is ill-conditioned -- QAFIRO starts with a
    primal residual of 8e6 -- and the block-elimination engine of the device and the LDL^T of the assembled KKT matrix
    of the oracle solve it to the refinement tolerance, 1e-7 relative apart, not to the same bits), the iteration counts
    stay within 10 % of each other."""
    from conftest import split_maros
    import os
    same, forks = [], {}
    devnull, saved = os.open(os.devnull, os.O_WRONLY), os.dup(1)  # (the library prints the lines it recorded)
    try:
        for name, (P, q, A, l, u) in problems.items():
            H, g, Aeq, b, C, lin, uin = split_maros(P, q, A, l, u)
            n, ne, ni = H.shape[0], Aeq.shape[0], C.shape[0]
            bt = N.Batch(1, n, ne, ni, lib=lib)
            bt.init(0, H, g, Aeq, b, C, lin, uin)
            qo = oracle.QP(n, ne, ni)
            for st in (bt.settings(0), qo.settings):
                st.eps_abs, st.eps_rel, st.eps_primal_inf, st.eps_dual_inf, st.max_iter, st.verbose = 2e-8, 0, 1e-12, 1e-12, 1000, 1
            qo.init(H, g, Aeq, b, C, lin, uin)
            os.dup2(devnull, 1)
            try:
                bt.solve()
            finally:
                os.dup2(saved, 1)
            qo.solve()
            t, to = bt.trace(0), qo.trace()
            info, ri = bt.results(0)[5], qo.results.info
            assert info.status == ri.status == QPSolverOutput.PROXQP_SOLVED, (name, info.status, ri.status)
            m = min(len(t), len(to))
            eq = np.all(t[:m, :2] == to[:m, :2], axis=1)
            if len(t) == len(to) and eq.all():
                outer = t[:, 0] == 1
                assert np.array_equal(t[outer, 5:7], to[outer, 5:7]), name
                assert (info.iter, info.iter_ext, info.mu_updates) == (ri.iter, ri.iter_ext, ri.mu_updates), name
                same.append(name)
            else:
                forks[name] = (int(np.argmin(eq)) if not eq.all() else m, len(to))
                assert abs(info.iter - ri.iter) <= max(2, 0.1 * ri.iter), (name, info.iter, ri.iter)
            bt.close()
    finally:
        os.close(devnull)
        os.close(saved)
    return same, forks


def case_errors(lib):
    """reference error convention: std::invalid_argument -> ValueError."""
    import pytest
    with pytest.raises(ValueError):
        N.Batch(1, 0, 0, 0, lib=lib)  # dense/model.hpp:65-68
    b = N.Batch(2, 5, 1, 2, lib=lib)
    with pytest.raises(ValueError):
        b.init(0, np.zeros((4, 4)))  # wrong size
    with pytest.raises(ValueError):
        b.init(0, np.eye(5), l_box=np.zeros(5), u_box=np.ones(5))  # wrapper.hpp:542-546
    with pytest.raises(ValueError):
        b.init(5, np.eye(5))
    b.close()


def case_determinism(lib, randqp, n=20, ne=5, ni=8, B=6):
    """reference test/src/parallel_qp_solve.cpp:74-76: bitwise-equal x between two runs."""
    m = randqp.dense_strongly_convex_qp_batch(B, n, ne, ni, 0.15, 1e-2)
    out = []
    for _ in range(2):
        b = N.Batch(B, n, ne, ni, lib=lib)
        settings_all(b, eps_abs=EPS, eps_rel=0)
        b.init(-1, m.H, m.g, m.A, m.b, m.C, m.l, m.u)
        b.solve()
        out.append(b.results()[0].copy())
        b.close()
    assert np.array_equal(out[0], out[1])


def case_launch_size_invariance(lib, randqp, n, ne, ni, B, chunk, box=False, dense_backend=0, exact=True):
    """A QP's result must not depend on how many QPs share its launch: device-filling launches take
    the 128-VGPR builds of the solve kernels (four / two workgroups per CU), small ones the builds with
    the larger register budget (csrc/pqp_kernels.hip) -- same arithmetic in the same order, so the
    results are compared bit for bit."""
    m = randqp.dense_strongly_convex_qp_batch(B, n, ne, ni, 0.15, 1e-2)
    kw = {}
    if box:
        rng = np.random.default_rng(3)
        kw = dict(l_box=-1.0 - rng.random((B, n)), u_box=1.0 + rng.random((B, n)))
    out = []
    for step in (B, chunk):
        b = N.Batch(B, n, ne, ni, box_constraints=box, dense_backend=dense_backend, lib=lib)
        settings_all(b, eps_abs=EPS, eps_rel=0)
        b.init(-1, m.H, m.g, m.A, m.b, m.C, m.l, m.u, **kw)
        for first in range(0, B, step):
            b.solve(first, min(step, B - first))
        out.append([a.copy() for a in b.results()[:3]] +
                   [np.array([(i.status, i.iter, i.iter_ext, i.mu_updates, i.rho_updates) for i in b.infos()])])
        b.close()
    if exact:
        for a, c in zip(out[0], out[1]):
            assert np.array_equal(a, c)
    else:
        # two kernels that sum in different orders (the launch size picks one): same decisions, values to rounding
        assert np.array_equal(out[0][3], out[1][3])
        for a, c in zip(out[0][:3], out[1][:3]):
            assert np.all(np.abs(a - c) <= 1e-10 * (1.0 + np.abs(c)))
    return out[0]


def case_backward(lib, oracle, randqp, n=10, ne=4, ni=7, B=6, with_dual_terms=True):
    """QPLayer backward (reference dense/compute_ECJ.hpp:29-189; test/src/dense_backward.cpp): the
    seven loss jacobians of a solved batch against the oracle's literal restatement, with loss
    derivatives on x only and on (x, y, z)."""
    m = randqp.dense_strongly_convex_qp_batch(B, n, ne, ni, 0.85, 1e-1)
    b = N.Batch(B, n, ne, ni, lib=lib)
    settings_all(b, eps_abs=EPS, eps_rel=0)
    b.init(-1, m.H, m.g, m.A, m.b, m.C, m.l, m.u)
    b.solve()
    rng = np.random.default_rng(5)
    ntot = n + ne + ni
    for variant in range(2 if with_dual_terms else 1):
        ld = np.zeros((B, ntot))
        ld[:, :n] = rng.standard_normal((B, n))
        if variant == 1:
            ld[:, n:] = rng.standard_normal((B, ne + ni))
        b.backward(ld, 1e-5, 1e-7, 1e-7)
        got = b.backward_results(-1)
        for i in range(B):
            q = oracle.QP(n, ne, ni)
            q.settings.eps_abs = EPS
            q.settings.eps_rel = 0
            q.init(m.H[i], m.g[i], m.A[i], m.b[i], m.C[i], m.l[i], m.u[i])
            q.solve()
            ref = q.compute_backward(ld[i], 1e-5, 1e-7, 1e-7)
            for k, v in ref.items():
                scale = 1 + (np.max(np.abs(v)) if v.size else 0.0)
                assert np.max(np.abs(got[k][i] - v), initial=0.0) <= 1e-6 * scale, (variant, i, k)
    # one QP of the batch alone (compute_backward on qps.get(i)) gives the same rows
    b.solve()
    ld1 = np.zeros((1, ntot))
    ld1[0, :n] = 1.0
    b.backward(ld1, 1e-5, 1e-7, 1e-7, first=2, count=1)
    one = b.backward_results(2)
    ldB = np.zeros((B, ntot))
    ldB[:, :n] = 1.0
    b.solve()
    b.backward(ldB, 1e-5, 1e-7, 1e-7)
    allr = b.backward_results(-1)
    for k in one:
        assert np.array_equal(one[k], allr[k][2]), k
    b.close()


def case_c4_shape(lib, oracle, randqp, B=8):
    """BASELINE.json configs[3] at its real shape: n=512, n_eq=200, n_in=400 (1024-thread
    workgroups, blocked matrix-core LDL^T, row-wise triangular inverse), every QP against the oracle."""
    case_random_batch(lib, oracle, randqp, 512, 200, 400, B=B, compare="all")


def case_c3_one_launch(lib, oracle, randqp, B=16384, chunk=2048, n=100, ne=50, ni=100):
    """BASELINE.json configs[2] on ONE GPU: the 16 384 QPs of the 8-GPU configuration (n=100, n_eq=50, n_in=100,
    seeds 0 .. B-1: rank r of the sharded run owns the seeds r B/8 ...) in one launch of one handle -- the N = 1 point of
    the strong-scaling curve.  The full gate on EVERY QP: SOLVED, unscaled KKT residuals <= 1e-9 in numpy, (x, y, z)
    against the oracle to 1e-10 and equal Info counters (the oracle runs chunk by chunk to bound host memory)."""
    m = randqp.dense_strongly_convex_qp_batch(B, n, ne, ni, 0.15, 1e-2)
    b = N.Batch(B, n, ne, ni, lib=lib)
    b.set_all_settings(eps_abs=EPS, eps_rel=0.0, initial_guess=int(InitialGuess.NO_INITIAL_GUESS))
    b.init(-1, m.H, m.g, m.A, m.b, m.C, m.l, m.u)
    b.solve()
    ms = b.last_solve_ms
    x, y, z, se, si, info = b.results()
    st = np.array([info[i].status for i in range(B)])
    assert np.all(st == int(QPSolverOutput.PROXQP_SOLVED)), np.nonzero(st)[0][:10]
    dua = np.einsum("bij,bj->bi", m.H, x) + m.g + np.einsum("bij,bi->bj", m.A, y) + np.einsum("bij,bi->bj", m.C, z)
    Cx = np.einsum("bij,bj->bi", m.C, x)
    pri = np.maximum(np.abs(np.einsum("bij,bj->bi", m.A, x) - m.b).max(axis=1),
                     np.abs(np.maximum(Cx - m.u, 0) + np.minimum(Cx - m.l, 0)).max(axis=1))
    worst = float(max(pri.max(), np.abs(dua).max()))
    assert worst <= EPS, worst
    worst_delta = 0.0
    for lo in range(0, B, chunk):
        idx = range(lo, min(B, lo + chunk))
        qs = oracle_solve_many(oracle, [(m.H[i], m.g[i], m.A[i], m.b[i], m.C[i], m.l[i], m.u[i]) for i in idx], n, ne, ni)
        for i, q in zip(idx, qs):
            assert close(x[i], q.results.x) and close(y[i], q.results.y) and close(z[i], q.results.z), i
            bad = info_close(info[i], q.results.info)
            assert bad is None, (i, bad)
            worst_delta = max(worst_delta, float(np.max(np.abs(x[i] - q.results.x))), float(np.max(np.abs(z[i] - q.results.z))))
        del qs
    b.close()
    return dict(qps=B, kernel_ms=ms, qps_per_s=B / (ms * 1e-3), max_kkt=worst, max_abs_delta_vs_oracle=worst_delta)


def c5_models(randqp, B, dim=200, seed0=0):
    """BASELINE.json configs[4]: dense_box_constrained_qp(dim, 0, dim) (reference
    utils/random_qp_problems.hpp:591-628) with H <- diag(H) (benchmark/timings-diagonal-hessian.cpp:43-56)."""
    H = np.zeros((B, dim, dim))
    g = np.zeros((B, dim))
    Cm = np.zeros((B, dim, dim))
    l = np.zeros((B, dim))
    u = np.zeros((B, dim))
    for s in range(B):
        randqp.set_seed(seed0 + s)
        m = randqp.dense_box_constrained_qp(dim, 0, dim, 0.15, 1e-2)
        H[s] = np.diag(np.diag(m.H))
        g[s], Cm[s], l[s], u[s] = m.g, m.C, m.l, m.u
    return H, g, Cm, l, u


def case_c5(lib, oracle, randqp, B, sample, box, dim=200):
    """BASELINE.json configs[4] through both of its forms: `box=False` passes the bounds as general
    inequalities C = I (timings-box-constraints shape), `box=True` through the box-constraint feature
    (timings-diagonal-hessian.cpp:72-92).  Every QP is KKT-gated in numpy, `sample` of them are
    compared with the oracle; the two forms must agree with each other as well."""
    H, g, Cm, l, u = c5_models(randqp, B, dim)
    hess = HessianType.Diagonal
    if box:
        b = N.Batch(B, dim, 0, 0, box_constraints=True, hessian_type=int(hess), lib=lib)
        settings_all(b, eps_abs=EPS, eps_rel=0, initial_guess=int(InitialGuess.NO_INITIAL_GUESS))
        b.init(-1, H, g, None, None, None, None, None, l, u)
    else:
        b = N.Batch(B, dim, 0, dim, hessian_type=int(hess), lib=lib)
        settings_all(b, eps_abs=EPS, eps_rel=0, initial_guess=int(InitialGuess.NO_INITIAL_GUESS))
        b.init(-1, H, g, None, None, Cm, l, u)
    b.solve()
    x, y, z, se, si, info = b.results()
    z0 = np.zeros(0)
    for s in range(B):
        assert info[s].status == QPSolverOutput.PROXQP_SOLVED, (s, info[s].status)
        if box:
            pri, dua = kkt_numpy(H[s], g[s], None, None, None, z0, z0, x[s], y[s], z[s], l[s], u[s])
        else:
            pri, dua = kkt_numpy(H[s], g[s], None, None, Cm[s], l[s], u[s], x[s], y[s], z[s])
        assert pri <= EPS and dua <= EPS, (s, pri, dua)
    idx = list(range(min(sample, B)))
    if box:
        qs = oracle_solve_many(oracle, [(H[i], g[i], None, None, None, None, None, l[i], u[i]) for i in idx],
                               dim, 0, 0, box_constraints=True, hessian_type=hess)
    else:
        qs = oracle_solve_many(oracle, [(H[i], g[i], None, None, Cm[i], l[i], u[i]) for i in idx],
                               dim, 0, dim, hessian_type=hess)
    for i, q in zip(idx, qs):
        assert close(x[i], q.results.x) and close(z[i], q.results.z), i
        bad = info_close(info[i], q.results.info)
        assert bad is None, (i, bad)
    b.close()
    return x, z


def case_diag_mixed_handle(lib, oracle, randqp, dim=24, B=6):
    """A handle whose QP 0 has a GENERAL constraint matrix beside QPs in diagonal structure: a range / subset launch of the
    structured ones takes the one-wavefront diagonal kernel (pqp_diag_dispatch looks at the QPs of the launch), the
    general one and a whole-batch launch the 256-thread kernel -- same HBM state either way; every QP against the oracle
    after each launch."""
    H, g, Cm, l, u = c5_models(randqp, B, dim)
    Cm[0][0, 1] = 0.3  # QP 0: not one variable per row any more
    hess = HessianType.Diagonal
    b = N.Batch(B, dim, 0, dim, hessian_type=int(hess), lib=lib)
    settings_all(b, eps_abs=EPS, eps_rel=0, initial_guess=int(InitialGuess.NO_INITIAL_GUESS))
    b.init(-1, H, g, None, None, Cm, l, u)
    qs = oracle_solve_many(oracle, [(H[i], g[i], None, None, Cm[i], l[i], u[i]) for i in range(B)], dim, 0, dim, hessian_type=hess)

    def check(which):
        x, y, z, se, si, info = b.results()
        for i in which:
            assert info[i].status == QPSolverOutput.PROXQP_SOLVED, (i, info[i].status)
            assert close(x[i], qs[i].results.x) and close(z[i], qs[i].results.z), i
            bad = info_close(info[i], qs[i].results.info)
            assert bad is None, (i, bad)

    b.solve(1, B - 1)          # a range of structured QPs
    check(range(1, B))
    b.solve(0, 1)              # the general one alone
    check([0])
    b.solve_subset([2, 4])     # a subset of structured QPs
    check([2, 4])
    b.solve()                  # everything in one launch: the general kernel
    check(range(B))
    b.close()


def case_diag_wave_flows(lib, oracle, randqp, dim, box, hessian=HessianType.Diagonal, merit=0, B=3, constrained=True):
    """The one-wavefront diagonal-structure kernel (proxsuite_amd/csrc/pqp_diag.hpp) through every entry of the solve
    state machine, mirrored call by call on the oracle: cold solve, dirty re-solve (stored equilibration re-applied),
    update(g) + WARM_START_WITH_PREVIOUS_RESULT (factor, slot list and D_S restored from HBM), explicit warm start,
    COLD_START_WITH_PREVIOUS_RESULT, update of mu_in / rho, EQUALITY_CONSTRAINED_INITIAL_GUESS, a verbose solve with its
    per-iteration trace.  C = I form or box form, diagonal or zero Hessian, GPDAL or PDAL merit function;
    `constrained=False`: no inequality at all.  Every result against the oracle (solutions to XYZ_TOL, Info counters
    equal) and KKT-gated in numpy."""
    H, g, Cm, l, u = c5_models(randqp, B, dim, seed0=7)
    if hessian == HessianType.Zero:
        H = np.zeros_like(H)
    ni = 0 if (box or not constrained) else dim
    useb = box and constrained

    def margs(i=None, gg=None):
        pk = (lambda a: a) if i is None else (lambda a: a[i])
        gv = g if gg is None else gg
        a = [pk(H), pk(gv), None, None, pk(Cm) if ni else None, pk(l) if ni else None, pk(u) if ni else None]
        kw = dict(l_box=pk(l), u_box=pk(u)) if useb else {}
        return a, kw

    def make(guess):
        b = N.Batch(B, dim, 0, ni, box_constraints=useb, hessian_type=int(hessian), lib=lib)
        settings_all(b, eps_abs=EPS, eps_rel=0, initial_guess=int(guess), merit_function_type=merit)
        qs = []
        for i in range(B):
            q = oracle.QP(dim, 0, ni, box_constraints=useb, hessian_type=hessian)
            q.settings.eps_abs, q.settings.eps_rel, q.settings.initial_guess = EPS, 0, guess
            q.settings.merit_function_type = merit
            qs.append(q)
        a, kw = margs()
        b.init(-1, *a, **kw)
        for i, q in enumerate(qs):
            a, kw = margs(i)
            q.init(*a, **kw)
        return b, qs

    forked = [0]

    def check(b, qs, gg, what):
        x, y, z, se, si, info = b.results()
        for i, q in enumerate(qs):
            r = q.results
            assert info[i].status == r.info.status == QPSolverOutput.PROXQP_SOLVED, (what, i, info[i].status, r.info.status)
            if useb:
                pri, dua = kkt_numpy(H[i], gg[i], None, None, None, np.zeros(0), np.zeros(0), x[i], y[i], z[i], l[i], u[i])
            elif ni:
                pri, dua = kkt_numpy(H[i], gg[i], None, None, Cm[i], l[i], u[i], x[i], y[i], z[i])
            else:
                pri, dua = 0.0, float(np.max(np.abs(H[i] @ x[i] + gg[i])))
            assert pri <= EPS and dua <= EPS, (what, i, pri, dua)
            # (zero Hessian + EQUALITY_CONSTRAINED_INITIAL_GUESS: the guess is -g / rho ~ 1e6, and the duality gap at the end a
            # difference of terms of that size -- 1e-9 apart between two orders of summation: counters and solutions only)
            loose = hessian == HessianType.Zero and what.startswith("equality")
            same = info_close(info[i], r.info, residuals=(merit == 0 and not loose)) is None
            if merit == 1 and not same:  # (PDAL: phi' jumps at the breakpoints, see case_random_sweep)
                forked[0] += 1
                assert close(x[i], r.x, 1e-5), (what, i)
                continue
            assert close(x[i], r.x) and close(z[i], r.z), (what, i, float(np.max(np.abs(x[i] - r.x))))
            assert same, (what, i, info_close(info[i], r.info, residuals=not loose))

    b, qs = make(InitialGuess.NO_INITIAL_GUESS)
    for what in ("cold", "dirty re-solve"):
        b.solve()
        oracle.solve_in_parallel(qs)
        check(b, qs, g, what)
    g2 = g + 0.3 * np.random.default_rng(3).standard_normal(g.shape)
    for i, q in enumerate(qs):
        b.settings(i).initial_guess = int(InitialGuess.WARM_START_WITH_PREVIOUS_RESULT)
        q.settings.initial_guess = InitialGuess.WARM_START_WITH_PREVIOUS_RESULT
        q.update(g=g2[i])
    b.update(-1, g=g2)
    for what in ("update g + previous result", "previous result again"):
        b.solve()
        oracle.solve_in_parallel(qs)
        check(b, qs, g2, what)
    x, y, z, *_ = b.results()
    b.warm_start(-1, x + 1e-3, y, 0.5 * z)
    for i, q in enumerate(qs):
        q.solve(x[i] + 1e-3, y[i], 0.5 * z[i])
    b.solve()
    check(b, qs, g2, "explicit warm start")
    for i, q in enumerate(qs):
        b.settings(i).initial_guess = int(InitialGuess.COLD_START_WITH_PREVIOUS_RESULT)
        q.settings.initial_guess = InitialGuess.COLD_START_WITH_PREVIOUS_RESULT
    b.solve()
    oracle.solve_in_parallel(qs)
    check(b, qs, g2, "cold start with previous result")
    b.update(-1, rho=1e-7, mu_in=1e-2)
    for i, q in enumerate(qs):
        q.update(rho=1e-7, mu_in=1e-2)
        b.settings(i).initial_guess = int(InitialGuess.WARM_START_WITH_PREVIOUS_RESULT)
        q.settings.initial_guess = InitialGuess.WARM_START_WITH_PREVIOUS_RESULT
    b.solve()
    oracle.solve_in_parallel(qs)
    check(b, qs, g2, "update rho, mu_in + previous result")
    b.close()
    b, qs = make(InitialGuess.EQUALITY_CONSTRAINED_INITIAL_GUESS)
    b.solve()
    oracle.solve_in_parallel(qs)
    check(b, qs, g, "equality constrained initial guess")
    b.close()
    # verbose: the per-iteration trace against the oracle's (GPDAL only: the path must be the same line by line)
    if merit == 0:
        b, qs = make(InitialGuess.NO_INITIAL_GUESS)
        for i, q in enumerate(qs):
            b.settings(i).verbose = 1
            q.settings.verbose = 1
        b.solve()
        for i, q in enumerate(qs):
            q.solve()
            t, to = b.trace(i), q.trace()
            assert t.shape == to.shape and t.shape[0] > 0, (i, t.shape, to.shape)
            assert np.array_equal(t[:, :2], to[:, :2]), i
            outer = t[:, 0] == 1
            assert np.array_equal(t[outer, 5:7], to[outer, 5:7]), i  # mu_in, rho
        check(b, qs, g, "verbose")
        b.close()
    return forked[0]


def case_diag_wave_backward(lib, oracle, randqp, dim=24, B=3):
    """QPLayer backward (reference dense/compute_ECJ.hpp:29-189) on diagonal-structure QPs (C = I form): the backward
    kernel reads the state the SOLVE kernel left in HBM -- solution, slot list, persistent active-set flags -- whichever
    solve kernel that was.  The seven loss jacobians after a solve by the one-wavefront kernel against the oracle's."""
    H, g, Cm, l, u = c5_models(randqp, B, dim, seed0=3)
    b = N.Batch(B, dim, 0, dim, hessian_type=int(HessianType.Diagonal), lib=lib)
    settings_all(b, eps_abs=EPS, eps_rel=0)
    b.init(-1, H, g, None, None, Cm, l, u)
    b.solve()
    rng = np.random.default_rng(9)
    ld = np.zeros((B, 2 * dim))
    ld[:, :dim] = rng.standard_normal((B, dim))
    ld[:, dim:] = 0.1 * rng.standard_normal((B, dim))
    b.backward(ld, 1e-5, 1e-7, 1e-7)
    got = b.backward_results(-1)
    for i in range(B):
        q = oracle.QP(dim, 0, dim, hessian_type=HessianType.Diagonal)
        q.settings.eps_abs, q.settings.eps_rel = EPS, 0
        q.init(H[i], g[i], None, None, Cm[i], l[i], u[i])
        q.solve()
        ref = q.compute_backward(ld[i], 1e-5, 1e-7, 1e-7)
        for k, v in ref.items():
            scale = 1 + (np.max(np.abs(v)) if v.size else 0.0)
            assert np.max(np.abs(got[k][i] - v), initial=0.0) <= 1e-6 * scale, (i, k)
    b.close()
    return got


def case_diag_wave_infeasible(lib, oracle):
    """the one-wavefront diagonal kernel on an unbounded problem -- a linear objective (zero Hessian) that decreases along a
    coordinate whose upper bound is infinite: the reference's certificate test never fires on it (both sides run into
    max_iter with a diverging iterate), and status, counters and the direction of the iterate must be the oracle's"""
    dim = 12
    rng = np.random.default_rng(11)
    H = np.zeros((dim, dim))
    g = rng.standard_normal(dim)
    lb, ub = -np.ones(dim), np.ones(dim)
    ub[5], g[5] = np.inf, -1.0  # x_5 -> +inf lowers the objective without bound
    b = N.Batch(1, dim, 0, 0, box_constraints=True, hessian_type=int(HessianType.Zero), lib=lib)
    settings_all(b, eps_abs=EPS, eps_rel=0, initial_guess=int(InitialGuess.NO_INITIAL_GUESS), max_iter=60)
    b.init(0, H, g, None, None, None, None, None, lb, ub)
    b.solve()
    x, y, z, se, si, info = b.results()
    q = oracle.QP(dim, 0, 0, box_constraints=True, hessian_type=HessianType.Zero)
    q.settings.eps_abs, q.settings.eps_rel, q.settings.initial_guess = EPS, 0, InitialGuess.NO_INITIAL_GUESS
    q.settings.max_iter = 60
    q.init(H, g, None, None, None, None, None, lb, ub)
    q.solve()
    assert info[0].status == q.results.info.status, (info[0].status, q.results.info.status)
    assert (info[0].iter, info[0].iter_ext) == (q.results.info.iter, q.results.info.iter_ext)
    assert _direction_close(x[0], q.results.x) and _direction_close(z[0], q.results.z)
    b.close()
    return int(info[0].status)


INFEASIBLE_QP = dict(  # reference test/src/dense_qp_eq.cpp:217-256 ("infeasible qp")
    H=2.0 * np.eye(2), g=np.array([-18.0, -12.0]), C=np.array([[1.0, 0.0], [0.0, 1.0], [-1.0, 0.0]]),
    u=np.array([10.0, 10.0, -20.0]), l=np.full(3, -np.inf))


def _direction_close(a, ref, tol=1e-6):
    """certificates are rays: compare after normalisation"""
    na, nr = float(np.max(np.abs(a))), float(np.max(np.abs(ref)))
    if nr == 0.0:
        return na == 0.0
    return na > 0 and float(np.max(np.abs(a / na - ref / nr))) <= tol


def case_infeasibility_statuses(lib, oracle):
    """PROXQP_PRIMAL_INFEASIBLE and PROXQP_DUAL_INFEASIBLE outcomes and their certificates
    (reference dense/utils.hpp:269-419, dense/solver.hpp:1028-1063, 1572-1580): the reference's
    known-answer instance plus unbounded QPs / LPs; status and certificate rays against the oracle."""
    P = INFEASIBLE_QP
    b = N.Batch(1, 2, 0, 3, lib=lib)
    b.init(0, P["H"], P["g"], None, None, P["C"], P["l"], P["u"])
    settings_all(b, eps_abs=1e-9, eps_rel=0)
    b.solve()
    x, y, z, se, si, info = b.results(0)
    q = oracle.QP(2, 0, 3)
    q.init(P["H"], P["g"], None, None, P["C"], P["l"], P["u"])
    q.settings.eps_abs, q.settings.eps_rel = 1e-9, 0
    q.solve()
    assert q.results.info.status == QPSolverOutput.PROXQP_PRIMAL_INFEASIBLE  # the reference's own check
    assert info.status == QPSolverOutput.PROXQP_PRIMAL_INFEASIBLE
    assert info.iter == q.results.info.iter and info.iter_ext == q.results.info.iter_ext
    # the certificate dz (solver.hpp:1572-1580 leaves (dx, dy, dz) in the results): C^T dz ~ 0 with
    # u^T dz+ - l^T dz- < 0 (utils.hpp:269-324)
    assert _direction_close(z, q.results.z)
    dz = z / np.max(np.abs(z))
    assert np.max(np.abs(P["C"].T @ dz)) <= 1e-3 and float(P["u"] @ np.maximum(dz, 0)) < 0
    b.close()
    # dual infeasible: a direction of zero curvature along which the cost decreases for ever
    cases = [
        (np.diag([1.0, 1.0, 0.0]), np.array([0.0, 0.0, -1.0]), np.array([[1.0, 0.0, 0.0]]),
         np.array([-np.inf]), np.array([1.0]), HessianType.Dense),
        (np.zeros((3, 3)), np.array([1.0, 0.0, -1.0]), np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]),
         np.zeros(2), np.ones(2), HessianType.Zero),
    ]
    for H, g, Cm, l, u, ht in cases:
        n, ni = H.shape[0], Cm.shape[0]
        b = N.Batch(1, n, 0, ni, hessian_type=int(ht), lib=lib)
        settings_all(b, eps_abs=1e-9, eps_rel=0)
        b.init(0, H, g, None, None, Cm, l, u)
        b.solve()
        x, y, z, se, si, info = b.results(0)
        q = oracle.QP(n, 0, ni, hessian_type=ht)
        q.settings.eps_abs, q.settings.eps_rel = 1e-9, 0
        q.init(H, g, None, None, Cm, l, u)
        q.solve()
        assert q.results.info.status == QPSolverOutput.PROXQP_DUAL_INFEASIBLE
        assert info.status == QPSolverOutput.PROXQP_DUAL_INFEASIBLE
        assert info.iter == q.results.info.iter
        assert _direction_close(x, q.results.x)
        dx = x / np.max(np.abs(x))
        assert np.max(np.abs(H @ dx)) <= 1e-6 and float(g @ dx) < 0 and np.max(np.abs(Cm @ dx)) <= 1e-6
        b.close()


def infeasible_family(randqp, seeds, dim=20):
    """reference test/src/dense_qp_wrapper.cpp:7153-7215: strongly convex QPs pushed out of
    feasibility (b += 10, u -= 100)."""
    ne = ni = dim // 4
    out = []
    for sd in seeds:
        randqp.set_seed(sd)
        m = randqp.dense_strongly_convex_qp(dim, ne, ni, 0.15, 1e-2)
        out.append((m.H, m.g, m.A, m.b + 10.0, m.C, m.l, m.u - 100.0))
    return out, dim, ne, ni


def case_closest_feasible(lib, oracle, randqp, seeds, max_oracle_iter_ext=None):
    """primal_infeasibility_solving (reference dense/solver.hpp:1581-1595, 1757-1767;
    test/src/dense_qp_wrapper.cpp:7153-7215): same instances, with and without closest-feasible
    solving.  Status, iteration counts and solutions against the oracle; for the closest-feasible
    runs also the reference test's own acceptance lines."""
    models, dim, ne, ni = infeasible_family(randqp, seeds)
    eps = 1e-5
    seen = set()
    for pis in (False, True):
        qs = []
        for mod in models:
            q = oracle.QP(dim, ne, ni)
            s = q.settings
            s.eps_abs, s.eps_rel, s.initial_guess = eps, 0, InitialGuess.NO_INITIAL_GUESS
            s.primal_infeasibility_solving, s.eps_primal_inf, s.eps_dual_inf = pis, 1e-4, 1e-4
            q.init(*mod)
            qs.append(q)
        oracle.solve_in_parallel(qs)
        keep = [i for i, q in enumerate(qs)
                if max_oracle_iter_ext is None or q.results.info.iter_ext <= max_oracle_iter_ext]
        B = len(keep)
        if B == 0:
            continue
        b = N.Batch(B, dim, ne, ni, lib=lib)
        settings_all(b, eps_abs=eps, eps_rel=0, initial_guess=int(InitialGuess.NO_INITIAL_GUESS),
                     primal_infeasibility_solving=int(pis), eps_primal_inf=1e-4, eps_dual_inf=1e-4)
        stack = [np.stack([models[i][k] for i in keep]) for k in range(7)]
        b.init(-1, *stack)
        b.solve()
        x, y, z, se, si, info = b.results()
        for j, i in enumerate(keep):
            r = qs[i].results
            H, g, A, bb, Cm, l, u = models[i]
            done = (QPSolverOutput.PROXQP_SOLVED, QPSolverOutput.PROXQP_SOLVED_CLOSEST_PRIMAL_FEASIBLE)
            seen.add(int(info[j].status))
            if pis and r.info.status in done and info[j].status in done and info[j].status != r.info.status:
                # With check_duality_gap off, the reference's exit test at the TOP of the outer loop
                # reports plain SOLVED (solver.hpp:1509-1511) while the one after the inner loop reports
                # SOLVED_CLOSEST_PRIMAL_FEASIBLE (:1655-1663); which of the two sees convergence first
                # hangs on residuals at the rounding floor.  Same point, either label: x is compared.
                assert close(x[j], r.x), (pis, i)
                continue
            if pis and r.info.status == QPSolverOutput.PROXQP_MAX_ITER_REACHED:
                # (seed 14, see tests/test_oracle_known_answers.py::test_seed14_...: a feasible, badly scaled instance
                # on which the reference's BCL rule falls into a 12-periodic cycle through cold restarts; the only
                # exit is the safe guard, info.iter > 1e4, and whether the run then converges depends on the value
                # mu has at that moment, i.e. on the PHASE of the cycle -- 11 of its 15 phases end SOLVED, 4 end
                # MAX_ITER_REACHED.  The phase hangs on inner-iteration counts at a stagnated iterate, which are
                # decided at the rounding level, so the device may leave the cycle where the oracle does not.  If it
                # does, its answer must pass the reference test's acceptance lines, checked below.)
                assert info[j].status in done + (QPSolverOutput.PROXQP_MAX_ITER_REACHED,), (pis, i, info[j].status)
                # ... and whichever way it ends, it must end by the MECHANISM (case_seed14_mechanism): the run left
                # the cycle through the safe guard, and the outcome is the one the value of mu at that exit implies
                assert info[j].iter > 10000, (i, info[j].iter)
                assert (info[j].status in done) == (info[j].mu_in < 5e-3), (i, info[j].status, info[j].mu_in)
            else:
                assert info[j].status == r.info.status, (pis, i, info[j].status, r.info.status)
            if not pis:
                assert info[j].iter_ext == r.info.iter_ext, (pis, i)
            if info[j].status == QPSolverOutput.PROXQP_SOLVED and r.info.status == QPSolverOutput.PROXQP_SOLVED and not (pis and r.info.iter_ext > 1000):
                assert close(x[j], r.x) and close(y[j], r.y) and close(z[j], r.z), (pis, i)
            elif info[j].status in done and r.info.status in done:
                # the closest-feasible point is unique; the multipliers of the violated constraints
                # are not -- they grow by residual / mu at every outer iteration (1e13 after the
                # 10^4 iterations these runs take) and the BCL acceptance test, fed with primal
                # residuals at the 1e-14 rounding floor, shifts a mu update by one iteration between
                # two correct implementations: x and the well-defined multipliers are compared
                assert close(x[j], r.x), (pis, i)
                if ni:
                    live = np.abs(r.z) <= 1e6
                    assert close(z[j][live], r.z[live]), (pis, i)
            elif info[j].status == QPSolverOutput.PROXQP_PRIMAL_INFEASIBLE:
                assert _direction_close(np.concatenate([y[j], z[j]]), np.concatenate([r.y, r.z])), (pis, i)
            if pis and info[j].status != QPSolverOutput.PROXQP_MAX_ITER_REACHED:
                # the reference test's acceptance lines (dense_qp_wrapper.cpp:7189-7212)
                scaled_eps = float(np.max(np.abs(A.T @ np.ones(ne) + Cm.T @ np.ones(ni)))) * eps
                Cx = Cm @ x[j]
                pri = np.max(np.abs(A.T @ (A @ x[j] - bb) + Cm.T @ (np.maximum(Cx - u, 0) + np.minimum(Cx - l, 0))))
                dua = np.max(np.abs(H @ x[j] + g + A.T @ y[j] + Cm.T @ z[j]))
                assert pri <= scaled_eps and dua <= eps, (i, pri, dua)
        b.close()
    return seen


def case_seed14_mechanism(lib, randqp, guards=range(40, 55), max_iter=2500):
    """Seed 14 of the reference's closest-feasible family (test/src/dense_qp_wrapper.cpp:7153-7215) ON THE DEVICE: the
    mechanism that tests/test_oracle_known_answers.py::test_seed14_is_a_fixed_point_of_the_reference_bcl_rule
    establishes for the oracle, asserted on the kernel's own Info records instead of accepting either outcome.
      * 11 outer iterations bring the run to the mu floors with y = z = 0 and the primal residual of the penalty
        minimiser there, 0.129091 (computed in numpy, no oracle) -- above the BCL threshold 0.125893: a bad step
        for ever, i.e. the cycle;
      * the only exit is the safe guard (info.iter > safe_guard: every step accepted, mu frozen where the cycle
        stood): sweeping the guard over one period of the cycle, every run that ends SOLVED passes the reference
        test's acceptance lines and has mu_in < 5e-3 at the exit, every other run ends MAX_ITER_REACHED with
        mu_in >= 5e-3 and fails them -- the outcome is a function of the PHASE -- and over a full period (15 guard
        values) 11 phases end SOLVED, as for the oracle."""
    (H, g, A, b, C, l, u), = infeasible_family(randqp, [14])[0]
    dim, ne, ni = 20, 5, 5
    mu_eq, mu_in = 1e-9, 1e-8
    act = np.ones(ni, bool)
    for _ in range(50):
        K = H + A.T @ A / mu_eq + C[act].T @ C[act] / mu_in
        xs = np.linalg.solve(K, -g + A.T @ b / mu_eq + C[act].T @ u[act] / mu_in)
        new = (C @ xs - u) > 0
        if (new == act).all():
            break
        act = new
    pri_fixed_point = max(np.max(np.abs(A @ xs - b)), np.max(np.maximum(C @ xs - u, 0)))
    assert abs(pri_fixed_point - 0.129091) < 1e-6

    def batch(B):
        bt = N.Batch(B, dim, ne, ni, lib=lib)
        settings_all(bt, eps_abs=1e-5, eps_rel=0, initial_guess=int(InitialGuess.NO_INITIAL_GUESS),
                     primal_infeasibility_solving=1, eps_primal_inf=1e-4, eps_dual_inf=1e-4)
        bt.init(-1, *[np.stack([a] * B) for a in (H, g, A, b, C, l, u)])
        return bt

    b1 = batch(1)
    b1.settings(0).max_iter = 11
    b1.solve()
    x, y, z, se, si, info = b1.results()
    assert abs(info[0].pri_res - pri_fixed_point) <= 1e-6 * pri_fixed_point, info[0].pri_res
    assert np.max(np.abs(y)) == 0 and np.max(np.abs(z)) == 0
    # (the 11th outer iteration is the first to see both residuals unchanged at the floors mu_in = 1e-8, mu_eq = 1e-9,
    # settings.hpp:222-223, so it ends with the cold restart of solver.hpp:1700-1712: mu back to 1 / 1.1)
    assert abs(info[0].mu_in - 1.0 / 1.1) <= 1e-15 and abs(info[0].mu_eq - 1.0 / 1.1) <= 1e-15, (info[0].mu_in, info[0].mu_eq)
    b1.close()
    guards = list(guards)
    bg = batch(len(guards))
    for k, gd in enumerate(guards):
        st = bg.settings(k)
        st.safe_guard, st.max_iter = gd, max_iter
    bg.solve()
    x, y, z, se, si, info = bg.results()
    done = (QPSolverOutput.PROXQP_SOLVED, QPSolverOutput.PROXQP_SOLVED_CLOSEST_PRIMAL_FEASIBLE)
    n_ok = 0
    for k, gd in enumerate(guards):
        scaled_eps = float(np.max(np.abs(A.T @ np.ones(ne) + C.T @ np.ones(ni)))) * 1e-5
        Cx = C @ x[k]
        pri = np.max(np.abs(A.T @ (A @ x[k] - b) + C.T @ (np.maximum(Cx - u, 0) + np.minimum(Cx - l, 0))))
        dua = np.max(np.abs(H @ x[k] + g + A.T @ y[k] + C.T @ z[k]))
        ok = bool(pri <= scaled_eps and dua <= 1e-5)
        assert info[k].iter > gd, (gd, info[k].iter)  # left through the guard
        assert (info[k].status in done and ok) or (info[k].status == QPSolverOutput.PROXQP_MAX_ITER_REACHED and not ok), \
            (gd, info[k].status, pri, dua)
        assert ok == (info[k].mu_in < 5e-3), (gd, info[k].mu_in)  # decided by mu at the exit = the phase of the cycle
        n_ok += ok
    bg.close()
    return n_ok, len(guards)


def case_primal_ldlt(lib, oracle, randqp, dim, B, seed0=1):
    """DenseBackend::PrimalLDLT at the shape of benchmark/timings-dense-backend.cpp:25-75: n_eq = n_in =
    2 dim with box constraints around a known feasible point (b = A x_sol, u = C x_sol + delta).  The
    device factorises the dim x dim primal matrix P_J; the oracle solves the same Newton systems through
    the reference's rank-updated KKT factorisation.  With 2 dim equalities on dim variables the
    multipliers are not unique: x, the KKT residuals and the status are compared."""
    from proxsuite_amd._ctypes_defs import DenseBackend
    ne = ni = 2 * dim
    H = np.zeros((B, dim, dim))
    g = np.zeros((B, dim))
    A = np.zeros((B, ne, dim))
    bb = np.zeros((B, ne))
    Cm = np.zeros((B, ni, dim))
    l = np.zeros((B, ni))
    u = np.zeros((B, ni))
    lb = np.zeros((B, dim))
    ub = np.zeros((B, dim))
    for s in range(B):
        randqp.set_seed(seed0 + s)
        m = randqp.dense_strongly_convex_qp(dim, ne, ni, 0.75, 1e-2)
        x_sol = np.array([randqp.normal_rand() for _ in range(dim)])
        delta = np.array([randqp.uniform_rand() for _ in range(ni)])
        shift = np.array([randqp.uniform_rand() for _ in range(dim)])
        H[s], g[s], A[s], Cm[s], l[s] = m.H, m.g, m.A, m.C, m.l
        u[s] = m.C @ x_sol + delta
        bb[s] = m.A @ x_sol
        ub[s], lb[s] = x_sol + shift, x_sol - shift
    b = N.Batch(B, dim, ne, ni, box_constraints=True, dense_backend=int(DenseBackend.PrimalLDLT), lib=lib)
    assert b.dense_backend == int(DenseBackend.PrimalLDLT)
    settings_all(b, eps_abs=EPS, eps_rel=0, initial_guess=int(InitialGuess.NO_INITIAL_GUESS))
    b.init(-1, H, g, A, bb, Cm, l, u, lb, ub)
    b.solve()
    x, y, z, se, si, info = b.results()
    qs = oracle_solve_many(oracle, [(H[i], g[i], A[i], bb[i], Cm[i], l[i], u[i], lb[i], ub[i]) for i in range(B)],
                           dim, ne, ni, box_constraints=True, dense_backend=DenseBackend.PrimalLDLT)
    solved = 0
    for i in range(B):
        # (2 dim equalities on dim variables: a few seeds trip the primal-infeasibility certificate at the
        # default eps_primal_inf although b = A x_sol is consistent -- in the oracle and on the device alike)
        assert info[i].status == qs[i].results.info.status, (i, info[i].status, qs[i].results.info.status)
        if info[i].status != QPSolverOutput.PROXQP_SOLVED:
            continue
        solved += 1
        pri, dua = kkt_numpy(H[i], g[i], A[i], bb[i], Cm[i], l[i], u[i], x[i], y[i], z[i], lb[i], ub[i])
        assert pri <= EPS and dua <= EPS, (i, pri, dua)
        assert close(x[i], qs[i].results.x), i
        bad = info_close(info[i], qs[i].results.info)
        assert bad is None, (i, bad)
    assert solved >= (3 * B) // 4, solved
    # automatic choice picks this engine at this shape (reference dense/wrapper.hpp:81-113)
    b2 = N.Batch(1, dim, ne, ni, box_constraints=True, lib=lib)
    assert b2.dense_backend == int(DenseBackend.PrimalLDLT)
    b2.close()
    b.close()


def case_refinement_fallback(lib, oracle, names=("QADLITTL", "QSHARE2B", "QPCBOEI2"), need_stats=True):
    """Row a14: the refinement fallback of iterative_solve_with_permut_fact (reference dense/solver.hpp:474-532:
    err >= max(eps, eps_refact) after the refinement loop -> refactorize() and solve + refine once more).
    It needs ill-conditioned data: the Maros-Meszaros fixtures QADLITTL (takes it once even with default
    settings), QSHARE2B and QPCBOEI2, run with nb_iterative_refinement = 1 (no refinement pass after the first
    solve) and eps_refact = 0 so that every linear solve that misses the inner tolerance takes it (oracle: 4, 11
    and 6 fallbacks).  On the device the fallback rebuilds an EDITED dual Schur factor through the step's own
    factorisation call site and repeats the solve (a factor without edits since its last full factorisation
    would be rebuilt to the same bits and is left alone); the oracle runs the reference's refactorize.  Both must
    meet the reference test's acceptance lines (test/src/dense_maros_meszaros.cpp:139-161), end with the same
    status and agree on the optimal value; the device's event counter shows that the path ran (instrumented builds only: the
    emulator and libproxqp_hip_stats.so)."""
    import os
    from conftest import split_maros
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "maros_meszaros_small.npz"))
    eps = 2e-8
    kw = dict(eps_abs=eps, eps_rel=0, eps_primal_inf=1e-12, eps_dual_inf=1e-12, nb_iterative_refinement=1, eps_refact=0.0)
    taken = 0
    for name in names:
        H, g, Aeq, bb, C, lin, uin = split_maros(*(d["%s/%s" % (name, k)] for k in "PqAlu"))
        n, n_eq, n_in = H.shape[0], Aeq.shape[0], C.shape[0]
        bt = N.Batch(1, n, n_eq, n_in, lib=lib)
        settings_all(bt, **kw)
        bt.init(0, H, g, Aeq, bb, C, lin, uin)
        bt.solve()
        x, y, z, se, si, info = bt.results(0)
        q = oracle.QP(n, n_eq, n_in)
        for k, v in kw.items():
            setattr(q.settings, k, v)
        q.init(H, g, Aeq, bb, C, lin, uin)
        q.solve()
        assert q.counters()["n_refactorize"] > 0, name
        assert info.status == q.results.info.status == QPSolverOutput.PROXQP_SOLVED, (name, info.status)
        dua = H @ x + g + (Aeq.T @ y if n_eq else 0) + (C.T @ z if n_in else 0)
        mag = np.abs(H) @ np.abs(x) + np.abs(g) + (np.abs(Aeq.T) @ np.abs(y) if n_eq else 0) + (np.abs(C.T) @ np.abs(z) if n_in else 0)
        assert np.max(np.abs(dua)) < 2 * eps + 4 * np.finfo(float).eps * np.max(mag), name
        if n_eq:
            assert np.max(np.abs(Aeq @ x - bb)) < eps * 1.0001, name
        if n_in:
            assert (C @ x - lin).min() > -eps and (C @ x - uin).max() < eps, name
        assert info.dua_res <= eps and info.pri_res <= eps, name
        # (these problems have flat directions in H: the minimiser is not unique, the optimal value is)
        assert abs(info.objValue - q.results.info.objValue) <= 1e-6 * (1 + abs(q.results.info.objValue)), (
            name, info.objValue, q.results.info.objValue)
        if need_stats:
            from proxsuite_amd._native import STAT_NAMES
            taken += int(bt.stats()[0, STAT_NAMES.index("n_refactorize")])
        bt.close()
    if need_stats:
        assert taken > 0, "the device never took the refinement fallback"


def case_schur_factor_identity(lib, randqp, n=100, ne=50, ni=100, B=64, tol=1e-11):
    """Rows a10-a13 checked directly on the factor the device leaves behind: after a cold solve the dual
    Schur block of most QPs has taken rank-1 row appends (reference insert_block_at, linalg/dense/modify.hpp:
    129-264) and deletions (delete_at, :80-127) since its last full factorisation.  The inverse factor W_S and
    D_S are read back and the identity  W_S (M_J + G_JJ) W_S^T = D_S  is evaluated in numpy (holes = identity
    rows): max |.| <= tol relative to max |D_S|.  Iterative refinement would hide a sloppy edit as a slowdown;
    this does not."""
    m = randqp.dense_strongly_convex_qp_batch(B, n, ne, ni, 0.15, 1e-2)
    b = N.Batch(B, n, ne, ni, lib=lib)
    b.init(-1, m.H, m.g, m.A, m.b, m.C, m.l, m.u)
    edited = worst = 0
    # a converged solve often ends right after a mu update (factor rebuilt, no edits): runs stopped after a few
    # outer iterations catch the factor in the middle of its life as well
    for max_iter in (10000, 2, 3, 4, 5, 6):
      settings_all(b, eps_abs=EPS, eps_rel=0, initial_guess=int(InitialGuess.NO_INITIAL_GUESS), max_iter=max_iter)
      b.solve()
      for q in range(B):
          WS, dS, G, slots, meta, mus = b.schur_factor(q)
          if not meta["ls_valid"]:
              continue
          r = ne + meta["n_slots"]
          cid = np.concatenate([np.arange(ne), ne + slots[:meta["n_slots"]]])
          live = np.concatenate([np.ones(ne, bool), slots[:meta["n_slots"]] >= 0])
          cidc = np.where(live, cid, 0)
          S = G[np.ix_(cidc, cidc)] + np.diag(np.concatenate([np.full(ne, mus[0]), np.full(meta["n_slots"], mus[1])]))
          S[~live, :] = 0
          S[:, ~live] = 0
          S[~live, ~live] = 1.0
          W = np.tril(WS[:r, :r])
          assert np.all(WS[:r, :r][np.triu_indices(r, 1)] == 0) and np.allclose(np.diag(W), 1.0)
          err = np.max(np.abs(W @ S @ W.T - np.diag(dS[:r]))) / max(1.0, np.max(np.abs(dS[:r])))
          worst = max(worst, err)
          edited += int(meta["ls_edited"])
          assert err <= tol, (q, err, meta)
          assert int(live[ne:].sum()) == meta["n_c"]
    assert edited >= B // 2, (edited, "too few QPs ended on an edited factor for this check to mean anything")
    b.close()
    return worst, edited


def case_random_sweep(lib, oracle, randqp, seed, count, verbose=True, n_range=(1, 120), shapes=None, only=None):
    """Randomised robustness sweep: shapes x {box constraints, Dense / Diagonal Hessian, DenseBackend Automatic /
    PrimalDualLDLT / PrimalLDLT} x {cold solve, then update(g) + WARM_START_WITH_PREVIOUS_RESULT re-solve, which
    restores the edited Schur factor -- holes included -- from HBM}, three QPs per shape.  Every QP must end with
    the oracle's status; SOLVED ones must have KKT <= 1e-9, the oracle's x to XYZ_TOL and the oracle's Info counters
    (iter, iter_ext, mu_updates, rho_updates EQUAL; objValue, mu, rho to 1e-9).  About a third of the instances are
    infeasible by construction (random bounds): they must be unsolved on both sides; on a few of those the two sides
    end with different non-SOLVED statuses (MAX_ITER_REACHED vs PRIMAL_INFEASIBLE) -- the reference's BCL rule cycles
    on them through cold restarts (same mechanism as seed 14 of the closest-feasible family,
    tests/test_oracle_known_answers.py) and where the cycle stands when the certificate test fires depends on the
    last bits: counted as `forks`, bounded by the caller.
    PDAL shapes (every third): see the two-tier gate below.
    Returns dict(failures, unsolved_alike, forks, solved, info_mismatch, pdal_same_path, pdal_forked)."""
    import time
    from proxsuite_amd._ctypes_defs import DenseBackend
    O, R = oracle, randqp
    rng = np.random.default_rng(int(seed))
    t0 = time.time()
    bad = notes = forks = solved = info_bad = pdal_tight = pdal_forked = 0
    for it in range(count):
        n = int(rng.integers(n_range[0], n_range[1]))
        ne = int(rng.integers(0, max(1, n // 2) + 1))
        ni = int(rng.integers(0, 2 * n + 2))
        box = bool(rng.integers(0, 3) == 0)
        hess = HessianType.Diagonal if rng.integers(0, 3) == 0 else HessianType.Dense
        backend = DenseBackend(int(rng.integers(0, 3)))
        if rng.integers(0, 6) == 0:  # the diagonal-structure path: no equality, box only or nothing dense
            ne, hess = 0, HessianType.Diagonal
            if rng.integers(0, 2):
                ni, box = 0, True
        if ne + ni == 0 and not box:
            ni = 1
        if shapes is not None:  # fixed shapes instead of the random draw (same data generation and checks)
            n, ne, ni, box, hess, backend = shapes[it]
            hess, backend = HessianType(hess), DenseBackend(backend)
        B = 3
        m = R.dense_strongly_convex_qp_batch(B, n, ne, ni, float(rng.uniform(0.1, 0.9)), 1e-2, seed0=int(rng.integers(0, 10000)))
        H = m.H if hess == HessianType.Dense else np.stack([np.diag(np.diag(h)) for h in m.H])
        lb = ub = None
        if box:
            xs = rng.standard_normal((B, n)); sh = rng.uniform(0.1, 1.0, (B, n))
            lb, ub = xs - sh, xs + sh
        g2 = m.g + 0.1 * rng.standard_normal(m.g.shape)
        if only is not None and it != only:  # (debugging: one shape of the stream)
            continue
        b = N.Batch(B, n, ne, ni, box_constraints=box, hessian_type=int(hess), dense_backend=int(backend), lib=lib)
        qs = []
        merit = 1 if it % 3 == 2 else 0  # every third shape with the PDAL merit function (settings.hpp: GPDAL is the default)
        for i in range(B):
            s = b.settings(i); s.eps_abs = 1e-9; s.eps_rel = 0; s.initial_guess = int(InitialGuess.NO_INITIAL_GUESS); s.max_iter = 2000
            s.merit_function_type = merit
            q = O.QP(n, ne, ni, box_constraints=box, hessian_type=hess, dense_backend=backend)
            q.settings.eps_abs = 1e-9; q.settings.eps_rel = 0; q.settings.initial_guess = InitialGuess.NO_INITIAL_GUESS; q.settings.max_iter = 2000
            q.settings.merit_function_type = merit
            qs.append(q)
        args = lambda i=None: [a if i is None else a[i] for a in (H, m.g)] + [
            (m.A if i is None else m.A[i]) if ne else None, (m.b if i is None else m.b[i]) if ne else None,
            (m.C if i is None else m.C[i]) if ni else None, (m.l if i is None else m.l[i]) if ni else None,
            (m.u if i is None else m.u[i]) if ni else None]
        bkw = lambda i=None: (dict(l_box=lb if i is None else lb[i], u_box=ub if i is None else ub[i]) if box else {})
        b.init(-1, *args(), **bkw())
        for i, q in enumerate(qs):
            q.init(*args(i), **bkw(i))
        for phase in (0, 1):
            gcur = m.g if phase == 0 else g2
            if phase == 1:
                for i in range(B):
                    b.settings(i).initial_guess = int(InitialGuess.WARM_START_WITH_PREVIOUS_RESULT)
                    qs[i].settings.initial_guess = InitialGuess.WARM_START_WITH_PREVIOUS_RESULT
                    qs[i].update(g=g2[i])
                b.update(-1, g=g2)
            b.solve()
            O.solve_in_parallel(qs)
            x, y, z, se, si, info = b.results()
            for i, q in enumerate(qs):
                r = q.results
                tag = (it, (n, ne, ni), "box" if box else "", hess.name, backend.name, "phase", phase, "qp", i)
                if info[i].status != r.info.status:
                    if info[i].status != 0 and r.info.status != 0:
                        # neither solves it (an infeasible instance): MAX_ITER_REACHED on one side and
                        # PRIMAL_INFEASIBLE on the other.  On such instances the iterates stagnate and the
                        # reference's cold-restart test (solver.hpp:1700-1712: new residual >= old residual)
                        # compares numbers equal to the last bit; the summation order decides the tie, the
                        # mu sequence forks, and the certificate fires -- or does not -- hundreds of outer
                        # iterations later (traced: scripts/README.md).  Counted, not failed.
                        forks += 1; continue
                    bad += 1
                    if verbose:
                        print("FAIL status", tag, info[i].status, r.info.status, flush=True)
                    continue
                if info[i].status != 0:
                    notes += 1; continue
                pri, dua = kkt_numpy(H[i], gcur[i], m.A[i], m.b[i], m.C[i], m.l[i], m.u[i], x[i], y[i], z[i],
                                        lb[i] if box else None, ub[i] if box else None)
                # (PrimalLDLT solves the normal equations, whose conditioning is the square of the KKT system's: two
                # orders of summation that walk the same path -- identical Info counters -- end 1e-9 apart on some
                # instances; measured 2.4e-9 at |x| = 2 with the PDAL merit function, seed 11 shape 38)
                xtol = 1e-8 if backend == DenseBackend.PrimalLDLT else None
                if merit == 1:
                    # PDAL: phi' has a JUMP at every breakpoint (the nu-term of linesearch.hpp:295-305 switches with the
                    # constraint), and the reference evaluates it exactly AT the breakpoints, where the activity test
                    # fl(r_i + alpha (C dx)_i) > 0 is decided by the last bit of its inputs.  Two summation orders take
                    # different steps from the first Newton iteration on (traced: same alpha, phi' = 5.60 vs 2.89) and
                    # meet again only at the solution.  Two tiers: a QP on which both sides took the SAME path (equal Info
                    # counters) gets the full gate of the GPDAL shapes -- x to XYZ_TOL, Info equal --, one whose path
                    # forked is compared by status and solution and counted (`pdal_forked`, bounded by the caller).
                    same_path = info_close(info[i], r.info, residuals=False) is None
                    if same_path:
                        pdal_tight += 1
                    else:
                        pdal_forked += 1
                        xtol = 1e-5
                if not (pri <= 1e-9 and dua <= 1e-9 and close(x[i], r.x, xtol)):
                    bad += 1; print("FAIL", tag, pri, dua, float(np.max(np.abs(x[i] - r.x))),
                                    "info", info_close(info[i], r.info, residuals=False), "iter", info[i].iter, r.info.iter,
                                    "mu_updates", info[i].mu_updates, r.info.mu_updates, "|x|", float(np.max(np.abs(r.x))), flush=True)
                else:
                    solved += 1
                    why = None if merit == 1 else info_close(info[i], r.info, residuals=False)
                    if why is not None:
                        info_bad += 1
                        if info_bad <= 12:
                            print("INFO", tag, why, flush=True)
        b.close()
    return dict(failures=bad, unsolved_alike=notes, forks=forks, solved=solved, info_mismatch=info_bad,
                pdal_same_path=pdal_tight, pdal_forked=pdal_forked, seconds=time.time() - t0)
