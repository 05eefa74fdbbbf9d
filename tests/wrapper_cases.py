"""The solver-flow test cases of the reference's dense wrapper suite (test/src/dense_qp_wrapper.cpp) and of
dense_qp_solve.cpp, dense_qp_eq.cpp, dense_unconstrained_qp.cpp, restated one by one: same problem generator and
seed, same sequence of init / solve / update / settings changes, same acceptance lines (primal and dual residual of the UNSCALED problem <= eps_abs, plus the exact-value checks on
rho / mu_eq / mu_eq_inv the reference makes).  Each case is a function `case(S)`; `S` is a `Side`: either the
oracle (tests pin the restatement against the reference's own known answers) or the device engine behind the
Python facade (emulator build on CPU, the real library on the GPU box).  A side records every checked solve
(x, y, z, iteration counts), so that a device run can also be compared with the oracle run step by step.

Test infrastructure.  The random helpers below continue the generator stream exactly where the reference's
tests do (utils::rand::vector_rand, sparse_matrix_rand_not_compressed, ...: random_qp_problems.hpp:150-364).
"""
import numpy as np

NO_GUESS, EQ_GUESS, WS_PREV, WARM, COLD_PREV = 0, 1, 2, 3, 4
PRIMAL_LDLT = 2
EPS = 1e-9


class M:
    """plain model record (reference dense::Model), numpy arrays"""

    def __init__(self, m):
        self.H, self.g, self.A, self.b, self.C, self.l, self.u = (np.array(getattr(m, k), dtype=np.float64)
                                                                   for k in ("H", "g", "A", "b", "C", "l", "u"))

    def args(self):
        return self.H, self.g, self.A, self.b, self.C, self.l, self.u


class NotForThisSource(Exception):
    """the case builds its own problem and does not exist in the other suite"""


class _Plain:
    pass


def mixed_qp(n, seed=1, reg=0.01):
    """The problem family of the reference's Python tests (test/src/dense_qp_wrapper.py:20-51 `generate_mixed_qp`),
    restated call for call so that numpy's global stream yields the same numbers: a sparse symmetric P shifted to
    positive definite, one random matrix split into equalities (first n/4 rows) and one-sided inequalities
    (from row n/4 on), right-hand sides from a fictitious solution, lower bounds at -1e20."""
    import scipy.sparse as spa
    np.random.seed(seed)
    n_eq = n_in = int(n / 4)
    m = n_eq + n_in
    P = spa.random(n, n, density=0.075, data_rvs=np.random.randn, format="csc").toarray()
    P = (P + P.T) / 2.0
    s = max(np.absolute(np.linalg.eigvals(P)))
    P = P + (abs(s) + reg) * np.eye(n)
    q = np.random.randn(n)
    A = spa.random(m, n, density=0.15, data_rvs=np.random.randn, format="csc").toarray(order="C")
    v = np.random.randn(n)
    np.random.rand(m)  # (the reference draws a vector here that it does not use)
    u = A @ v
    l = -1.0e20 * np.ones(m)
    r = _Plain()
    r.H, r.g, r.A, r.b, r.C, r.u, r.l = np.asarray(P), q, A[:n_eq, :], u[:n_eq], A[n_in:, :], u[n_in:], l[n_in:]
    return M(r)


class Side:
    def __init__(self, make_qp, kkt, randqp, name):
        self._make, self.kkt, self.R, self.name = make_qp, kkt, randqp, name
        self.backend = None  # a DenseBackend value forced on every QP object the case builds (None: the case's own)
        self.trace = []
        self.source = "cpp"  # "python": the problem of the reference's Python suite instead of the C++ generator's
        self.tol = 1e-8  # device against oracle on x, y, z (cases with a degenerate solution set widen it)
        self.unique_x = True

    def make_qp(self, *args, **kw):
        if self.backend is not None and len(args) < 6:
            kw.setdefault("dense_backend", self.backend)
        return self._make(*args, **kw)

    # -- generator stream (random_qp_problems.hpp:150-161, 308-364)
    def vector_rand(self, n):
        return np.array([self.R.normal_rand() for _ in range(n)])

    def sparse_matrix_rand_not_compressed(self, rows, cols, p):
        a = np.zeros((rows, cols))
        for i in range(rows):
            for j in range(cols):
                if self.R.uniform_rand() < p:
                    a[i, j] = self.R.normal_rand()
        return a

    def sparse_positive_definite_rand_not_compressed(self, n, rho, p):
        h = np.zeros((n, n))
        for i in range(n):
            for j in range(n):
                if self.R.uniform_rand() < p / 2:
                    h[i, j] = self.R.normal_rand()
        h = (h + h.T) * 0.5
        h[np.diag_indices(n)] += rho + abs(np.linalg.eigvalsh(h).min())
        return h

    def model(self, dim=10, n_eq=None, n_in=None, sparsity=0.15, seed=1):
        if self.source == "python":
            # the Python suite's problem (test/src/dense_qp_wrapper.py:20-51) for the cases that take the default
            # model of the C++ suite; the other cases are specific to the C++ suite
            if (dim, n_eq, n_in, sparsity, seed) != (10, None, None, 0.15, 1):
                raise NotForThisSource()
            self.R.set_seed(1)  # (the helpers that draw AFTER the model keep their own stream)
            return mixed_qp(10), 10, 2, 2
        if seed is not None:
            self.R.set_seed(seed)
        n_eq = dim // 4 if n_eq is None else n_eq
        n_in = dim // 4 if n_in is None else n_in
        return M(self.R.dense_strongly_convex_qp(dim, n_eq, n_in, sparsity, 1e-2)), dim, n_eq, n_in

    def qp(self, dim, n_eq, n_in, guess=None, eps=EPS, **kw):
        q = self.make_qp(dim, n_eq, n_in, **kw)
        q.settings.eps_abs = eps
        q.settings.eps_rel = 0
        if guess is not None:
            q.settings.initial_guess = guess
        return q

    def check(self, q, m, eps=EPS, l_box=None, u_box=None):
        """the reference's two acceptance lines on the unscaled problem"""
        r = q.results
        x, y, z = np.array(r.x), np.array(r.y), np.array(r.z)
        pri, dua = self.kkt(m.H, m.g, m.A, m.b, m.C, m.l, m.u, x, y, z, l_box, u_box)
        self.trace.append(dict(x=x, y=y, z=z, iter=int(r.info.iter), iter_ext=int(r.info.iter_ext),
                               status=int(r.info.status), rho=float(r.info.rho), mu_eq=float(r.info.mu_eq),
                               mu_in=float(r.info.mu_in), tol=self.tol, unique_x=self.unique_x))
        assert pri <= eps, "%s: primal residual %.3e > %.1e" % (self.name, pri, eps)
        assert dua <= eps, "%s: dual residual %.3e > %.1e" % (self.name, dua, eps)


def _init_solve_check(S, q, m, **kw):
    q.init(*m.args(), **kw)
    q.solve()
    S.check(q, m)


# ---------------------------------------------------------------------------------------------------------
# dense_qp_wrapper.cpp:16-162  empty equality constraints given three ways
def case_empty_equality(S):
    m, dim, n_eq, n_in = S.model(n_eq=0)
    q = S.qp(dim, 0, n_in)
    q.init(m.H, m.g, m.A, m.b, m.C, m.l, m.u)  # A of size (0, 10)
    q.solve()
    S.check(q, m)
    q2 = S.qp(dim, 0, n_in)
    q2.init(m.H, m.g, np.zeros((0, 0)), np.zeros(0), m.C, m.l, m.u)  # A of size (0, 0)
    q2.solve()
    S.check(q2, m)
    q3 = S.qp(dim, 0, n_in)
    q3.init(m.H, m.g, None, None, m.C, m.l, m.u)  # nullopt
    q3.solve()
    S.check(q3, m)


def _update_case(mutate):
    """dense_qp_wrapper.cpp:163-1108: solve, change part of the model, update, solve; a fresh QP object on the
    updated model must pass too"""

    def case(S):
        m, dim, n_eq, n_in = S.model()
        q = S.qp(dim, n_eq, n_in)
        _init_solve_check(S, q, m)
        upd = mutate(S, m, dim, n_eq, n_in)
        q.update(**upd)
        q.solve()
        S.check(q, m)
        q2 = S.qp(dim, n_eq, n_in)
        _init_solve_check(S, q2, m)

    return case


def _mut_H(S, m, dim, n_eq, n_in):  # :163-293
    m.H = np.eye(dim)
    return dict(H=m.H)


def _new_A(S, m, dim, n_eq):
    """C++ suite: a fresh random A (b kept); Python suite: A and b of the same family with seed 2
    (dense_qp_wrapper.py:3343, 3359)"""
    if S.source == "python":
        other = mixed_qp(dim, seed=2)
        return dict(A=other.A, b=other.b)
    return dict(A=S.sparse_matrix_rand_not_compressed(n_eq, dim, 0.15))


def _mut_A(S, m, dim, n_eq, n_in):  # :294-426
    upd = _new_A(S, m, dim, n_eq)
    for k, v in upd.items():
        setattr(m, k, v)
    return upd


def _mut_C(S, m, dim, n_eq, n_in):  # :427-559
    m.C = S.sparse_matrix_rand_not_compressed(n_in, dim, 0.15)
    return dict(C=m.C)


def _mut_b(S, m, dim, n_eq, n_in):  # :560-692
    x_sol = S.vector_rand(dim)
    m.b = m.A @ x_sol
    return dict(b=m.b)


def _mut_u(S, m, dim, n_eq, n_in):  # :693-828
    x_sol = S.vector_rand(dim)
    delta = np.array([S.R.uniform_rand() for _ in range(n_in)])
    m.u = m.C @ x_sol + delta
    return dict(u=m.u)


def _mut_g(S, m, dim, n_eq, n_in):  # :829-960
    m.g = S.vector_rand(dim)
    return dict(g=m.g)


def _mut_all(S, m, dim, n_eq, n_in):  # :961-1108  H and A and b and u and l
    m.H = S.sparse_positive_definite_rand_not_compressed(dim, 1e-2, 0.15)
    m.A = S.sparse_matrix_rand_not_compressed(n_eq, dim, 0.15)
    x_sol = S.vector_rand(dim)
    delta = np.array([S.R.uniform_rand() for _ in range(n_in)])
    m.b = m.A @ x_sol
    m.u = m.C @ x_sol + delta
    m.l = m.C @ x_sol - delta
    return dict(H=m.H, A=m.A, b=m.b, l=m.l, u=m.u)


case_update_H = _update_case(_mut_H)
case_update_A = _update_case(_mut_A)
case_update_C = _update_case(_mut_C)
case_update_b = _update_case(_mut_b)
case_update_u = _update_case(_mut_u)
case_update_g = _update_case(_mut_g)
case_update_H_A_b_u_l = _update_case(_mut_all)


# :1109-1238  update rho ("restart the problem with default options")
def case_update_rho(S):
    m, dim, n_eq, n_in = S.model()
    q = S.qp(dim, n_eq, n_in)
    _init_solve_check(S, q, m)
    q.update(update_preconditioner=True, rho=1e-7)
    q.solve()
    S.check(q, m)
    q2 = S.qp(dim, n_eq, n_in)
    _init_solve_check(S, q2, m, compute_preconditioner=True, rho=1e-7)


# :1239-1371  update mu_eq and mu_in
def case_update_mu(S):
    m, dim, n_eq, n_in = S.model()
    q = S.qp(dim, n_eq, n_in)
    _init_solve_check(S, q, m)
    q.update(update_preconditioner=True, mu_eq=1e-2, mu_in=1e-3)
    q.solve()
    S.check(q, m)
    q2 = S.qp(dim, n_eq, n_in)
    _init_solve_check(S, q2, m, compute_preconditioner=True, mu_eq=1e-2, mu_in=1e-3)


# :1372-1490  warm starting from random vectors
def case_warm_starting(S):
    m, dim, n_eq, n_in = S.model()
    q = S.qp(dim, n_eq, n_in)
    _init_solve_check(S, q, m)
    x_wm, y_wm, z_wm = S.vector_rand(dim), S.vector_rand(n_eq), S.vector_rand(n_in)
    q.settings.initial_guess = WARM
    q.solve(x_wm, y_wm, z_wm)
    S.check(q, m)
    q2 = S.qp(dim, n_eq, n_in, guess=WARM)
    q2.init(*m.args())
    q2.solve(x_wm, y_wm, z_wm)
    S.check(q2, m)


# :1491-1538  row-major inputs
def case_dense_init(S):
    m, dim, n_eq, n_in = S.model()
    q = S.qp(dim, n_eq, n_in)
    q.init(np.ascontiguousarray(m.H), m.g, np.ascontiguousarray(m.A), m.b, np.ascontiguousarray(m.C), m.l, m.u)
    q.solve()
    S.check(q, m)


def _two_objects(guess):  # :1539-1713  the same option on two objects
    def case(S):
        m, dim, n_eq, n_in = S.model()
        for _ in range(2):
            q = S.qp(dim, n_eq, n_in, guess=guess)
            _init_solve_check(S, q, m)

    return case


case_no_initial_guess = _two_objects(NO_GUESS)
case_equality_constrained_initial_guess = _two_objects(EQ_GUESS)


def _previous_result(guess, update_preconditioner):
    """:1714-1958  a second object warm started from the first one's solution pushed through the second's
    equilibration (ruiz.scale_primal_in_place / scale_dual_in_place_eq / _in: x / delta_x, c y / delta_eq,
    c z / delta_in), then the first object re-solved from its previous result"""

    def case(S):
        m, dim, n_eq, n_in = S.model()
        q = S.qp(dim, n_eq, n_in, guess=EQ_GUESS)
        _init_solve_check(S, q, m)
        q2 = S.qp(dim, n_eq, n_in, guess=WARM)
        q2.init(*m.args(), compute_preconditioner=True)
        delta, c = S.scaling(q2)
        r = q.results
        x = np.array(r.x) / delta[:dim]
        y = np.array(r.y) / delta[dim:dim + n_eq] * c
        z = np.array(r.z) / delta[dim + n_eq:] * c
        q2.solve(x, y, z)
        S.check(q2, m)
        q.settings.initial_guess = guess
        q.update(update_preconditioner=update_preconditioner)
        q.solve()
        S.check(q, m)

    return case


case_warm_start_with_previous_result = _previous_result(WS_PREV, False)
case_cold_start_option = _previous_result(COLD_PREV, True)


# :1959-2053  equilibration on / off at initialisation
def case_equilibration_at_init(S):
    m, dim, n_eq, n_in = S.model()
    for flag in (True, False):
        q = S.qp(dim, n_eq, n_in, guess=EQ_GUESS)
        _init_solve_check(S, q, m, compute_preconditioner=flag)


# :2054-2196  equilibration rederived / kept at update: "should get exact same results"
def case_equilibration_at_update(S):
    m, dim, n_eq, n_in = S.model()
    for flag in (True, False):
        q = S.qp(dim, n_eq, n_in, guess=EQ_GUESS)
        _init_solve_check(S, q, m, compute_preconditioner=True)
        first = np.array(q.results.x)
        q.update(update_preconditioner=flag)
        q.solve()
        S.check(q, m)
        assert np.max(np.abs(np.array(q.results.x) - first)) <= 1e-9


def _multi_solve(first, then=None, warm=False):
    """:2197-2959  four solves in a row; the option may change after the first"""

    def case(S):
        m, dim, n_eq, n_in = S.model()
        q = S.qp(dim, n_eq, n_in, guess=first)
        _init_solve_check(S, q, m)
        if then is not None:
            q.settings.initial_guess = then
        for _ in range(3):
            if warm:
                r = q.results
                q.solve(np.array(r.x), np.array(r.y), np.array(r.z))
            else:
                q.solve()
            S.check(q, m)

    return case


case_multi_no_guess = _multi_solve(NO_GUESS)                       # :2197-2320
case_multi_eq_guess = _multi_solve(EQ_GUESS)                       # :2321-2445
case_multi_eq_then_previous = _multi_solve(EQ_GUESS, WS_PREV)      # :2446-2574
case_multi_no_guess_then_previous = _multi_solve(NO_GUESS, WS_PREV)  # :2575-2703
case_multi_eq_then_cold = _multi_solve(EQ_GUESS, COLD_PREV)        # :2704-2833
case_multi_warm_start = _multi_solve(NO_GUESS, WARM, warm=True)    # :2834-2959


# :2960-3051  warm start of a second object from the first one's solution
def case_warm_start_from_init(S):
    m, dim, n_eq, n_in = S.model()
    q = S.qp(dim, n_eq, n_in, guess=NO_GUESS)
    _init_solve_check(S, q, m)
    q2 = S.qp(dim, n_eq, n_in)
    q2.init(*m.args())
    q2.settings.initial_guess = WARM
    r = q.results
    q2.solve(np.array(r.x), np.array(r.y), np.array(r.z))
    S.check(q2, m)


def _update_multi_solve(first, then=None):
    """:3052-3750  solve, H *= 2 and a new g with the preconditioner rederived, three more solves"""

    def case(S):
        m, dim, n_eq, n_in = S.model()
        q = S.qp(dim, n_eq, n_in, guess=first)
        _init_solve_check(S, q, m)
        if then is not None:
            q.settings.initial_guess = then
        m.H = m.H * 2.0
        m.g = S.vector_rand(dim)
        q.update(H=m.H, g=m.g, update_preconditioner=True)
        for _ in range(3):
            q.solve()
            S.check(q, m)

    return case


case_update_multi_no_guess = _update_multi_solve(NO_GUESS)                 # :3052-3188
case_update_multi_eq_guess = _update_multi_solve(EQ_GUESS)                 # :3189-3326
case_update_multi_eq_then_previous = _update_multi_solve(EQ_GUESS, WS_PREV)  # :3327-3469
case_update_multi_no_guess_then_previous = _update_multi_solve(NO_GUESS, WS_PREV)  # :3470-3609
case_update_multi_eq_then_cold = _update_multi_solve(EQ_GUESS, COLD_PREV)  # :3610-3750


# :3751-3927  update + warm start: a void update first ("the warm start should give the exact solution")
def case_update_multi_warm_start(S):
    m, dim, n_eq, n_in = S.model()
    q = S.qp(dim, n_eq, n_in, guess=NO_GUESS)
    _init_solve_check(S, q, m)
    q.settings.initial_guess = WARM
    r = q.results
    wm = (np.array(r.x), np.array(r.y), np.array(r.z))
    q.update(H=m.H, g=m.g, update_preconditioner=True)
    q.solve(*wm)
    S.check(q, m)
    r = q.results
    wm = (np.array(r.x), np.array(r.y), np.array(r.z))
    m.H = m.H * 2.0
    m.g = S.vector_rand(dim)
    q.update(H=m.H, g=m.g, update_preconditioner=True)
    q.solve(*wm)
    S.check(q, m)
    for _ in range(2):
        r = q.results
        q.solve(np.array(r.x), np.array(r.y), np.array(r.z))
        S.check(q, m)


_FIVE = (NO_GUESS, WS_PREV, EQ_GUESS, COLD_PREV, WARM)


def _solve_for(S, q, guess, donor):
    if guess == WARM:
        r = donor.results
        q.solve(np.array(r.x), np.array(r.y), np.array(r.z))
    else:
        q.solve()


# :3928-4128  rho given at init survives the solve, for the five options: CHECK(info.rho == 1e-7)
def case_init_with_rho(S):
    m, dim, n_eq, n_in = S.model()
    made = {}
    for guess in _FIVE:
        q = S.qp(dim, n_eq, n_in, guess=guess)
        q.init(*m.args(), compute_preconditioner=True, rho=1e-7)
        _solve_for(S, q, guess, made.get(EQ_GUESS))
        S.check(q, m)
        assert q.results.info.rho == 1e-7
        made[guess] = q


# :4129-4385  g updated, for the five options
def case_g_update_every_guess(S):
    m, dim, n_eq, n_in = S.model()
    old_g = m.g.copy()
    new_g = S.vector_rand(dim)
    made = {}
    for guess in _FIVE:
        m.g = old_g
        q = S.qp(dim, n_eq, n_in, guess=guess)
        q.init(*m.args())
        _solve_for(S, q, guess, made.get(EQ_GUESS))
        S.check(q, m)
        m.g = new_g
        q.update(g=m.g)
        q.solve()
        S.check(q, m)
        made[guess] = q


# :4386-4642  A updated, for the five options
def case_A_update_every_guess(S):
    m, dim, n_eq, n_in = S.model()
    old = dict(A=m.A.copy(), b=m.b.copy())
    new = _new_A(S, m, dim, n_eq)
    made = {}
    for guess in _FIVE:
        m.A, m.b = old["A"], old["b"]
        q = S.qp(dim, n_eq, n_in, guess=guess)
        q.init(*m.args())
        _solve_for(S, q, guess, made.get(EQ_GUESS))
        S.check(q, m)
        for k, v in new.items():
            setattr(m, k, v)
        q.update(**new)
        q.solve()
        S.check(q, m)
        made[guess] = q


# :4643-4937  rho updated, for the five options: CHECK(info.rho == 1e-7) after the re-solve
def case_rho_update_every_guess(S):
    m, dim, n_eq, n_in = S.model()
    made = {}
    for guess in _FIVE:
        q = S.qp(dim, n_eq, n_in, guess=guess)
        q.init(*m.args())
        _solve_for(S, q, guess, made.get(EQ_GUESS))
        S.check(q, m)
        q.update(update_preconditioner=True, rho=1e-7)
        q.solve()
        S.check(q, m)
        assert q.results.info.rho == 1e-7
        made[guess] = q


# :4938-5051  a slightly different g under WARM_START_WITH_PREVIOUS_RESULT
def case_g_update_previous_result(S):
    m, dim, n_eq, n_in = S.model()
    q = S.qp(dim, n_eq, n_in, guess=WS_PREV)
    _init_solve_check(S, q, m)
    m.g = m.g * 0.95
    q.update(g=m.g)
    q.solve()
    S.check(q, m)
    q2 = S.qp(dim, n_eq, n_in, guess=WS_PREV)
    _init_solve_check(S, q2, m)


def _near(a, b):
    return abs(a - b) <= 1e-9


def _defaults_after_updates(guess):
    """:5052-5815  rho / mu_eq given at init and at update become the defaults and are what info holds before
    and after every solve.  `guess`: WS_PREV is switched on after the update (the object starts with the
    default option); the other options are set at construction."""

    def case(S):
        if guess == WS_PREV and S.source == "python":
            # The C++ case switches to WARM_START_WITH_PREVIOUS_RESULT AFTER an update made under the default option.
            # That update cleans the workspace (n_c = 0, refactorize = false: workspace.hpp:372), so the solve takes
            # the first-solve branch without any setup_factorization (solver.hpp:1345-1375) and runs on the factor
            # the PREVIOUS solve left -- harmless on the C++ suite's problem, whose active set is empty at the
            # solution, a dimension mismatch (Eigen assertion / undefined behaviour in the reference, an assertion in
            # the oracle) on a problem with active inequalities.  The Python suite sets the option before init.
            raise NotForThisSource()
        m, dim, n_eq, n_in = S.model()
        rho, mu_eq = 1e-7, 1e-4
        at_start = None if guess == WS_PREV else guess

        def fresh():
            q = S.make_qp(dim, n_eq, n_in)
            if at_start is not None:
                q.settings.initial_guess = at_start
            assert int(q.settings.initial_guess) == (EQ_GUESS if at_start is None else at_start)
            q.settings.eps_abs = EPS
            q.settings.eps_rel = 0
            return q

        q = fresh()
        q.init(*m.args(), compute_preconditioner=True, rho=rho)
        assert _near(q.settings.default_rho, rho) and _near(q.results.info.rho, rho)
        q.solve()
        S.check(q, m)
        assert _near(q.settings.default_rho, rho) and _near(q.results.info.rho, rho)
        q.update(update_preconditioner=True, rho=1e-6)
        if guess == WS_PREV:
            q.settings.initial_guess = WS_PREV
        assert _near(q.settings.default_rho, 1e-6) and _near(q.results.info.rho, 1e-6)
        q.solve()
        S.check(q, m)
        assert _near(q.settings.default_rho, 1e-6) and _near(q.results.info.rho, 1e-6)

        q2 = fresh()
        q2.init(*m.args(), compute_preconditioner=True, mu_eq=mu_eq)
        for when in (0, 1):
            i = q2.results.info
            assert _near(q2.settings.default_mu_eq, mu_eq) and _near(i.mu_eq, mu_eq) and _near(i.mu_eq_inv, 1 / mu_eq)
            if when == 0:
                q2.solve()
                S.check(q2, m)

        q3 = fresh()
        q3.init(*m.args(), compute_preconditioner=True, rho=rho, mu_eq=mu_eq)
        for when in (0, 1):
            i = q3.results.info
            assert _near(q3.settings.default_rho, rho) and _near(i.rho, rho)
            assert _near(q3.settings.default_mu_eq, mu_eq) and _near(i.mu_eq, mu_eq) and _near(i.mu_eq_inv, 1 / mu_eq)
            if when == 0:
                q3.solve()
                S.check(q3, m)
        q3.update(update_preconditioner=True, rho=1e-6, mu_eq=1e-3)
        if guess == WS_PREV:
            q3.settings.initial_guess = WS_PREV
        i = q3.results.info
        assert _near(q3.settings.default_rho, 1e-6) and _near(i.rho, 1e-6)
        assert _near(q3.settings.default_mu_eq, 1e-3) and _near(i.mu_eq, 1e-3) and _near(i.mu_eq_inv, 1e3)
        q3.solve()
        S.check(q3, m)

    return case


case_defaults_after_updates_previous = _defaults_after_updates(WS_PREV)  # :5052-5242
case_defaults_after_updates_cold = _defaults_after_updates(COLD_PREV)    # :5243-5434
case_defaults_after_updates_eq = _defaults_after_updates(EQ_GUESS)       # :5435-5626
case_defaults_after_updates_no_guess = _defaults_after_updates(NO_GUESS)  # :5627-5815


def _defaults_after_several_solves(guess):
    """:5816-6732  the same with ten solves between the changes: the parameters given by the user are what
    info holds before and after each of them"""

    def case(S):
        m, dim, n_eq, n_in = S.model()
        rho, mu_eq = 1e-7, 1e-4
        at_start = None if guess == WS_PREV else guess

        def fresh():
            q = S.make_qp(dim, n_eq, n_in)
            if at_start is not None:
                q.settings.initial_guess = at_start
            q.settings.eps_abs = EPS
            q.settings.eps_rel = 0
            return q

        def holds(q, r=None, me=None):
            i = q.results.info
            if r is not None:
                assert _near(q.settings.default_rho, r) and _near(i.rho, r)
            if me is not None:
                assert _near(q.settings.default_mu_eq, me) and _near(i.mu_eq, me) and _near(i.mu_eq_inv, 1 / me)

        def ten(q, **kw):
            for _ in range(10):
                holds(q, **kw)
                q.solve()
                S.check(q, m)
                holds(q, **kw)

        q = fresh()
        q.init(*m.args(), compute_preconditioner=True, rho=rho)
        holds(q, r=rho)
        q.solve()
        S.check(q, m)
        holds(q, r=rho)
        if guess == WS_PREV:
            q.settings.initial_guess = WS_PREV
        ten(q, r=rho)
        q.update(update_preconditioner=True, rho=1e-6)
        ten(q, r=1e-6)

        q2 = fresh()
        q2.init(*m.args(), compute_preconditioner=True, mu_eq=mu_eq)
        holds(q2, me=mu_eq)
        q2.solve()
        S.check(q2, m)
        holds(q2, me=mu_eq)
        if guess == WS_PREV:
            q2.settings.initial_guess = WS_PREV
        ten(q2, me=mu_eq)

        q3 = fresh()
        q3.init(*m.args(), compute_preconditioner=True, rho=rho, mu_eq=mu_eq)
        ten(q3, r=rho, me=mu_eq)
        q3.update(update_preconditioner=True, rho=1e-6, mu_eq=1e-3)
        ten(q3, r=1e-6, me=1e-3)

    return case


case_defaults_after_solves_previous = _defaults_after_several_solves(WS_PREV)  # :5816-6051
case_defaults_after_solves_cold = _defaults_after_several_solves(COLD_PREV)    # :6052-6279
case_defaults_after_solves_eq = _defaults_after_several_solves(EQ_GUESS)       # :6280-6507
case_defaults_after_solves_no_guess = _defaults_after_several_solves(NO_GUESS)  # :6508-6732


# :6733-6802  update before init: update calls init internally
def case_update_before_init(S):
    m, dim, n_eq, n_in = S.model()
    q = S.qp(dim, n_eq, n_in, guess=NO_GUESS)
    q.update(*m.args(), update_preconditioner=True)
    q.solve()
    S.check(q, m)
    m.H = m.H * 2.0
    m.g = S.vector_rand(dim)
    q.update(H=m.H, g=m.g, update_preconditioner=True)
    q.solve()
    S.check(q, m)


# :7069-7152  box constraints widened by an update
def case_box_updates(S):
    m, dim, n_eq, n_in = S.model(dim=50, sparsity=1.0)
    q = S.qp(dim, n_eq, n_in, guess=NO_GUESS, box_constraints=True)
    u_box = np.full(dim, 1e2)
    l_box = np.full(dim, -1e2)
    q.init(*m.args(), l_box=l_box, u_box=u_box)
    q.solve()
    S.check(q, m, l_box=l_box, u_box=u_box)
    u_box = u_box + 1e1
    l_box = l_box - 1e1
    q.update(l=m.l, u=m.u, l_box=l_box, u_box=u_box)
    q.solve()
    S.check(q, m, l_box=l_box, u_box=u_box)


# :7618-7673  PrimalLDLT backend, only u given: CHECK(info.mu_updates > 0)
def case_primal_ldlt_mu_update(S):
    m, dim, n_eq, n_in = S.model(dim=3, n_eq=0, n_in=9, sparsity=1.0)
    q = S.qp(dim, 0, n_in, eps=1e-7, dense_backend=PRIMAL_LDLT)
    q.settings.compute_timings = True
    q.init(m.H, m.g, None, None, m.C, None, m.u)
    q.solve()
    assert q.results.info.mu_updates > 0
    S.check(q, m, eps=1e-7)


# ---------------------------------------------------------------------------------------------------------
# test/src/dense_qp_solve.cpp: the one-shot dense::solve function.  `S.one_shot` is the product's
# proxsuite_amd.proxqp.dense.solve on a device side and, on the oracle side, the reference's free function restated on
# top of the oracle's QP object (dense/wrapper.hpp:1000-1092: build a QP, copy the options, init, solve).
def oracle_one_shot(make_qp):
    def solve(H, g, A, b, C, l, u, x=None, y=None, z=None, eps_abs=None, eps_rel=None, rho=None, mu_eq=None,
              mu_in=None, verbose=None, compute_preconditioner=True, compute_timings=False, max_iter=None,
              initial_guess=EQ_GUESS):
        n = H.shape[0]
        n_eq = 0 if A is None else A.shape[0]
        n_in = 0 if C is None else C.shape[0]
        q = make_qp(n, n_eq, n_in, False, 1, 1)  # (dense Hessian, PrimalDualLDLT: wrapper.hpp:1043)
        q.settings.initial_guess = initial_guess
        for k, v in (("eps_abs", eps_abs), ("eps_rel", eps_rel), ("verbose", verbose), ("max_iter", max_iter)):
            if v is not None:
                setattr(q.settings, k, v)
        q.settings.compute_timings = compute_timings
        q.init(H, g, A, b, C, l, u, compute_preconditioner=compute_preconditioner, rho=rho, mu_eq=mu_eq, mu_in=mu_in)
        q.solve(x, y, z)
        return q

    return solve


def _results(r):
    return r if hasattr(r, "info") and hasattr(r, "x") and not hasattr(r, "results") else r.results


class _Holder:
    def __init__(self, results):
        self.results = results


def _one_shot_check(S, m, **kw):
    out = S.one_shot(*m.args(), **kw)
    holder = out if hasattr(out, "results") else _Holder(out)
    S.check(holder, m)
    return holder.results


# dense_qp_solve.cpp:16-83  fixed-size matrices: the solve function, then a QP object with equalities only
def case_solve_fixed_sizes(S):
    m, dim, n_eq, n_in = S.model(n_eq=5, n_in=2)
    _one_shot_check(S, m, eps_abs=EPS, eps_rel=0)
    q = S.make_qp(dim, n_eq, 0)
    q.init(m.H, m.g, m.A, m.b, None, None, None)
    q.settings.eps_abs = EPS
    q.solve()
    eq_only = M(m)
    eq_only.C, eq_only.l, eq_only.u = np.zeros((0, dim)), np.zeros(0), np.zeros(0)
    S.check(q, eq_only)


def case_solve_function(S):  # :85-132
    m, dim, n_eq, n_in = S.model()
    _one_shot_check(S, m, eps_abs=EPS, eps_rel=0)


def case_solve_with_rho(S):  # :134-182  CHECK(results.info.rho == 1e-7)
    m, dim, n_eq, n_in = S.model()
    r = _one_shot_check(S, m, eps_abs=EPS, eps_rel=0, rho=1e-7)
    assert r.info.rho == 1e-7


def case_solve_with_mu(S):  # :184-235
    m, dim, n_eq, n_in = S.model()
    _one_shot_check(S, m, eps_abs=EPS, eps_rel=0, mu_eq=1e-2, mu_in=1e-2)


def case_solve_warm_start(S):  # :237-276
    m, dim, n_eq, n_in = S.model()
    x_wm, y_wm, z_wm = S.vector_rand(dim), S.vector_rand(n_eq), S.vector_rand(n_in)
    _one_shot_check(S, m, x=x_wm, y=y_wm, z=z_wm, eps_abs=EPS, eps_rel=0)


def case_solve_verbose(S):  # :278-329
    m, dim, n_eq, n_in = S.model()
    _one_shot_check(S, m, eps_abs=EPS, eps_rel=0, verbose=True)


def case_solve_no_initial_guess(S):  # :331-385
    m, dim, n_eq, n_in = S.model()
    _one_shot_check(S, m, eps_abs=EPS, eps_rel=0, compute_preconditioner=True, compute_timings=True,
                    initial_guess=NO_GUESS)


# ---------------------------------------------------------------------------------------------------------
# test/src/dense_qp_eq.cpp:14-54  warm start AT the solution of an equality-constrained QP
def case_start_from_solution(S):
    dim, n_eq, n_in = 30, 6, 0
    S.R.set_seed(1)
    H = S.sparse_positive_definite_rand_not_compressed(dim, 1e-2, 0.15)
    A = S.sparse_matrix_rand_not_compressed(n_eq, dim, 0.15)
    sol = S.vector_rand(dim + n_eq)
    xs, ys = sol[:dim], sol[dim:]
    b = A @ xs
    g = -H @ xs - A.T @ ys

    class _Raw:
        pass

    raw = _Raw()
    raw.H, raw.g, raw.A, raw.b, raw.C, raw.l, raw.u = H, g, A, b, np.zeros((0, dim)), np.zeros(0), np.zeros(0)
    m = M(raw)
    q = S.make_qp(dim, n_eq, n_in)
    q.settings.eps_abs = EPS
    q.settings.initial_guess = WARM
    q.init(m.H, m.g, m.A, m.b, m.C, m.l, m.u)
    q.solve(xs, ys, np.zeros(0))
    S.check(q, m)


def _lp_with_equalities(hessian_type):
    """dense_qp_eq.cpp:103-156 (dense Hessian object) and :158-215 (the dedicated LP interface, HessianType::Zero):
    H = 0, g = -A^T y_sol keeps the LP bounded on the feasible set; the first dimensions of the reference's loop"""

    def case(S):
        S.R.set_seed(1)
        # g = -A^T y_sol makes the cost constant on {A x = b}: every feasible point is optimal, x is not unique and
        # is not compared between two arithmetic orders (y is: A has full row rank)
        S.unique_x = False
        S.tol = 1e-6
        for dim in (10, 110):
            m, dim, n_eq, n_in = S.model(dim=dim, n_eq=dim // 2, n_in=0, seed=None)
            m.H = np.zeros((dim, dim))
            y_sol = S.vector_rand(n_eq)
            m.g = -m.A.T @ y_sol
            q = S.qp(dim, n_eq, n_in, **({} if hessian_type is None else dict(hessian_type=hessian_type)))
            _init_solve_check(S, q, m)

    return case


case_lp_with_equalities = _lp_with_equalities(None)
case_lp_with_equalities_zero_hessian = _lp_with_equalities(0)


# ---------------------------------------------------------------------------------------------------------
# test/src/dense_unconstrained_qp.cpp:63-114  not strongly convex: g in the image of H
def case_unconstrained_not_strongly_convex(S):
    S.R.set_seed(1)
    for dim in (10, 110):
        m = M(S.R.dense_unconstrained_qp(dim, 0.15, 0.0))
        x_sol = S.vector_rand(dim)
        m.g = -m.H @ x_sol
        q = S.make_qp(dim, 0, 0)
        q.settings.eps_abs = EPS
        _init_solve_check(S, q, m)


def _unconstrained_identity(zero_g):  # :116-161 (g random as generated) and :163-208 (g = 0)
    def case(S):
        S.R.set_seed(1)
        dim = 100
        m = M(S.R.dense_unconstrained_qp(dim, 0.15, 1e-2))
        m.H = np.eye(dim)
        if zero_g:
            m.g = np.zeros(dim)
        q = S.make_qp(dim, 0, 0)
        q.settings.eps_abs = EPS
        _init_solve_check(S, q, m)

    return case


case_unconstrained_identity = _unconstrained_identity(False)
case_unconstrained_identity_zero_g = _unconstrained_identity(True)


# ---------------------------------------------------------------------------------------------------------
# test/src/dense_qp_wrapper.py cases without a C++ twin
def case_py_deterministic_behavior(S):  # :97-141  20 fresh objects give the same x, y, z to 1e-14
    m = mixed_qp(100)
    n, n_eq, n_in = 100, 25, 25
    prev = None
    for _ in range(21):
        q = S.make_qp(n, n_eq, n_in)
        q.settings.eps_abs = EPS
        q.init(*m.args())
        q.solve()
        r = q.results
        cur = (np.array(r.x), np.array(r.y), np.array(r.z))
        if prev is None:
            S.check(q, m)
        else:
            for a, b in zip(prev, cur):
                assert np.max(np.abs(a - b)) <= 1e-14
        prev = cur


def case_py_exact_solution_known(S):  # :3919-3958  x* = (2, ..., 2, 3) at the solver's default precision
    n = 150
    Mx = np.eye(n)
    for i in range(1, n - 1):
        Mx[i, i + 1] = -1
        Mx[i, i - 1] = 1
    raw = _Plain()
    raw.H, raw.g = Mx @ Mx.T, -np.ones(n)
    raw.A, raw.b = np.zeros((0, n)), np.zeros(0)
    raw.C, raw.l, raw.u = np.eye(n), 2.0 * np.ones(n), np.full(n, np.inf)
    m = M(raw)
    q = S.make_qp(n, 0, n)
    q.init(m.H, m.g, None, None, m.C, m.l, m.u)
    q.solve()
    S.tol = 1e-6  # (default eps_abs 1e-5: two arithmetic orders stop within that of each other)
    S.check(q, m, eps=1e-3)
    assert np.max(np.abs(np.array(q.results.x) - np.array([2.0] * 149 + [3.0]))) <= 1e-3


def case_py_initializing_with_None(S):  # :4540-4566
    raw = _Plain()
    raw.H = np.array([[65.0, -22.0, -16.0], [-22.0, 14.0, 7.0], [-16.0, 7.0, 5.0]])
    raw.g = np.array([-13.0, 15.0, 7.0])
    raw.A, raw.b, raw.C, raw.l, raw.u = np.zeros((0, 3)), np.zeros(0), np.zeros((0, 3)), np.zeros(0), np.zeros(0)
    m = M(raw)
    q = S.make_qp(3, 0, 0)
    q.init(m.H, m.g, None, None, None, None, None)
    q.solve()
    S.tol = 1e-6
    S.check(q, m, eps=1e-3)


# ---------------------------------------------------------------------------------------------------------
# dense_qp_wrapper.cpp:7212-7568: the estimate of the minimal eigenvalue of H handed to init() must come back in
# results.info.minimal_H_eigenvalue_estimate -- three cases (Eigen's exact solver, a manual value, power iteration),
# each on a 2 x 2 diagonal H with one negative entry, on 20 random diagonal H (dim 50) and on 20 dense H whose
# diagonal is shifted by 100 * normal draws (dim 50; generator stream as in the reference: the model first, then
# vector_rand for the diagonal).  The estimator is the product's host helper
# (proxsuite_amd.proxqp.dense.estimate_minimal_eigen_value_of_symmetric_matrix: the reference's
# dense/helpers.hpp:24-166 restated); the value the reference compares with is numpy's here (Eigen's there).
def _eig_estimator():
    from proxsuite_amd.proxqp import dense as _d
    return _d.estimate_minimal_eigen_value_of_symmetric_matrix, _d.EigenValueEstimateMethodOption


def _min_eig_case(method, tol, trivial_last):
    """method: "exact" | "manual" | "power"; trivial_last: the negative diagonal entry of the 2 x 2 problem"""

    def estimate(H, truth):
        est, Opt = _eig_estimator()
        if method == "exact":
            return est(H, Opt.ExactMethod, 1.0e-6, 10000)
        if method == "power":
            return est(H, Opt.PowerIteration, 1.0e-6, 10000)
        return truth

    def run(S, H, m, dim, truth):
        q = S.make_qp(dim, dim, dim)
        q.settings.max_iter = 1
        q.settings.max_iter_in = 1
        q.settings.initial_guess = NO_GUESS
        e = estimate(H, truth)
        q.init(H, m.g, m.A, m.b, m.C, m.l, m.u, compute_preconditioner=True, manual_minimal_H_eigenvalue=e)
        got = float(q.results.info.minimal_H_eigenvalue_estimate)
        S.trace.append(dict(value=got, handed=float(e)))
        assert abs(got - truth) <= tol, "%s: minimal_H_eigenvalue_estimate %.9g, expected %.9g" % (S.name, got, truth)

    def case(S):
        if S.source == "python":
            raise NotForThisSource()
        # trivial 2 x 2 problem (seed 0)
        S.R.set_seed(0)
        m = M(S.R.dense_strongly_convex_qp(2, 2, 2, 1.0, 1e-2))
        H = np.diag([1.0, trivial_last])
        run(S, H, m, 2, trivial_last)
        for i in range(20):  # random diagonal H, dim 50
            S.R.set_seed(i)
            m = M(S.R.dense_strongly_convex_qp(50, 50, 50, 1.0, 1e-2))
            d = S.vector_rand(50)
            run(S, np.diag(d), m, 50, float(d.min()))
        for i in range(20):  # dense H with a shifted diagonal, dim 50
            S.R.set_seed(i)
            m = M(S.R.dense_strongly_convex_qp(50, 50, 50, 1.0, 1e-2))
            H = m.H.copy()
            H[np.diag_indices(50)] += 100.0 * S.vector_rand(50)
            run(S, H, m, 50, float(np.linalg.eigvalsh(H).min()))

    return case


case_min_eigenvalue_exact = _min_eig_case("exact", 1e-6, -1.0)    # :7212-7330
case_min_eigenvalue_manual = _min_eig_case("manual", 1e-6, -1.0)  # :7331-7438
case_min_eigenvalue_power = _min_eig_case("power", 1e-3, -0.5)    # :7439-7568


CASES = {k[5:]: v for k, v in sorted(globals().items()) if k.startswith("case_")}
# cases that build their own problem without Side.model(): nothing changes for them on the Python suite's family
OWN_PROBLEM = {"py_deterministic_behavior", "py_exact_solution_known", "py_initializing_with_None",
               "start_from_solution", "lp_with_equalities", "lp_with_equalities_zero_hessian",
               "unconstrained_not_strongly_convex", "unconstrained_identity", "unconstrained_identity_zero_g",
               "min_eigenvalue_exact", "min_eigenvalue_manual", "min_eigenvalue_power"}
assert OWN_PROBLEM <= set(CASES)


def compare_traces(dev, ref):
    """device run against the oracle run of the same case, checked solve by checked solve"""
    assert len(dev) == len(ref)
    for i, (a, b) in enumerate(zip(dev, ref)):
        if "value" in a:  # a recorded scalar (minimal_H_eigenvalue_estimate after init): the same number on both sides
            assert a["value"] == b["value"] and a["handed"] == b["handed"], "entry %d: %r != %r" % (i, a, b)
            continue
        tol = a["tol"]
        assert a["status"] == b["status"], "solve %d: status %d != %d" % (i, a["status"], b["status"])
        for k in ("x", "y", "z") if a["unique_x"] else ("y", "z"):
            d = float(np.max(np.abs(a[k] - b[k]))) if a[k].size else 0.0
            assert d <= tol * (1.0 + float(np.max(np.abs(b[k]))) if b[k].size else 1.0), \
                "solve %d: %s differs by %.3e" % (i, k, d)
        for k in ("iter", "iter_ext"):
            assert a[k] == b[k], "solve %d: %s %d != %d" % (i, k, a[k], b[k])
        for k in ("rho", "mu_eq", "mu_in"):
            assert abs(a[k] - b[k]) <= 1e-12 * max(1.0, abs(b[k])), "solve %d: %s %r != %r" % (i, k, a[k], b[k])
