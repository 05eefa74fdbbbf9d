"""Cases for the Python surface (proxsuite_amd.proxqp.dense, proxsuite_amd.torch) shared by the
emulator suite (CPU, tests/test_emu_api.py) and the GPU suite (tests/test_gpu_api.py).  They read
like the reference's python tests (test/src/dense_qp_wrapper.py, parallel_qp_solve.py,
qplayer tests) and check results against the oracle."""
import numpy as np

from proxsuite_amd.utils import random_qp as rq


def _qp_data(randqp, n, ne, ni, seed):
    randqp.set_seed(seed)
    m = randqp.dense_strongly_convex_qp(n, ne, ni, 0.15, 1e-2)
    return dict(H=m.H, g=m.g, A=m.A, b=m.b, C=m.C, l=m.l, u=m.u)


def _kkt(oracle_mod, d, x, y, z):
    return oracle_mod.kkt_residuals(d["H"], d["g"], d["A"], d["b"], d["C"], d["l"], d["u"], x, y, z)


def case_qp_object(dense, oracle, randqp):
    """init / solve / update / warm re-solve on a standalone QP (reference
    test/src/dense_qp_wrapper.py: test_case_update_rho .. test_case_warm_start_with_previous_result)."""
    n, ne, ni = 10, 3, 4
    d = _qp_data(randqp, n, ne, ni, 1)
    qp = dense.QP(n, ne, ni)
    assert not qp.is_box_constrained()
    assert qp.which_hessian_type() == dense.HessianType.Dense
    assert qp.which_dense_backend() in (dense.DenseBackend.PrimalDualLDLT, dense.DenseBackend.PrimalLDLT)
    qp.settings.eps_abs = 1e-9
    qp.settings.eps_rel = 0
    assert qp.settings.initial_guess == dense.InitialGuess.EQUALITY_CONSTRAINED_INITIAL_GUESS
    assert qp.settings.verbose is False
    assert qp.results.info.status == dense.QPSolverOutput.PROXQP_NOT_RUN  # results.hpp:101
    qp.init(d["H"], d["g"], d["A"], d["b"], d["C"], d["l"], d["u"])
    # setup() -> cleanup_statistics() presets MAX_ITER_REACHED (results.hpp:172)
    assert qp.results.info.status == dense.QPSolverOutput.PROXQP_MAX_ITER_REACHED
    qp.solve()
    r = qp.results
    assert r.info.status == dense.QPSolverOutput.PROXQP_SOLVED
    pri, dua = _kkt(oracle, d, r.x, r.y, r.z)
    assert max(pri, dua) <= 1e-9
    assert r.x.shape == (n,) and r.y.shape == (ne,) and r.z.shape == (ni,)
    np.testing.assert_allclose(qp.model.H, d["H"])
    # oracle agreement incl. iteration counts
    o = oracle.QP(n, ne, ni)
    o.settings.eps_abs = 1e-9
    o.settings.eps_rel = 0
    o.init(d["H"], d["g"], d["A"], d["b"], d["C"], d["l"], d["u"])
    o.solve()
    np.testing.assert_allclose(r.x, o.results.x, atol=1e-9)
    assert r.info.iter == o.results.info.iter and r.info.iter_ext == o.results.info.iter_ext
    # update g and re-solve with the previous result (default guess after a first solve)
    g2 = d["g"] * 1.5
    qp.settings.initial_guess = dense.InitialGuess.WARM_START_WITH_PREVIOUS_RESULT
    qp.update(g=g2)
    qp.solve()
    d2 = dict(d, g=g2)
    r2 = qp.results
    assert max(_kkt(oracle, d2, r2.x, r2.y, r2.z)) <= 1e-9
    np.testing.assert_allclose(qp.model.g, g2)
    # warm start from the solution: zero iterations (reference test/src/cvxpy.cpp:104-160)
    qp.solve(r2.x, r2.y, r2.z)
    assert qp.settings.initial_guess == dense.InitialGuess.WARM_START
    assert qp.results.info.iter <= 1
    # positional overload with the preconditioner flag and rho (reference wrapper.hpp:354)
    qp.init(d["H"], d["g"], d["A"], d["b"], d["C"], d["l"], d["u"], True, 1e-7, 1e-4, 1e-2)
    assert qp.results.info.rho == 1e-7 and qp.results.info.mu_eq == 1e-4 and qp.results.info.mu_in == 1e-2
    qp.cleanup()
    assert np.all(qp.results.x == 0)


def case_errors(dense):
    """argument checks of the reference (wrapper.hpp:380-451, :542-546; model.hpp:65-68)."""
    import pytest
    with pytest.raises(ValueError):
        dense.QP(0, 0, 0)
    qp = dense.QP(4, 1, 2)
    with pytest.raises(ValueError):
        qp.init(np.eye(3), np.zeros(4), None, None, None, None, None)
    with pytest.raises(ValueError):
        qp.init(np.eye(4), np.zeros(4), np.zeros((1, 4)), np.zeros(1), np.zeros((2, 4)), np.zeros(2), np.zeros(2),
                l_box=np.zeros(4), u_box=np.ones(4))
    with pytest.raises(AttributeError):
        qp.settings.no_such_field = 1
    # reference test/src/dense_qp_wrapper.cpp:7569-7591 "check that model.is_valid function for symmetric matrices works
    # for epsilon precision": H symmetric up to one ulp in one entry initialises and is a valid model (model.hpp:121-132)
    rng = np.random.default_rng(5)
    M = rng.uniform(-1, 1, (3, 3))
    S = M + M.T
    S[0, 1] = S[1, 0] + np.finfo(np.float64).eps
    assert not np.array_equal(S, S.T)
    q3 = dense.QP(3, 0, 0)
    q3.init(S, None, None, None, None, None, None)
    assert q3.model.is_valid(False) is True
    q3.model.H = S + np.triu(np.ones((3, 3)), 1)
    with pytest.raises(ValueError, match="H is not symmetric"):
        q3.model.is_valid(False)
    q3.model.H = S
    q3.model.g = np.zeros(4)
    with pytest.raises(ValueError, match="g has not the expected size"):
        q3.model.is_valid(False)
    q5 = dense.QP(3, 0, 2)
    with pytest.raises(ValueError, match="C is zero, while n_in != 0"):  # (model.hpp:144-145; a fresh model holds zeros)
        q5.model.is_valid(False)


def case_box(dense, oracle, randqp):
    n, ni = 8, 3
    rng = np.random.default_rng(0)
    M = rng.standard_normal((n, n))
    H = M @ M.T + np.eye(n)
    g = rng.standard_normal(n)
    C = rng.standard_normal((ni, n))
    xs = rng.standard_normal(n)
    l, u = C @ xs - 1.0, C @ xs + 1.0
    lb, ub = xs - 0.3, xs + 0.3
    qp = dense.QP(n, 0, ni, True)
    assert qp.is_box_constrained()
    qp.settings.eps_abs = 1e-9
    qp.init(H, g, None, None, C, l, u, lb, ub)
    qp.solve()
    r = qp.results
    assert r.info.status == dense.QPSolverOutput.PROXQP_SOLVED
    assert r.z.shape == (ni + n,)
    # reference test dense_qp_wrapper.cpp:6889-6900: z = [z_in; z_box]; check KKT with the stacked C
    pri, dua = oracle.kkt_residuals(H, g, np.zeros((0, n)), np.zeros(0), C, l, u, r.x, r.y, r.z, lb, ub)
    assert max(pri, dua) <= 1e-9
    # one-shot solve with boxes gives the same answer
    r2 = dense.solve(H, g, None, None, C, l, u, eps_abs=1e-9, l_box=lb, u_box=ub)
    # the reference binding's positional box overload: (..., u, l_box, u_box, x, y, z, eps_abs)
    r3 = dense.solve(H, g, None, None, C, l, u, lb, ub, None, None, None, 1e-9)
    assert np.array_equal(r3.x, r2.x) and np.array_equal(r3.z, r2.z)
    np.testing.assert_allclose(r2.x, r.x, atol=1e-8)
    # l_box = None given positionally: still the box overload (the 9th argument cannot be y: n_eq != dim)
    r4 = dense.solve(H, g, None, None, C, l, u, None, ub, None, None, None, 1e-9)
    r5 = dense.solve(H, g, None, None, C, l, u, eps_abs=1e-9, u_box=ub)
    assert np.array_equal(r4.x, r5.x) and r4.z.shape == (ni + n,)
    # a plain positional call with a warm start x and y = None stays the plain overload
    r6 = dense.solve(H, g, None, None, C, l, u, r.x, None, None, 1e-9)
    assert r6.z.shape == (ni,)
    # n_eq == dim: (x, y) and (l_box, u_box) have the same shapes.  The 11th argument decides: an array there is `y` of
    # the box overload, a scalar `eps_abs` of the plain one; a call that leaves it open is the plain overload (the
    # reference-style warm start solve(..., u, x, y)); keywords always work
    A = np.eye(n)
    bvec = np.zeros(n)
    x0, y0 = np.zeros(n), np.zeros(n)
    r7 = dense.solve(H, g, A, bvec, C, l, u, x=x0, y=y0, eps_abs=1e-9)
    assert r7.z.shape == (ni,)
    r8 = dense.solve(H, g, A, bvec, C, l, u, x0, y0, None, 1e-9)  # plain overload: (x, y, z, eps_abs)
    assert r8.z.shape == (ni,) and np.array_equal(r8.x, r7.x)
    r9 = dense.solve(H, g, A, bvec, C, l, u, x0, y0)              # plain overload, short form
    assert r9.z.shape == (ni,)
    r10 = dense.solve(H, g, A, bvec, C, l, u, lb, ub, x0, y0, None, 1e-9)  # box overload: an array in the 11th place
    assert r10.z.shape == (ni + n,)
    r11 = dense.solve(H, g, A, bvec, C, l, u, x=x0, y=y0, eps_abs=1e-9, l_box=lb, u_box=ub)
    assert np.array_equal(r10.x, r11.x)


def case_multi_device_batch(dense, randqp, devices, B=7):
    """BatchQP spread over several devices from ONE process (one pool per device with a contiguous share of the batch,
    every pool launched before any is waited for): same answers bit for bit as QP-by-QP solves, QPs land on the
    devices in contiguous ranges, a second signature rides along, per-QP methods reach the right pool."""
    n, ne, ni = 10, 3, 4
    qps = dense.BatchQP(B, devices=devices)
    datas = [_qp_data(randqp, n, ne, ni, i) for i in range(B)]
    args = lambda d: (d["H"], d["g"], d["A"], d["b"], d["C"], d["l"], d["u"])
    for d in datas:
        qp = qps.init_qp_in_place(n, ne, ni)
        qp.settings.eps_abs = 1e-9
        qp.settings.initial_guess = dense.InitialGuess.NO_INITIAL_GUESS
        qp.init(*args(d))
    other = _qp_data(randqp, 6, 0, 5, 50)
    qo = qps.init_qp_in_place(6, 0, 5)
    qo.settings.eps_abs = 1e-9
    qo.init(other["H"], other["g"], None, None, other["C"], other["l"], other["u"])
    pools = qps._all_pools()
    G = len(devices)
    first_sig = [p for p in pools if p.batch.n == n]
    assert len(first_sig) == min(G, B) and sum(p.used for p in first_sig) == B
    assert [p.device for p in first_sig] == list(devices)[:len(first_sig)]
    assert max(p.capacity for p in first_sig) - min(p.capacity for p in first_sig) <= 1
    dense.solve_in_parallel(qps)
    for i, d in enumerate(datas):
        ref = dense.QP(n, ne, ni)
        ref.settings.eps_abs = 1e-9
        ref.settings.initial_guess = dense.InitialGuess.NO_INITIAL_GUESS
        ref.init(*args(d))
        ref.solve()
        r = qps.get(i).results
        assert r.info.status == dense.QPSolverOutput.PROXQP_SOLVED
        assert np.array_equal(r.x, ref.results.x) and np.array_equal(r.z, ref.results.z)
        assert r.info.iter == ref.results.info.iter
    assert qo.results.info.status == dense.QPSolverOutput.PROXQP_SOLVED
    # a Results object is a snapshot: a later solve of the batch does not change it
    keep = qps.get(B - 1).results
    x_before = keep.x.copy()
    qps.get(B - 1).update(g=datas[B - 1]["g"] * 2.0)
    dense.solve_in_parallel(qps)
    assert np.array_equal(keep.x, x_before) and not np.array_equal(qps.get(B - 1).results.x, x_before)
    # the vector form: QPs of several pools, one launch per pool
    vec = dense.VectorQP()
    for i in (0, B - 1, B // 2):
        vec.append(qps.get(i))
    dense.solve_in_parallel(vec)
    assert all(q.results.info.status == dense.QPSolverOutput.PROXQP_SOLVED for q in vec)


def case_batch_and_parallel(dense, oracle, randqp, B=6):
    """reference test/src/parallel_qp_solve.py / examples/python/solve_dense_qp_in_parallel.py:
    BatchQP.init_qp_in_place + solve_in_parallel == QP-by-QP solves; mixed sizes allowed."""
    shapes = [(10, 3, 4)] * B + [(6, 0, 5), (6, 0, 5)]
    qps = dense.BatchQP(B)
    datas = []
    for i, (n, ne, ni) in enumerate(shapes):
        d = _qp_data(randqp, n, ne, ni, i)
        qp = qps.init_qp_in_place(n, ne, ni)
        qp.settings.eps_abs = 1e-9
        qp.settings.initial_guess = dense.InitialGuess.NO_INITIAL_GUESS
        qp.init(d["H"], d["g"], d["A"] if ne else None, d["b"] if ne else None, d["C"], d["l"], d["u"])
        datas.append(d)
    assert qps.size() == len(shapes)
    dense.solve_in_parallel(qps, num_threads=4)
    xs = []
    for i, d in enumerate(datas):
        r = qps.get(i).results
        assert r.info.status == dense.QPSolverOutput.PROXQP_SOLVED
        assert max(_kkt(oracle, d, r.x, r.y, r.z)) <= 1e-9
        xs.append(np.array(r.x))
    # serial, standalone QPs collected in a VectorQP (reference test/src/parallel_qp_solve.cpp:33-76)
    vec = dense.VectorQP()
    for (n, ne, ni), d in zip(shapes, datas):
        qp = dense.QP(n, ne, ni)
        qp.settings.eps_abs = 1e-9
        qp.settings.initial_guess = dense.InitialGuess.NO_INITIAL_GUESS
        qp.init(d["H"], d["g"], d["A"] if ne else None, d["b"] if ne else None, d["C"], d["l"], d["u"])
        vec.append(qp)
    dense.solve_in_parallel(vec)
    for x, qp in zip(xs, vec):
        assert np.array_equal(x, qp.results.x)  # same kernel, same order of operations: bit-identical
    # a single QP of the batch re-solved alone (qps.get(i).solve(), qplayer.py:160-162)
    q3 = qps.get(3)
    q3.solve()
    assert np.array_equal(q3.results.x, xs[3])
    assert np.array_equal(qps.get(2).results.x, xs[2])
    # insert() copies a QP into the batch
    new = qps.insert(vec[0])
    assert qps.size() == len(shapes) + 1
    new.solve()
    np.testing.assert_allclose(new.results.x, xs[0], atol=1e-9)


def case_one_shot_solve(dense, oracle, randqp):
    n, ne, ni = 12, 4, 6
    d = _qp_data(randqp, n, ne, ni, 5)
    r = dense.solve(d["H"], d["g"], d["A"], d["b"], d["C"], d["l"], d["u"], eps_abs=1e-9, eps_rel=0)
    assert r.info.status == dense.QPSolverOutput.PROXQP_SOLVED
    assert max(_kkt(oracle, d, r.x, r.y, r.z)) <= 1e-9
    # warm-started one-shot call
    r2 = dense.solve(d["H"], d["g"], d["A"], d["b"], d["C"], d["l"], d["u"], r.x, r.y, r.z, eps_abs=1e-9)
    assert r2.info.iter <= 1


def case_qpfunction(QPFunction, oracle, randqp, device="cpu", B=5):
    """forward of the QPLayer (reference bindings/python/proxsuite/torch/qplayer.py:103-167) against
    QP-by-QP oracle solves configured the way the reference configures them."""
    import torch
    n, ne, ni = 10, 3, 6
    ds = [_qp_data(randqp, n, ne, ni, 100 + i) for i in range(B)]
    t = lambda k: torch.tensor(np.stack([d[k] for d in ds]), dtype=torch.float64, device=device)
    Q, p, A, b, G, u = t("H"), t("g"), t("A"), t("b"), t("C"), t("u")
    l = torch.full_like(u, -1.0e20)
    x, lam, nu = QPFunction(eps=1e-9, maxIter=1000)(Q, p, A, b, G, l, u)
    assert x.shape == (B, n) and lam.shape == (B, ne) and nu.shape == (B, ni)
    assert x.device.type == torch.device(device).type
    for i, d in enumerate(ds):
        o = oracle.QP(n, ne, ni)
        o.settings.max_iter = 1000
        o.settings.max_iter_in = 100
        o.settings.default_rho = 5e-5
        o.settings.refactor_rho_threshold = 5e-5
        o.settings.eps_abs = 1e-9
        o.init(d["H"], d["g"], d["A"], d["b"], d["C"], np.full(ni, -1e20), d["u"], rho=5e-5)
        o.solve()
        ox, oy, oz = o.results.x, o.results.y, o.results.z
        np.testing.assert_allclose(x[i].cpu().numpy(), ox, atol=1e-8)
        np.testing.assert_allclose(lam[i].cpu().numpy(), oy, atol=1e-7)
        np.testing.assert_allclose(nu[i].cpu().numpy(), oz, atol=1e-7)
    # un-batched parameters are broadcast (utils.py expandParam): shared Q, batched p
    x2, _, _ = QPFunction(eps=1e-9)(Q[0], p, A[0], b[0], G[0], l[0], u[0])
    np.testing.assert_allclose(x2[0].cpu().numpy(), x[0].cpu().numpy(), atol=1e-9)
    # closest-feasible variant returns 5 tensors with double-sided multipliers
    out = QPFunction(eps=1e-9, structural_feasibility=False)(Q, p, A, b, G, l, u)
    assert len(out) == 5 and out[2].shape == (B, ni) and out[4].shape == (B, ni)
    np.testing.assert_allclose(out[0].cpu().numpy(), x.cpu().numpy(), atol=1e-6)


def case_backward_api(dense, oracle, randqp):
    """compute_backward / solve_backward_in_parallel / model.backward_data (reference
    expose-backward.hpp, expose-parallel.hpp:48-82) against the oracle."""
    n, ne, ni, B = 9, 3, 5, 4
    qps = dense.BatchQP(B)
    datas, lds = [], dense.VectorLossDerivatives()
    rng = np.random.default_rng(2)
    for i in range(B):
        d = _qp_data(randqp, n, ne, ni, 40 + i)
        qp = qps.init_qp_in_place(n, ne, ni)
        qp.settings.eps_abs = 1e-9
        qp.init(d["H"], d["g"], d["A"], d["b"], d["C"], d["l"], d["u"])
        datas.append(d)
        ld = np.zeros(n + ne + ni)
        ld[:n] = rng.standard_normal(n)
        lds.append(ld)
    dense.solve_in_parallel(qps)
    dense.solve_backward_in_parallel(None, qps, lds, 1e-5, 1e-7, 1e-7)
    for i, d in enumerate(datas):
        o = oracle.QP(n, ne, ni)
        o.settings.eps_abs = 1e-9
        o.init(d["H"], d["g"], d["A"], d["b"], d["C"], d["l"], d["u"])
        o.solve()
        ref = o.compute_backward(lds[i], 1e-5, 1e-7, 1e-7)
        bd = qps.get(i).model.backward_data
        for k, v in ref.items():
            np.testing.assert_allclose(getattr(bd, k), v, atol=1e-6 * (1 + np.max(np.abs(v), initial=0.0)))
    # single-QP entry point
    q1 = qps.get(1)
    before = np.array(q1.model.backward_data.dL_dg)
    q1.solve()
    dense.compute_backward(q1, lds[1], 1e-5, 1e-7, 1e-7)
    np.testing.assert_allclose(q1.model.backward_data.dL_dg, before, atol=1e-9)


def case_qpfunction_backward(QPFunction, device="cpu"):
    """torch.autograd through QPFunction: gradients of L = sum(w * x*) wrt p, b, u (batched) and Q
    (shared across the batch) against central finite differences of the forward."""
    import torch
    torch.manual_seed(0)
    B, n, ne, ni = 3, 6, 2, 4
    M = torch.randn(n, n, dtype=torch.float64)
    Q = (M @ M.T + torch.eye(n, dtype=torch.float64)).to(device).requires_grad_(True)   # shared
    p = torch.randn(B, n, dtype=torch.float64, device=device, requires_grad=True)
    A = torch.randn(B, ne, n, dtype=torch.float64, device=device)
    xs = torch.randn(B, n, dtype=torch.float64, device=device)
    b = torch.einsum("bij,bj->bi", A, xs).detach().requires_grad_(True)
    G = torch.randn(B, ni, n, dtype=torch.float64, device=device)
    u = (torch.einsum("bij,bj->bi", G, xs) + 0.05).detach().requires_grad_(True)
    l = torch.full((B, ni), -1.0e20, dtype=torch.float64, device=device)
    w = torch.randn(B, n, dtype=torch.float64, device=device)
    f = QPFunction(eps=1e-10, maxIter=1000, eps_backward=1e-9, rho_backward=1e-9, mu_backward=1e-9)

    def loss(Q_, p_, b_, u_):
        x, _, _ = f(Q_, p_, A, b_, G, l, u_)
        return (w * x).sum()

    L = loss(Q, p, b, u)
    L.backward()
    h = 1e-6
    with torch.no_grad():
        for (bi, k) in [(0, 0), (1, 3), (2, 5)]:
            pp, pm = p.clone(), p.clone()
            pp[bi, k] += h; pm[bi, k] -= h
            fd = (loss(Q, pp, b, u) - loss(Q, pm, b, u)) / (2 * h)
            assert abs(fd.item() - p.grad[bi, k].item()) < 1e-5
        for (bi, k) in [(0, 1), (2, 0)]:
            bp, bm = b.clone(), b.clone()
            bp[bi, k] += h; bm[bi, k] -= h
            fd = (loss(Q, p, bp, u) - loss(Q, p, bm, u)) / (2 * h)
            assert abs(fd.item() - b.grad[bi, k].item()) < 1e-5
        for (bi, k) in [(0, 0), (1, 2), (2, 3)]:
            up, um = u.clone(), u.clone()
            up[bi, k] += h; um[bi, k] -= h
            fd = (loss(Q, p, b, up) - loss(Q, p, b, um)) / (2 * h)
            assert abs(fd.item() - u.grad[bi, k].item()) < 1e-5
        for (i, j) in [(0, 0), (1, 4)]:
            Qp, Qm = Q.clone(), Q.clone()
            Qp[i, j] += h; Qp[j, i] += h if i != j else 0
            Qm[i, j] -= h; Qm[j, i] -= h if i != j else 0
            fd = (loss(Qp, p, b, u) - loss(Qm, p, b, u)) / (2 * h)
            ref = Q.grad[i, j] + (Q.grad[j, i] if i != j else 0)
            assert abs(fd.item() - ref.item()) < 1e-5
    assert Q.grad.shape == Q.shape and p.grad.shape == p.shape


def case_nonconvex_helpers(dense):
    """estimate_minimal_eigen_value_of_symmetric_matrix + manual_minimal_H_eigenvalue (reference
    test/src/dense_qp_wrapper.cpp:7153-7617 pattern: estimate, pass to init, solve a non-convex QP
    whose minimum is pinned by box-like constraints)."""
    rng = np.random.default_rng(4)
    n = 12
    M = rng.standard_normal((n, n))
    H = (M + M.T) / 2  # indefinite
    lam = np.linalg.eigvalsh(H)[0]
    assert lam < 0
    e_exact = dense.estimate_minimal_eigen_value_of_symmetric_matrix(H)
    e_power = dense.estimate_minimal_eigen_value_of_symmetric_matrix(
        H, dense.EigenValueEstimateMethodOption.PowerIteration, 1e-10, 100000)
    assert abs(e_exact - lam) < 1e-10 and abs(e_power - lam) < 1e-6
    import pytest
    with pytest.raises(ValueError):
        dense.estimate_minimal_eigen_value_of_symmetric_matrix(H + np.triu(np.ones((n, n)), 1))
    # non-convex QP on a box: a KKT point is found once rho covers the negative curvature
    g = rng.standard_normal(n)
    C = np.eye(n)
    l, u = -np.ones(n), np.ones(n)
    qp = dense.QP(n, 0, n)
    qp.settings.eps_abs = 1e-9
    qp.init(H, g, None, None, C, l, u, manual_minimal_H_eigenvalue=e_exact)
    assert qp.settings.default_H_eigenvalue_estimate == e_exact
    qp.solve()
    r = qp.results
    assert r.info.status == dense.QPSolverOutput.PROXQP_SOLVED
    assert np.max(np.abs(H @ r.x + g + r.z)) <= 1e-9
    assert np.all(r.x <= 1 + 1e-9) and np.all(r.x >= -1 - 1e-9)


def case_timings_and_verbose(dense, randqp, capfd=None):
    """settings.compute_timings / settings.verbose (reference dense/wrapper.hpp:374-377, 495-497,
    dense/solver.hpp:1112-1115, 1783-1787, 1789-1830): Info timings are microseconds, non-zero, and
    run_time = setup_time + solve_time on a first solve; a dirty re-solve clears setup_time
    (results.hpp:157-174: cleanup_statistics); without the setting the fields stay 0."""
    n, ne, ni = 10, 3, 4
    d = _qp_data(randqp, n, ne, ni, 3)
    qp = dense.QP(n, ne, ni)
    qp.settings.eps_abs = 1e-9
    qp.init(d["H"], d["g"], d["A"], d["b"], d["C"], d["l"], d["u"])
    qp.solve()
    i = qp.results.info
    assert i.setup_time == 0 and i.solve_time == 0 and i.run_time == 0
    qp = dense.QP(n, ne, ni)
    qp.settings.eps_abs = 1e-9
    qp.settings.compute_timings = True
    qp.init(d["H"], d["g"], d["A"], d["b"], d["C"], d["l"], d["u"])
    assert qp.results.info.setup_time > 0
    qp.solve()
    i = qp.results.info
    assert i.setup_time > 0 and i.solve_time > 0
    assert abs(i.run_time - (i.setup_time + i.solve_time)) <= 1e-9 * i.run_time
    assert i.solve_time < 60e6  # microseconds: a 10-variable QP does not take a minute
    first = i.solve_time
    qp.solve()  # dirty re-solve: statistics cleaned, setup_time back to 0
    i = qp.results.info
    assert i.setup_time == 0 and i.solve_time > 0 and i.run_time == i.solve_time
    assert first > 0
    # verbose: the header and the statistics block of the reference, printed by the host after the launch
    qp.settings.verbose = True
    qp.solve()
    if capfd is not None:
        out = capfd.readouterr().out
        for word in ("SOLVER STATISTICS", "outer iter:", "total iter:", "mu updates:", "objective:", "status:         Solved",
                     "variables n = 10", "eps_abs = 1e-09", "run time"):
            assert word in out, (word, out)


def case_alias_package():
    """`proxsuite` import name (reference bindings/python/proxsuite/__init__.py, torch/qplayer.py:12-20)"""
    import proxsuite
    import proxsuite.proxqp.dense as pd
    from proxsuite.proxqp.dense import QP, BatchQP, solve_in_parallel  # noqa: F401
    from proxsuite.torch.qplayer import QPFunction
    from proxsuite_amd.proxqp import dense as impl
    from proxsuite_amd.torch import qplayer
    assert pd is impl and proxsuite.proxqp.dense is impl and QP is impl.QP
    assert QPFunction is qplayer.QPFunction
    assert proxsuite.proxqp.InitialGuess.NO_INITIAL_GUESS == impl.InitialGuess.NO_INITIAL_GUESS
    assert proxsuite.torch.QPFunction is QPFunction
