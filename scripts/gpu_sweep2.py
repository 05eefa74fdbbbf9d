"""Randomised robustness sweep, round 2: shapes x {box constraints, Dense / Diagonal Hessian, DenseBackend
Automatic / PrimalDualLDLT / PrimalLDLT} x {cold solve, then update(g) + WARM_START_WITH_PREVIOUS_RESULT
re-solve, which restores the edited Schur factor -- holes included -- from HBM}.  Every QP must end with the
oracle's status; SOLVED ones must have KKT <= 1e-9 and x equal to the oracle's; a QP the oracle does not solve
is only noted.  Usage: python scripts/gpu_sweep2.py <seed> <count> [library.so]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from proxsuite_amd import _native as N
from proxsuite_amd._ctypes_defs import InitialGuess, HessianType, DenseBackend
from proxsuite_amd.utils import random_qp as R
from oracle import oracle as O
import parity_cases as pc

lib = N.NativeLib(sys.argv[3]) if len(sys.argv) > 3 else N.load()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = notes = forks = 0
t0 = time.time()
for it in range(count):
    n = int(rng.integers(1, 120))
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
    B = 3
    m = R.dense_strongly_convex_qp_batch(B, n, ne, ni, float(rng.uniform(0.1, 0.9)), 1e-2, seed0=int(rng.integers(0, 10000)))
    H = m.H if hess == HessianType.Dense else np.stack([np.diag(np.diag(h)) for h in m.H])
    lb = ub = None
    if box:
        xs = rng.standard_normal((B, n)); sh = rng.uniform(0.1, 1.0, (B, n))
        lb, ub = xs - sh, xs + sh
    g2 = m.g + 0.1 * rng.standard_normal(m.g.shape)
    b = N.Batch(B, n, ne, ni, box_constraints=box, hessian_type=int(hess), dense_backend=int(backend), lib=lib)
    qs = []
    for i in range(B):
        s = b.settings(i); s.eps_abs = 1e-9; s.eps_rel = 0; s.initial_guess = int(InitialGuess.NO_INITIAL_GUESS); s.max_iter = 2000
        q = O.QP(n, ne, ni, box_constraints=box, hessian_type=hess, dense_backend=backend)
        q.settings.eps_abs = 1e-9; q.settings.eps_rel = 0; q.settings.initial_guess = InitialGuess.NO_INITIAL_GUESS; q.settings.max_iter = 2000
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
                bad += 1; print("FAIL status", tag, info[i].status, r.info.status, flush=True); continue
            if info[i].status != 0:
                notes += 1; continue
            pri, dua = pc.kkt_numpy(H[i], gcur[i], m.A[i], m.b[i], m.C[i], m.l[i], m.u[i], x[i], y[i], z[i],
                                    lb[i] if box else None, ub[i] if box else None)
            if not (pri <= 1e-9 and dua <= 1e-9 and pc.close(x[i], r.x)):
                bad += 1; print("FAIL", tag, pri, dua, float(np.max(np.abs(x[i] - r.x))), flush=True)
    b.close()
print("sweep2: %d shapes x 3 QPs x 2 phases, %d failures, %d QPs unsolved alike in the oracle, %d infeasible QPs "
      "ending with different non-SOLVED statuses, %.1f s" % (count, bad, notes, forks, time.time() - t0))
