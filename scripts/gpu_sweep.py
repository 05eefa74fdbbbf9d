"""Randomised robustness sweep on the GPU: many shapes (odd sizes, every workgroup size, with and
without boxes / equalities / inequalities, all initial guesses), every QP must be SOLVED with
KKT <= 1e-9, and a subset is compared with the oracle."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from proxsuite_amd import _native as N
from proxsuite_amd._ctypes_defs import InitialGuess
from proxsuite_amd.utils import random_qp as R
from oracle import oracle as O

lib = N.load()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
t0 = time.time()
for it in range(count):
    n = int(rng.integers(1, 140))
    ne = int(rng.integers(0, max(1, n // 2) + 1))
    ni = int(rng.integers(0, 2 * n + 2))
    if ne + ni == 0:
        ni = 1
    if ne + ni + n > 900:
        ni = max(0, 900 - n - ne)
    B = 4
    guess = int(rng.integers(0, 2))  # NO_INITIAL_GUESS / EQUALITY_CONSTRAINED
    m = R.dense_strongly_convex_qp_batch(B, n, ne, ni, float(rng.uniform(0.1, 0.9)), 1e-2, seed0=int(rng.integers(0, 10000)))
    b = N.Batch(B, n, ne, ni, lib=lib)
    for i in range(B):
        s = b.settings(i); s.eps_abs = 1e-9; s.eps_rel = 0; s.initial_guess = guess; s.max_iter = 2000
    b.init(-1, m.H, m.g, m.A if ne else None, m.b if ne else None, m.C if ni else None, m.l if ni else None, m.u if ni else None)
    b.solve()
    x, y, z, se, si, info = b.results()
    for i in range(B):
        pri, dua = O.kkt_residuals(m.H[i], m.g[i], m.A[i], m.b[i], m.C[i], m.l[i], m.u[i], x[i], y[i], z[i])
        ok = info[i].status == 0 and pri <= 1e-9 and dua <= 1e-9
        if ok and i == 0:
            q = O.QP(n, ne, ni)
            q.settings.eps_abs = 1e-9; q.settings.eps_rel = 0; q.settings.initial_guess = guess
            q.init(m.H[i], m.g[i], m.A[i] if ne else None, m.b[i] if ne else None, m.C[i] if ni else None,
                   m.l[i] if ni else None, m.u[i] if ni else None)
            q.solve()
            ok = np.max(np.abs(x[i] - q.results.x)) <= 1e-7 * (1 + np.max(np.abs(q.results.x)))
        if not ok:
            # not solved: acceptable only if the reference algorithm (oracle) ends the same way
            q = O.QP(n, ne, ni)
            q.settings.eps_abs = 1e-9; q.settings.eps_rel = 0; q.settings.initial_guess = guess; q.settings.max_iter = 2000
            q.init(m.H[i], m.g[i], m.A[i] if ne else None, m.b[i] if ne else None, m.C[i] if ni else None,
                   m.l[i] if ni else None, m.u[i] if ni else None)
            q.solve()
            if q.results.info.status == info[i].status and q.results.info.iter == info[i].iter:
                print("note: shape", (n, ne, ni), "qp", i, "ends with status", info[i].status, "after", info[i].iter,
                      "iterations on the device AND in the oracle", flush=True)
                continue
            bad += 1
            print("FAIL shape", (n, ne, ni), "guess", guess, "qp", i, "status", info[i].status, "iter", info[i].iter, pri, dua, flush=True)
    b.close()
print("sweep: %d shapes, %d failures, %.1f s" % (count, bad, time.time() - t0))
