"""Wall time of ONE QP through the Python facade (what a user of `QP::solve()` sees), against the kernel time of the
same solve: python scripts/gpu_single_qp_latency.py [n n_eq n_in]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proxsuite_amd import proxqp
from proxsuite_amd.utils import random_qp as R

n, ne, ni = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (100, 50, 100)
m = R.dense_strongly_convex_qp_batch(4, n, ne, ni, 0.15, 1e-2)


def one(i, fresh):
    t0 = time.perf_counter()
    qp = proxqp.dense.QP(n, ne, ni)
    qp.settings.eps_abs, qp.settings.eps_rel = 1e-9, 0
    qp.settings.initial_guess = proxqp.InitialGuess.NO_INITIAL_GUESS
    t1 = time.perf_counter()
    qp.init(m.H[i], m.g[i], m.A[i], m.b[i], m.C[i], m.l[i], m.u[i])
    t2 = time.perf_counter()
    qp.solve()
    t3 = time.perf_counter()
    x = qp.results.x.copy()
    t4 = time.perf_counter()
    # update + warm re-solve (the MPC pattern)
    qp.settings.initial_guess = proxqp.InitialGuess.WARM_START_WITH_PREVIOUS_RESULT
    qp.update(g=m.g[i] * 1.1)
    t5 = time.perf_counter()
    qp.solve()
    t6 = time.perf_counter()
    return dict(create=t1 - t0, init=t2 - t1, solve=t3 - t2, read=t4 - t3, update=t5 - t4, resolve=t6 - t5, iter=qp.results.info.iter,
                run_time_us=qp.results.info.run_time), qp


keep = []
for rep in range(4):
    r, qp = one(rep % 4, rep == 0)
    keep.append(qp)
    print("rep %d: " % rep + "  ".join("%s %.3f ms" % (k, v * 1e3) for k, v in r.items() if k not in ("iter", "run_time_us")),
          " iter", r["iter"])
# the functional form
t0 = time.perf_counter()
res = proxqp.dense.solve(m.H[0], m.g[0], m.A[0], m.b[0], m.C[0], m.l[0], m.u[0], eps_abs=1e-9, eps_rel=0)
print("dense.solve(...): %.3f ms" % ((time.perf_counter() - t0) * 1e3))
