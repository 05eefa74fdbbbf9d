#!/usr/bin/env python
"""Fast on-GPU sanity check of one build of the HIP library (PQP_HIP_LIBRARY selects it): a few
small batches, statuses / iteration counts / worst KKT residual, a hard iteration cap so that a
broken kernel costs seconds, not minutes.   python scripts/gpu_quickcheck.py [max_iter]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proxsuite_amd import _native as N  # noqa: E402
from proxsuite_amd._ctypes_defs import InitialGuess  # noqa: E402
from proxsuite_amd.utils import random_qp as R  # noqa: E402

max_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 200
lib = N.load()
print("library:", lib.path)
for (B, n, ne, ni) in [(16, 30, 7, 9), (64, 100, 50, 100), (8, 50, 0, 20), (8, 120, 30, 0)]:
    m = R.dense_strongly_convex_qp_batch(B, n, ne, ni, 0.15, 1e-2)
    b = N.Batch(B, n, ne, ni, lib=lib)
    for i in range(B):
        s = b.settings(i)
        s.eps_abs, s.eps_rel, s.initial_guess, s.max_iter = 1e-9, 0.0, int(InitialGuess.NO_INITIAL_GUESS), max_iter
    b.init(-1, m.H, m.g, m.A, m.b, m.C, m.l, m.u)
    t0 = time.perf_counter()
    b.solve()
    x, y, z, se, si, info = b.results()
    dt = time.perf_counter() - t0
    worst = 0.0
    for i in range(B):
        r = m.H[i] @ x[i] + m.g[i] + m.A[i].T @ y[i] + m.C[i].T @ z[i]
        worst = max(worst, float(np.max(np.abs(r))) if r.size else 0.0)
    st = [int(info[i].status) for i in range(B)]
    it = [int(info[i].iter) for i in range(B)]
    print("shape %s: solved %d/%d  iter max %d mean %.1f  worst dual residual %.2e  %.1f ms"
          % ((B, n, ne, ni), sum(1 for v in st if v == 0), B, max(it), sum(it) / B, worst, dt * 1e3), flush=True)
