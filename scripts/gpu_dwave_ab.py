#!/usr/bin/env python
"""A/B on one box: the one-wavefront dense kernel (PQP_DENSE_KERNEL=wave) against the 256-thread workgroup kernel
(PQP_DENSE_KERNEL=workgroup) on C2-shaped batches: kernel time per batch (device events, interleaved repetitions),
agreement of the two kernels QP by QP (x, y, z, Info counters), worst KKT residual.
    python scripts/gpu_dwave_ab.py [B] [reps] [n ne ni]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proxsuite_amd import _native as N  # noqa: E402
from proxsuite_amd._ctypes_defs import InitialGuess  # noqa: E402
from proxsuite_amd.utils import random_qp as R  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n, ne, ni = (int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (100, 50, 100)
lib = N.load()
print("library:", lib.path, "B", B, "shape", (n, ne, ni), flush=True)
m = R.dense_strongly_convex_qp_batch(B, n, ne, ni, 0.15, 1e-2)
b = N.Batch(B, n, ne, ni, lib=lib)
for i in range(B):
    s = b.settings(i)
    s.eps_abs, s.eps_rel, s.initial_guess = 1e-9, 0.0, int(InitialGuess.NO_INITIAL_GUESS)
b.init(-1, m.H, m.g, m.A, m.b, m.C, m.l, m.u)
res = {}
MODES = tuple(os.environ.get("PQP_AB_MODES", "workgroup,wave").split(","))
times = {k: [] for k in MODES}
for rep in range(reps):
    for mode in MODES:
        os.environ["PQP_DENSE_KERNEL"] = mode
        b.solve()
        times[mode].append(b.last_solve_ms)
        if rep == 0:
            res[mode] = b.results()
for mode in times:
    t = times[mode]
    print("%-10s kernel ms: %s   best %.3f  median %.3f  -> %.1f k QPs/s" % (mode, " ".join("%.3f" % v for v in t), min(t),
          float(np.median(t)), B / float(np.median(t))), flush=True)
if len(MODES) < 2:
    sys.exit(0)
xa, ya, za, _, _, ia = res["workgroup"]
xb, yb, zb, _, _, ib = res["wave"]
dx = float(np.max(np.abs(xa - xb)))
dy = float(np.max(np.abs(ya - yb))) if ya.size else 0.0
dz = float(np.max(np.abs(za - zb))) if za.size else 0.0
bad = [i for i in range(B) if (ia[i].status, ia[i].iter, ia[i].iter_ext, ia[i].mu_updates) != (ib[i].status, ib[i].iter, ib[i].iter_ext, ib[i].mu_updates)]
worst = 0.0
for i in range(B):
    r = m.H[i] @ xb[i] + m.g[i] + m.A[i].T @ yb[i] + m.C[i].T @ zb[i]
    worst = max(worst, float(np.max(np.abs(r))))
print("wave vs workgroup: max|dx| %.2e |dy| %.2e |dz| %.2e; Info counters differ on %d of %d QPs %s; solved %d; worst dual residual %.2e"
      % (dx, dy, dz, len(bad), B, bad[:8], sum(1 for i in range(B) if ib[i].status == 0), worst), flush=True)
print("iter mean %.2f max %d" % (np.mean([ib[i].iter for i in range(B)]), max(ib[i].iter for i in range(B))))
