#!/usr/bin/env python
"""Per-phase device cycles and event counts of a C2-shaped batch (instrumented library: PQP_HIP_LIBRARY = the -DPQP_STATS
build), for the kernel PQP_DENSE_KERNEL selects.   python scripts/gpu_dwave_phases.py [B]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proxsuite_amd import _native as N  # noqa: E402
from proxsuite_amd._ctypes_defs import InitialGuess  # noqa: E402
from proxsuite_amd.utils import random_qp as R  # noqa: E402

NAMES = ["CYC_TOTAL", "CYC_SCALE", "CYC_FACTOR_H", "CYC_ZG", "CYC_SCHUR", "CYC_KKT_SOLVE", "CYC_RESIDUAL", "CYC_LINESEARCH",
         "CYC_GLOBAL_RES", "CYC_NEWTON_MISC", "N_NEWTON", "N_SCHUR_FACT", "N_NEW_ROWS", "N_KKT_SOLVES", "N_LS_BREAKPOINTS",
         "N_ACTIVE_FINAL", "CYC_F_LOAD", "CYC_F_UPDATE", "CYC_F_PANEL", "CYC_F_WRITEBACK", "CYC_F_TINV", "CYC_S_GATHER",
         "CYC_SOLVE_LDLT", "N_SCHUR_BLOCKED", "N_APPEND", "N_DELETE", "BYTES_ENGINE", "N_REFACTORIZE", "CYC_LS_EVAL", "CYC_CERT",
         "CYC_UPDATE", "WALL_TICKS", "FLOPS_FACT"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n, ne, ni = 100, 50, 100
lib = N.load()
m = R.dense_strongly_convex_qp_batch(B, n, ne, ni, 0.15, 1e-2)
b = N.Batch(B, n, ne, ni, lib=lib)
for i in range(B):
    s = b.settings(i)
    s.eps_abs, s.eps_rel, s.initial_guess = 1e-9, 0.0, int(InitialGuess.NO_INITIAL_GUESS)
b.init(-1, m.H, m.g, m.A, m.b, m.C, m.l, m.u)
b.solve()
b.solve()
st = b.stats().astype(np.float64)
print("library", lib.path, "kernel", os.environ.get("PQP_DENSE_KERNEL"), "last_solve_ms", b.last_solve_ms)
tot = st[:, 0].mean()
for k, name in enumerate(NAMES):
    v = st[:, k]
    if name.startswith("CYC"):
        print("%-18s mean %12.0f  (%5.1f %% of total)  max %12.0f" % (name, v.mean(), 100 * v.mean() / tot, v.max()))
    else:
        print("%-18s mean %12.1f  max %12.0f" % (name, v.mean(), v.max()))
