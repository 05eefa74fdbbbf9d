#!/usr/bin/env python
"""Per-problem KKT residuals of the small Maros-Meszaros fixtures on the GPU (both solves of
tests/parity_cases.case_maros_meszaros), to see how close each one sits to the 2*eps acceptance
line of the reference's test (test/src/dense_maros_meszaros.cpp:85-169)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import split_maros  # noqa: E402
from proxsuite_amd import _native as N  # noqa: E402
from proxsuite_amd._ctypes_defs import InitialGuess  # noqa: E402

lib = N.load()
gold = np.load(os.path.join(ROOT, "tests", "golden", "maros_meszaros_small.npz"))
names = [str(s) for s in gold["names"]]
eps = 2e-8
for name in names:
    P, q, A, l, u = (gold["%s/%s" % (name, f)] for f in "PqAlu")
    H, g, Aeq, b, C, lin, uin = split_maros(P, q, A, l, u)
    n, n_eq, n_in = H.shape[0], Aeq.shape[0], C.shape[0]
    bt = N.Batch(1, n, n_eq, n_in, lib=lib)
    bt.init(0, H, g, Aeq, b, C, lin, uin)
    s = bt.settings(0)
    s.eps_abs, s.eps_rel, s.eps_primal_inf, s.eps_dual_inf, s.max_iter = eps, 0, 1e-12, 1e-12, 1000
    out = []
    for it in range(2):
        if it > 0:
            s.initial_guess = InitialGuess.WARM_START_WITH_PREVIOUS_RESULT
        bt.solve()
        x, y, z, se, si, info = bt.results(0)
        dua = H @ x + g + (Aeq.T @ y if n_eq else 0) + (C.T @ z if n_in else 0)
        out.append("dua %.2e (solver %.2e) iter %d status %d" % (np.max(np.abs(dua)), info.dua_res, info.iter, info.status))
    print("%-10s n=%3d n_eq=%3d n_in=%3d |H|max %.1e : %s" % (name, n, n_eq, n_in, np.max(np.abs(H)), " | ".join(out)), flush=True)
