"""Kernel time of a shape outside bench.WORKLOADS for several builds, interleaved:
  python scripts/gpu_time_shape.py B n n_eq n_in box(0/1) backend(0/1/2) rounds lib1.so lib2.so ...   (+ bit comparison of the first two)"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proxsuite_amd import _native as N
from proxsuite_amd.utils import random_qp as R

B, n, ne, ni, box, backend, rounds = map(int, sys.argv[1:8])
libs = sys.argv[8:]
m = R.dense_strongly_convex_qp_batch(B, n, ne, ni, 0.15, 1e-2)
kw = {}
if box:
    rng = np.random.default_rng(3)
    kw = dict(l_box=-1.0 - rng.random((B, n)), u_box=1.0 + rng.random((B, n)))
res = {l: [] for l in libs}
out = {}
for r in range(rounds):
    for l in libs:
        lib = N.NativeLib(l, legacy=True)
        b = N.Batch(B, n, ne, ni, box_constraints=bool(box), dense_backend=backend, lib=lib)
        for i in range(B):
            s = b.settings(i)
            s.eps_abs, s.eps_rel, s.initial_guess = 1e-9, 0.0, 0
        b.init(-1, m.H, m.g, m.A, m.b, m.C, m.l, m.u, **kw)
        b.flush()
        ms = []
        for k in range(8):
            b.solve()
            ms.append(b.last_solve_ms)
        res[l].append(float(np.mean(ms[2:])))
        out[l] = [a.copy() for a in b.results()[:3]]
        b.close()
for l in libs:
    print("%s %-28s %s  mean %.3f ms" % (tuple(sys.argv[1:7]), os.path.basename(l), " ".join("%.3f" % v for v in res[l]), np.mean(res[l])))
if len(libs) > 1:
    print("bit-identical:", all(np.array_equal(a, c) for a, c in zip(out[libs[0]], out[libs[1]])))
