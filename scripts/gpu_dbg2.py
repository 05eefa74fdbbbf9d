import sys, numpy as np
sys.path.insert(0, ".")
from proxsuite_amd import _native as N
from proxsuite_amd.utils import random_qp as R
n, ne, ni, B = map(int, sys.argv[1:5])
lib = N.load()
m = R.dense_strongly_convex_qp_batch(B, n, ne, ni, 0.15, 1e-2)
b = N.Batch(B, n, ne, ni, lib=lib)
for i in range(B):
    s = b.settings(i); s.eps_abs = 1e-9; s.initial_guess = 0; s.max_iter = 100
b.init(-1, m.H, m.g, m.A, m.b, m.C, m.l, m.u)
b.solve()
x, y, z, se, si, info = b.results()
st = b.stats()
print((n, ne, ni), "status", [info[i].status for i in range(B)], "iter", [info[i].iter for i in range(B)], "newton", st[:, 10], "schur", st[:, 11], "pri", [float("%.2e" % info[i].pri_res) for i in range(B)], flush=True)
