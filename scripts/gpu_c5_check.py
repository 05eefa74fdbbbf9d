"""the diagonal-structure kernel against the oracle: python scripts/gpu_c5_check.py dim B box(0/1)   (PQP_DIAG_NT=64|256 picks the kernel; LIB=<so> another build / the emulator)"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import parity_cases as pc
from proxsuite_amd import _native as N
from proxsuite_amd.utils import random_qp as randqp
from oracle import oracle
from proxsuite_amd._ctypes_defs import HessianType, InitialGuess
lib = N.NativeLib(os.environ["LIB"], legacy=True) if os.environ.get("LIB") else N.load()
dim = int(sys.argv[1]); B = int(sys.argv[2]); box = int(sys.argv[3])
H, g, Cm, l, u = pc.c5_models(randqp, B, dim)
hess = HessianType.Diagonal
if box:
    b = N.Batch(B, dim, 0, 0, box_constraints=True, hessian_type=int(hess), lib=lib)
    pc.settings_all(b, eps_abs=1e-9, eps_rel=0, initial_guess=int(InitialGuess.NO_INITIAL_GUESS))
    b.init(-1, H, g, None, None, None, None, None, l, u)
else:
    b = N.Batch(B, dim, 0, dim, hessian_type=int(hess), lib=lib)
    pc.settings_all(b, eps_abs=1e-9, eps_rel=0, initial_guess=int(InitialGuess.NO_INITIAL_GUESS))
    b.init(-1, H, g, None, None, Cm, l, u)
b.solve()
x, y, z, se, si, info = b.results()
idx = list(range(B))
if box:
    qs = pc.oracle_solve_many(oracle, [(H[i], g[i], None, None, None, None, None, l[i], u[i]) for i in idx], dim, 0, 0, box_constraints=True, hessian_type=hess)
else:
    qs = pc.oracle_solve_many(oracle, [(H[i], g[i], None, None, Cm[i], l[i], u[i]) for i in idx], dim, 0, dim, hessian_type=hess)
bad = 0
for i, q in zip(idx, qs):
    same = (info[i].iter, info[i].iter_ext, info[i].mu_updates, info[i].status) == (q.results.info.iter, q.results.info.iter_ext, q.results.info.mu_updates, int(q.results.info.status))
    d = max(np.max(np.abs(x[i]-q.results.x)), np.max(np.abs(z[i]-q.results.z)))
    if same and d < 1e-10:
        continue
    bad += 1
    print(i, "dx %.3e dz %.3e" % (np.max(np.abs(x[i]-q.results.x)), np.max(np.abs(z[i]-q.results.z))), "iter", info[i].iter, q.results.info.iter, "ext", info[i].iter_ext, q.results.info.iter_ext, "mu_upd", info[i].mu_updates, q.results.info.mu_updates, "status", info[i].status)

print("dim", dim, "B", B, "box", box, "mismatching QPs:", bad)
