import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from proxsuite_amd import _native as N
lib = N.NativeLib(os.environ["LIB"], legacy=True) if os.environ.get("LIB") else N.load()
B = int(os.environ.get("B", 4096))
Bw, n, ne, ni, kind = bench.WORKLOADS["c5"]
w = bench.Workload(kind, B, n, ne, ni)
args, kw = w.init_args()
b = N.Batch(B, n, ne, ni, box_constraints=w.box, hessian_type=w.hessian, lib=lib)
b.set_all_settings(eps_abs=1e-9, eps_rel=0.0, initial_guess=0)
b.init(-1, *args, **kw)
b.flush()
ms = []
for k in range(12):
    t0 = time.perf_counter(); b.solve(); t1 = time.perf_counter()
    ms.append((b.last_solve_ms, 1e3 * (t1 - t0)))
print("cold GPU :", " ".join("%.2f/%.2f" % m for m in ms))
print(N.box_calibration(0, N.load()))
ms = []
for k in range(12):
    t0 = time.perf_counter(); b.solve(); t1 = time.perf_counter()
    ms.append((b.last_solve_ms, 1e3 * (t1 - t0)))
print("after cal:", " ".join("%.2f/%.2f" % m for m in ms))
st = b.stats()
print("cycles total per QP: mean %.0f max %.0f" % (st[:, 0].mean(), st[:, 0].max()))
