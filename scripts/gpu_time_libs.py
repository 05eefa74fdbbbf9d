"""Kernel time of the solve (HIP events, pqp_batch_last_solve_ms) for several builds of the library on one box,
interleaved -- uses only the entries every build since round 1 has, so old and new libraries can be compared:
  python scripts/gpu_time_libs.py <workload> <rounds> lib1.so lib2.so ...        (B=<n> in the environment: other batch size)"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from proxsuite_amd import _native as N

wl, rounds, libs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
B, n, ne, ni, kind = bench.WORKLOADS[wl]
B = int(os.environ.get("B", B))
w = bench.Workload(kind, B, n, ne, ni)
args, kw = w.init_args()
res = {l: [] for l in libs}
pro = {l: [] for l in libs}  # the factorisation prologue's part (two-kernel launches of the one-wavefront dense kernel)
for r in range(rounds):
    for l in libs:
        lib = N.NativeLib(l, legacy=True)
        b = N.Batch(B, n, ne, ni, box_constraints=w.box, hessian_type=w.hessian, lib=lib)
        for i in range(B):
            s = b.settings(i)
            s.eps_abs, s.eps_rel, s.initial_guess = 1e-9, 0.0, 0
        b.init(-1, *args, **kw)
        b.flush()
        ms, pm = [], []
        for k in range(8):
            b.solve()
            ms.append(b.last_solve_ms)
            try:
                pm.append(b.last_prologue_ms)
            except AttributeError:
                pm.append(0.0)
        res[l].append(float(np.mean(ms[2:])))
        pro[l].append(float(np.mean(pm[2:])))
        b.close()
for l in libs:
    print("%-6s %-28s %s  mean %.3f ms  %.0f QPs/s  (prologue %.3f ms)" % (wl, os.path.basename(l), " ".join("%.3f" % v for v in res[l]), np.mean(res[l]), B / np.mean(res[l]) * 1e3, np.mean(pro[l])))
