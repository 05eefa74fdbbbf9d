"""HBM traffic of one phase of the C2 solve: run the instrumented library with
PQP_REPEAT_PHASE=<k> PQP_REPEAT_COUNT=2 under `rocprofv3 --pmc FETCH_SIZE` (or WRITE_SIZE) and compare with the
plain run.  This script is the workload under the profiler: B QPs of the C2 shape, `reps` solves, and one JSON line
with the engine-byte and event counters of the last solve (so that traffic delta / engine-byte delta is per phase).
  python scripts/gpu_phase_traffic.py <libproxqp_hip_stats.so> [B] [reps]        (PQP_SHAPE=n,n_eq,n_in: another shape than C2's)"""
import json
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proxsuite_amd import _native as N
from proxsuite_amd.utils import random_qp as R

lib = N.NativeLib(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n, ne, ni = (int(v) for v in os.environ.get("PQP_SHAPE", "100,50,100").split(","))
m = R.dense_strongly_convex_qp_batch(B, n, ne, ni, 0.15, 1e-2, seed0=0)
b = N.Batch(B, n, ne, ni, lib=lib)
for i in range(B):
    s = b.settings(i)
    s.eps_abs, s.eps_rel, s.initial_guess = 1e-9, 0.0, 0
b.init(-1, m.H, m.g, m.A, m.b, m.C, m.l, m.u)
b.flush()
ms = []
for _ in range(reps):
    b.solve()
    ms.append(b.last_solve_ms)
st = b.stats().astype(np.float64).mean(axis=0)
d = dict(zip(N.STAT_NAMES, st))
x, y, z, se, si, info = b.results()
print(json.dumps({"phase": int(os.environ.get("PQP_REPEAT_PHASE", "0")), "count": int(os.environ.get("PQP_REPEAT_COUNT", "1")),
                  "B": B, "solves": reps, "kernel_ms": ms, "bytes_engine": d["bytes_engine"], "n_newton": d["n_newton"],
                  "n_schur_fact": d["n_schur_fact"], "n_append": d["n_append"], "n_delete": d["n_delete"],
                  "unsolved": int(sum(1 for i in range(B) if info[i].status != 0)),
                  "iter_sum": int(sum(info[i].iter for i in range(B)))}))
