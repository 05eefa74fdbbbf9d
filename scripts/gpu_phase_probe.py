"""Per-call device cycles of the blocked Schur factorisation at C4 for a given -DPQP_STATS build of the library:
python scripts/gpu_phase_probe.py <libstats.so> [max_iter]   (timing probe: also used with builds that skip a phase and
therefore compute garbage -- max_iter bounds those runs)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from proxsuite_amd import _native as N
from proxsuite_amd.utils import random_qp as R

lib = N.NativeLib(sys.argv[1])
max_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
B, n, ne, ni = 256, 512, 200, 400
m = R.dense_strongly_convex_qp_batch(B, n, ne, ni, 0.15, 1e-2, seed0=1)
b = N.Batch(B, n, ne, ni, lib=lib)
for i in range(B):
    s = b.settings(i)
    s.eps_abs, s.eps_rel, s.initial_guess, s.max_iter = 1e-9, 0.0, 0, max_iter
b.init(-1, m.H, m.g, m.A, m.b, m.C, m.l, m.u)
b.solve()
st = b.stats().astype(np.float64).mean(axis=0)
names = N.STAT_NAMES
d = dict(zip(names, st))
k = max(d["n_schur_blocked"], 1e-9)
print("%s: schur factorisations %.2f per QP; per factorisation: gather %.0f  ldlt %.0f  inverse %.0f cycles; primal panel %.0f tinv %.0f; total %.0f"
      % (sys.argv[1].split("/")[-1], d["n_schur_blocked"], d["cyc_f_load"] / k, d["cyc_f_update"] / k, d["cyc_f_writeback"] / k,
         d["cyc_f_panel"], d["cyc_f_tinv"], d["cyc_total"]))
