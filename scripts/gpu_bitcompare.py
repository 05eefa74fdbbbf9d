"""Are two builds of the library bit-identical on a workload?  python scripts/gpu_bitcompare.py <workload> libA.so libB.so"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from proxsuite_amd import _native as N

wl = sys.argv[1]
B, n, ne, ni, kind = bench.WORKLOADS[wl]
B = int(os.environ.get("B", B))
w = bench.Workload(kind, B, n, ne, ni)
args, kw = w.init_args()
out = []
for l in sys.argv[2:4]:
    lib = N.NativeLib(l, legacy=True)
    b = N.Batch(B, n, ne, ni, box_constraints=w.box, hessian_type=w.hessian, lib=lib)
    for i in range(B):
        s = b.settings(i)
        s.eps_abs, s.eps_rel, s.initial_guess = 1e-9, 0.0, 0
    b.init(-1, *args, **kw)
    b.solve()
    x, y, z, se, si, info = b.results()
    out.append((x, y, z, [(info[i].iter, info[i].iter_ext, info[i].status) for i in range(B)]))
    b.close()
same = all(np.array_equal(a, c) for a, c in zip(out[0][:3], out[1][:3])) and out[0][3] == out[1][3]
print(wl, "bit-identical:", same, "max |dx| %.2e" % float(np.max(np.abs(out[0][0] - out[1][0]))))
