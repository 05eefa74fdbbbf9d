"""Set-up (init: Ruiz equilibration + scaled copies) time of a workload, by HIP events around pqp_batch_flush with the model
already on the device:  python scripts/gpu_setup_time.py <workload> lib1.so lib2.so ...   (+ bit comparison of the scaled models)"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from proxsuite_amd import _native as N

wl, libs = sys.argv[1], sys.argv[2:]
B, n, ne, ni, kind = bench.WORKLOADS[wl]
B = int(os.environ.get("B", B))
w = bench.Workload(kind, B, n, ne, ni)
args, kw = w.init_args()
dev = torch.device("cuda:0")
targs = [torch.as_tensor(a, device=dev) if a is not None else None for a in args]
tkw = {k: (torch.as_tensor(v, device=dev) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
res, scaled = {l: [] for l in libs}, {}
for rep in range(4):
    for l in libs:
        lib = N.NativeLib(l, legacy=True)
        b = N.Batch(B, n, ne, ni, box_constraints=w.box, hessian_type=w.hessian, lib=lib)
        b.init(-1, *targs, **tkw)   # device pointers: the copies are device-to-device
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.flush()
        e1.record()
        torch.cuda.synchronize()
        res[l].append(e0.elapsed_time(e1))
        scaled[l] = [b.scaled(i) for i in (0, B - 1)]
        b.close()
for l in libs:
    print("%-6s %-28s flush (copies + set-up kernel) %s ms" % (wl, os.path.basename(l), " ".join("%.3f" % v for v in res[l][1:])))
if len(libs) > 1:
    a, c = scaled[libs[0]], scaled[libs[1]]
    print("scaled models bit-identical:", all(np.array_equal(np.asarray(a[i][k]), np.asarray(c[i][k])) for i in range(2) for k in a[i]))
