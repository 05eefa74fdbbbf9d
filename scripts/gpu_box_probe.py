"""One line per box: pqp_box_calibrate next to the kernel times of the BASELINE configurations on the same box (how
profiles/r05_box_calibration.txt and the reference box of profiles/perf_guard.json were made).
  python scripts/gpu_box_probe.py [c2 c5 c4 c1 ...]   -> appends to gpurun_out/box_probe.jsonl"""
import json
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as BN  # noqa: E402
from proxsuite_amd import _native as N  # noqa: E402
from proxsuite_amd._ctypes_defs import InitialGuess  # noqa: E402

lib = N.load()
which = sys.argv[1:] or ["c2", "c5", "c4", "c1"]
rec = {"host": socket.gethostname(), "time": time.strftime("%Y-%m-%dT%H:%M:%S"), "lib": os.environ.get("PQP_HIP_LIBRARY", "product")}
rec["cal0"] = N.box_calibration(0, lib)
rec["smi"] = BN._smi()
for wname in which:
    B, n, ne, ni, kind = BN.WORKLOADS[wname]
    w = BN.Workload(kind, B, n, ne, ni)
    b = N.Batch(B, n, ne, ni, box_constraints=w.box, hessian_type=w.hessian, lib=lib)
    b.set_all_settings(eps_abs=1e-9, eps_rel=0.0, initial_guess=int(InitialGuess.NO_INITIAL_GUESS))
    args, kw = w.init_args()
    b.init(-1, *args, **kw)
    b.flush()
    b.solve()
    ms = []
    for _ in range(6 if wname != "c4" else 3):
        b.solve()
        ms.append(b.last_solve_ms)
    infos = b.infos()
    rec[wname] = {"kernel_ms_min": float(min(ms)), "kernel_ms_mean": float(np.mean(ms)), "unsolved": int(sum(1 for i in range(B) if infos[i].status != 0))}
    b.close()
rec["cal1"] = N.box_calibration(0, lib)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "box_probe.jsonl"), "a") as f:
    f.write(json.dumps(rec) + "\n")
c0, c1 = rec["cal0"], rec["cal1"]
print("BOX %s chain %.3f/%.3f ms hbm %.0f/%.0f GB/s sclk~%.0f/%.0f MHz valu %.3f ms | %s" % (
    rec["host"], c0["chain_ms"], c1["chain_ms"], c0["hbm_read_gbs"], c1["hbm_read_gbs"], c0["sclk_mhz_est"], c1["sclk_mhz_est"], c0["valu_ms"],
    " ".join("%s %.3f" % (k, rec[k]["kernel_ms_min"]) for k in which)))
