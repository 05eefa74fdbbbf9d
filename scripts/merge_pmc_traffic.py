#!/usr/bin/env python
"""gpurun_out/pmc_traffic_<w>.json (scripts/gpu_pmc_traffic.sh) -> profiles/<tag>_pmc_<w>.json and the
per-workload table profiles/pmc_traffic.json that bench.py reports as `roofline.traffic`."""
import glob
import json
import os
import subprocess
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
table_path = os.path.join(root, "profiles", "pmc_traffic.json")
table = {}
if os.path.exists(table_path):
    old = json.load(open(table_path))
    table = old if "workloads" in old else {"workloads": {}}
table.setdefault("workloads", {})
for f in sorted(glob.glob(os.path.join(root, "gpurun_out", "pmc_traffic_*.json"))):
    j = json.load(open(f))
    w = j["workload"]
    dst = os.path.join(root, "profiles", "%s_pmc_%s.json" % (tag, w))
    json.dump(j, open(dst, "w"), indent=1)
    if "hbm_bytes_per_launch" in j:
        sys.path.insert(0, root)
        from proxsuite_amd import _build
        diag = any("diag" in k for k in j["kernel"])
        dwave = any("dwave" in k for k in j["kernel"])
        try:
            commit = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
        except Exception:
            commit = None
        table["workloads"][w] = {"hbm_bytes_per_launch": j["hbm_bytes_per_launch"], "kernel": j["kernel"],
                                 # what it was measured on: bench.py prints traffic_stale when the kernel sources have changed since
                                 "kernel_sources_sha": _build.kernel_sources_sha(diag, dwave), "measured_at_commit": commit,
                                 **({"per_kernel": {k: {"hbm_bytes": v.get("hbm_bytes"), "ms_under_profiler": v.get("ms_under_profiler")}
                                                    for k, v in j["per_kernel"].items()}} if j.get("per_kernel") else {}),
                                 # what the kernel waits for, from the same passes (SQ_* / TCC_* counters)
                                 "counters": {k: j.get(k) for k in ("wait_any_over_wave_cycles", "valu_active_over_wave_cycles",
                                                                     "valu_issue_frac_of_peak", "l2_hit_rate",
                                                                     "kernel_ms_under_profiler") if j.get(k) is not None},
                                 "source": "profiles/%s_pmc_%s.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, "
                                           "separate passes; bytes = 1024*(2*FETCH_SIZE+WRITE_SIZE))" % (tag, w)}
    print("merged", w)
json.dump(table, open(table_path, "w"), indent=1)
