#!/usr/bin/env python
"""Thread-count sweep of the CPU baseline (oracle under the reference's OpenMP driver) on the
workload of bench.py, to check that the `cpu_baseline` figure bench.py reports is the best the
host can do and not an artefact of thread placement or of the allocator.

    python scripts/cpu_baseline_sweep.py [--workload c2] [--sample 2048] [--threads 1,8,32,64,128,256]

Each allocator variant runs in its own process (GLIBC_TUNABLES is read at start-up).  No GPU needed.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = {
    "default": "",
    "heap": "glibc.malloc.mmap_threshold=1073741824:glibc.malloc.trim_threshold=1073741824:"
            "glibc.malloc.top_pad=67108864",
}


def child(args):
    import bench
    count, n, ne, ni, kind = bench.WORKLOADS[args.workload]
    sample = min(args.sample, count)
    w = bench.Workload(kind, sample, n, ne, ni)
    O, qps = bench._oracle_qps(w, sample, 1e-9)
    cores = os.cpu_count() or 1
    O.solve_in_parallel(qps, cores)
    out = {}
    for nt in [int(t) for t in args.threads.split(",")]:
        if nt > cores:
            continue
        sub = qps if nt >= 8 else qps[:max(64, sample // 16)]
        O.solve_in_parallel(sub, nt)
        reps, t0 = 0, time.perf_counter()
        while True:
            O.solve_in_parallel(sub, nt)
            reps += 1
            dt = time.perf_counter() - t0
            if dt > args.budget or reps >= 8:
                break
        out[nt] = len(sub) * reps / dt
        print("  %4d threads: %9.0f QPs/s  (%d QPs x %d passes, %.2f s)" % (nt, out[nt], len(sub), reps, dt),
              flush=True)
    print("JSON " + json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--sample", type=int, default=2048)
    ap.add_argument("--threads", default="1,8,32,64,128,192,256")
    ap.add_argument("--budget", type=float, default=4.0)
    ap.add_argument("--variants", default="default,heap")
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.child:
        return child(args)
    result = {}
    for v in args.variants.split(","):
        env = dict(os.environ)
        if VARIANTS[v]:
            env["GLIBC_TUNABLES"] = VARIANTS[v]
        print("== allocator variant %s  (GLIBC_TUNABLES=%s)" % (v, VARIANTS[v] or "-"), flush=True)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--workload", args.workload,
                            "--sample", str(args.sample), "--threads", args.threads, "--budget", str(args.budget)],
                           env=env, stdout=subprocess.PIPE, text=True)
        for line in p.stdout.splitlines():
            if line.startswith("JSON "):
                result[v] = json.loads(line[5:])
            else:
                print(line)
    print(json.dumps({"workload": args.workload, "logical_cpus": os.cpu_count(), "QPs_per_s": result}))


if __name__ == "__main__":
    main()
