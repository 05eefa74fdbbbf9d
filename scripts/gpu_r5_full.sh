#!/bin/bash
# full GPU suite + the bench lines of every BASELINE configuration (round 5)
export TMPDIR=/tmp
mkdir -p gpurun_out/r5c
O=gpurun_out/r5c
python scripts/gpu_box_probe.py c2 c5 c4 c1 2>&1 | grep BOX
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -12 | tee $O/gputest_tail.txt
timeout 400 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?"
for w in c1 c5 c5box; do timeout 300 python bench.py --workload $w --steps 10 --warmup 2 > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$?"; done
timeout 400 python bench.py --gpus 4 --inprocess --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_inprocess4.json 2> $O/bench_inprocess4.err; echo "inproc rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5c/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "value %.0f norm %s ms %.3f kernel %.3f unsolved %s kkt %.2e frac %s box_slowdown %s cpu %s init_ms %s" % (d["value"], ("%.0f" % d["value_normalised"]) if "value_normalised" in d else None, d["ms_per_step"], d.get("roofline",{}).get("kernel_ms",0), d.get("unsolved"), d.get("max_kkt_residual",0), d.get("roofline",{}).get("frac"), d.get("box",{}).get("slowdown_vs_reference_box"), (d.get("cpu_baseline") or {}).get("value"), d.get("init_ms")))
    except Exception as e:
        print(f, "ERR", e)
PY
