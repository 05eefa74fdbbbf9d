export TMPDIR=/tmp
mkdir -p gpurun_out/r5e
python scripts/gpu_box_probe.py c2 c5 c5box c1 2>&1 | grep BOX
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -6
for w in c5 c5box; do timeout 300 python bench.py --workload $w --steps 10 --warmup 2 --stats > gpurun_out/r5e/bench_$w.log 2>&1; tail -1 gpurun_out/r5e/bench_$w.log > gpurun_out/r5e/bench_$w.json; done
python - <<'PY'
import json
for w in ("c5","c5box"):
    d=json.loads(open("gpurun_out/r5e/bench_%s.json"%w).read()); print(w, round(d["value"]), "norm", round(d["value_normalised"]), "ms %.3f kernel %.3f"%(d["ms_per_step"], d["roofline"]["kernel_ms"]), "unsolved", d["unsolved"], "delta", d["max_abs_delta_vs_cpu"])
PY
