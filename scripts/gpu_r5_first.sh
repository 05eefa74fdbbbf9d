#!/bin/bash
# round 5, first look: the box (clocks, HBM rate), the headline at index order, and the one-wavefront diagonal kernel
export TMPDIR=/tmp
mkdir -p gpurun_out/r5a
O=gpurun_out/r5a
(rocm-smi --showclocks --showpower --showmaxpower --showtemp --showperflevel --json; echo; rocm-smi --showclocks --showpower --showtemp) > $O/smi_before.txt 2>&1
scripts/microbench/hbm_calib > $O/hbm_calib.txt 2>&1
timeout 300 python bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline > $O/c2.json 2> $O/c2.err; echo "c2 rc=$?"
for nt in 256 64; do
  for w in c5 c5box; do
    PQP_DIAG_NT=$nt timeout 300 python bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline > $O/${w}_nt$nt.json 2> $O/${w}_nt$nt.err; echo "$w nt=$nt rc=$?"
  done
done
for bx in 0 1; do PQP_DIAG_NT=64 timeout 300 python scripts/gpu_c5_check.py 200 256 $bx; done > $O/c5_check_nt64.txt 2>&1; tail -3 $O/c5_check_nt64.txt
PQP_DIAG_NT=64 timeout 300 python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline --stats > $O/c5_nt64_stats.json 2> $O/c5_nt64_stats.err
timeout 200 python bench.py --workload c1 --steps 20 --warmup 3 --no-cpu-baseline > $O/c1.json 2> $O/c1.err; echo "c1 rc=$?"
rocm-smi --showclocks --showpower --showtemp > $O/smi_after.txt 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5a/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value %.0f ms %.3f kernel %.3f unsolved %s kkt %.2e dcpu %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d.get("unsolved"), d.get("max_kkt_residual",0), d.get("max_abs_delta_vs_cpu")))
    except Exception as e:
        print(f, "ERR", e)
PY
head -40 gpurun_out/r5a/smi_before.txt; cat gpurun_out/r5a/hbm_calib.txt
