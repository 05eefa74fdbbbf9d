#!/bin/bash
# quick GPU iteration: sanity check (seconds; aborts the rest when the kernel is broken), parity
# tests, bench with phase stats at the register budgets given in $1 (default "3")
export TMPDIR=/tmp
mkdir -p gpurun_out
WPS_LIST=${1:-3}
echo "== quickcheck"; timeout 90 python scripts/gpu_quickcheck.py 100 2>&1 | grep -v amdgpu | tee gpurun_out/quickcheck.log
python - <<PY || { echo "quickcheck FAILED: skipping the rest"; exit 1; }
import re, sys
m = re.findall(r"solved (\d+)/(\d+)", open("gpurun_out/quickcheck.log").read())
sys.exit(0 if len(m) == 4 and all(a == b for a, b in m) else 1)
PY
echo "== pytest gpu"; timeout 300 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_gpu.log
for w in $WPS_LIST; do
  echo "== bench wps=$w"; PQP_WAVES_PER_SIMD=$w timeout 120 python bench.py --steps 5 --warmup 1 --stats --no-cpu-baseline > gpurun_out/bench_wps$w.log 2>&1; echo "rc=$?"
  grep -v amdgpu.ids gpurun_out/bench_wps$w.log | head -19
  python - <<PY
import json
l=[x for x in open("gpurun_out/bench_wps$w.log") if x.startswith("{")]
if l:
    j=json.loads(l[-1]); print("QPs/s %.0f  ms/step %.2f  kernel_ms %.2f  max_kkt %.2e unsolved %d" % (j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], j["max_kkt_residual"], j["unsolved"]))
PY
done
