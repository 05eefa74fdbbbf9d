#!/bin/bash
# first GPU pass: smoke -> parity tests -> bench at three register budgets -> rocprof
export TMPDIR=/tmp
mkdir -p gpurun_out
(nproc; lscpu | grep -E "Model name|Socket|Core|Thread|^CPU\(s\)"; rocm-smi --showproductname 2>/dev/null | head -20) > gpurun_out/box.txt 2>&1
echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
echo "== bench quick (wps=4)"; PQP_WAVES_PER_SIMD=4 timeout 300 python bench.py --steps 3 --warmup 1 --stats --no-cpu-baseline > gpurun_out/bench_wps4.log 2>&1; echo "rc=$?"; tail -25 gpurun_out/bench_wps4.log
echo "== bench quick (wps=2)"; PQP_WAVES_PER_SIMD=2 timeout 300 python bench.py --steps 3 --warmup 1 --stats --no-cpu-baseline > gpurun_out/bench_wps2.log 2>&1; echo "rc=$?"; tail -25 gpurun_out/bench_wps2.log
echo "== bench quick (wps=1)"; PQP_WAVES_PER_SIMD=1 timeout 300 python bench.py --steps 3 --warmup 1 --stats --no-cpu-baseline > gpurun_out/bench_wps1.log 2>&1; echo "rc=$?"; tail -25 gpurun_out/bench_wps1.log
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
echo "== bench full"; timeout 600 python bench.py --steps 10 --warmup 2 --stats > gpurun_out/bench.log 2>&1; echo "rc=$?"; tail -30 gpurun_out/bench.log
echo "== rocprof"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; echo "rocprof rc=$?"; cd $GRAFT_REPO_ROOT; find gpurun_out/prof -name "*stats*" | head; for f in $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); do head -5 $f; done
