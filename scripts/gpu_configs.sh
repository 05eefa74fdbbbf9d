#!/bin/bash
# bench.py on the other BASELINE.json configurations (parity check inside: max_kkt, unsolved)
export TMPDIR=/tmp
mkdir -p gpurun_out
for w in ${1:-c1 c5 c5box c4}; do
  echo "== $w"
  timeout ${2:-600} python bench.py --workload $w --steps 3 --warmup 1 > gpurun_out/bench_$w.log 2>&1; echo "rc=$?"
  grep -v amdgpu gpurun_out/bench_$w.log | tail -3 | cut -c1-1500
done
