#!/bin/bash
# round 3: A/B of the two-kernel solve (PQP_SPLIT_SOLVE=1: prepare + iterate) against the fused kernel, C2,
# prepare kernel at 4 / 3 / 2 workgroups per CU (128 / 168 / 213 VGPRs)
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { # label, lib, split
  if [ "$2" = new ]; then unset PQP_HIP_LIBRARY; else export PQP_HIP_LIBRARY=$2; fi
  PQP_SPLIT_SOLVE=$3 timeout 300 python bench.py --workload c2 --steps 8 --warmup 2 --no-cpu-baseline --mpc-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$1', round(j['value']), '%.3f ms' % j['ms_per_step'], 'lpt %.0f' % j['dispatch']['value_lpt_repeated_batch'], 'unsolved', j['unsolved'], 'kkt %.2e' % j['max_kkt_residual'], 'dx %.1e' % 0)"
}
for i in 1 2 3; do
  run fused new 0
  run split_prep4 new 1
  run split_prep3 build/dbg/lib_split_prep3.so 1
  run split_prep2 build/dbg/lib_split_prep2.so 1
done 2>&1 | tee gpurun_out/r03_ab_split.txt
PQP_SPLIT_SOLVE=1 timeout 600 python -m pytest tests -m gpu -q -x -k "full_size_c2 or launch_size or determinism or state_machine or random_batch" 2>&1 | tail -3
