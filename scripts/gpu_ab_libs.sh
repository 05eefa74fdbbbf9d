#!/bin/bash
# interleaved A/B of library builds on the C2 workload at several batch sizes (one box):
#   ROUNDS=3 BATCHES="2048 8192" scripts/gpu_ab_libs.sh out_tag lib1.so lib2.so ...
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
TAG=$1; shift
O=$R/gpurun_out/$TAG.txt
mkdir -p $R/gpurun_out
: > $O
one() {
  local lib=$1 batch=$2
  PQP_HIP_LIBRARY=$lib timeout 300 python bench.py --workload ${WORKLOAD:-c2} --batch $batch --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline --mpc-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$(basename $lib)', '$batch', round(j['value']), '%.3f ms' % j['ms_per_step'], 'kernel %.3f' % j['roofline']['kernel_ms'], 'first %.3f' % j['first_solve_ms'], 'lpt %.0f' % j['dispatch']['value_lpt_repeated_batch'], 'unsolved', j['unsolved'], 'kkt %.2e' % j['max_kkt_residual'], 'dxyz %.1e' % max(j['max_abs_delta_vs_cpu'][k] for k in 'xyz') if j.get('max_abs_delta_vs_cpu') else '')"
}
for b in ${BATCHES:-2048}; do
  for i in $(seq 1 ${ROUNDS:-3}); do
    for v in "$@"; do
      one $v $b | tee -a $O
    done
  done
done
