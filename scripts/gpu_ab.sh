#!/bin/bash
# A/B of builds of the HIP library in ONE session (same box, interleaved runs):
#   scripts/gpu_ab.sh build/dbg/lib_prev.so [more .so ...]
# prints QPs/s (index order = `value`) and ms per step of each given library and of `new` (the
# in-tree one), 3 rounds.  WORKLOAD=c4 etc. selects another BASELINE.json configuration.
W=${WORKLOAD:-c2}
for i in 1 2 3; do
  for v in "$@" new; do
    if [ $v = new ]; then unset PQP_HIP_LIBRARY; else export PQP_HIP_LIBRARY=$v; fi
    timeout 300 python bench.py --workload $W --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline --mpc-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$(basename $v)', '$W', round(j['value']), '%.3f ms' % j['ms_per_step'], 'lpt %.0f' % j['dispatch']['value_lpt_repeated_batch'], 'unsolved', j['unsolved'], 'kkt %.2e' % j['max_kkt_residual'])"
  done
done
