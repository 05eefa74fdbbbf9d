#!/bin/bash
# A/B of builds of the HIP library in ONE session (same box, interleaved runs):
#   scripts/gpu_ab.sh build/dbg/lib_prev.so [more .so ...]
# prints QPs/s and ms per step of each given library and of `new` (the in-tree one), 3 rounds
for i in 1 2 3; do
  for v in "$@" new; do
    if [ $v = new ]; then unset PQP_HIP_LIBRARY; else export PQP_HIP_LIBRARY=$v; fi
    timeout 120 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --mpc-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$(basename $v)', round(j['value']), '%.3f ms' % j['ms_per_step'], 'index-order %.0f' % j['dispatch']['value_index_order'], 'unsolved', j['unsolved'])"
  done
done
