#!/bin/bash
# A/B of two builds of the HIP library in ONE session (same box, interleaved runs):
#   scripts/gpu_ab.sh build/dbg/lib_prev.so [extra bench.py args]
# prints QPs/s and ms per step of `prev` (the given library) and `new` (the in-tree one), 3 rounds
A=$1; shift
for i in 1 2 3; do
  for v in prev new; do
    if [ $v = prev ]; then export PQP_HIP_LIBRARY=$A; else unset PQP_HIP_LIBRARY; fi
    timeout 120 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --mpc-steps 0 "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$v', round(j['value']), '%.3f ms' % j['ms_per_step'], 'index-order %.0f' % j['dispatch']['value_index_order'], 'unsolved', j['unsolved'])"
  done
done
