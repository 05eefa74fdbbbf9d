#!/bin/bash
# A/B of library builds on a custom shape: N NE NI BATCH then the .so files to compare against the in-tree one
N=$1; NE=$2; NI=$3; B=$4; shift 4
for i in 1 2 3; do
  for v in "$@" new; do
    if [ $v = new ]; then unset PQP_HIP_LIBRARY; else export PQP_HIP_LIBRARY=$v; fi
    timeout 300 python bench.py --n $N --n-eq $NE --n-in $NI --batch $B --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline --mpc-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$(basename $v)', '$N/$NE/$NI x$B', j['config'].get('threads_per_qp'), round(j['value']), '%.3f ms' % j['ms_per_step'], 'unsolved', j['unsolved'], 'kkt %.2e' % j['max_kkt_residual'])"
  done
done
