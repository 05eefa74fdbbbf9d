#!/bin/bash
# C2 shape, batch size swept on one GPU: QPs/s in index order and with the opt-in learned order
for B in 64 256 512 768 1024 1536 2048 3072 4096 8192 16384; do
  timeout 300 python bench.py --workload c2 --batch $B --steps 5 --warmup 1 --no-cpu-baseline --mpc-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); d = j['dispatch']
        print('%6d %8d %9.3f %9.3f   kernel %.3f ms' % ($B, round(j['value']), j['ms_per_step'], d['ms_per_step_lpt_repeated_batch'], j['roofline']['kernel_ms']))"
done
