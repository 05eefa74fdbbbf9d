#!/bin/bash
# What an LDS-resident H_s would cost in residency at C2: the product kernel launched with the LDS such a kernel would
# need (PQP_LDS_PAD_BYTES), interleaved, 3 rounds.  0 = product (4 workgroups per CU); 12000 -> 3; 40400 (packed
# triangle of H_s) -> 2; 80000 (full H_s) -> 1.
for i in 1 2 3; do
  for pad in 0 12000 40400 80000; do
    PQP_LDS_PAD_BYTES=$pad timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --mpc-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('pad $pad', 'lds', j['config']['lds_bytes_per_qp'], 'workgroups/CU', 163840 // j['config']['lds_bytes_per_qp'], round(j['value']), 'QPs/s', '%.3f ms' % j['ms_per_step'])"
  done
done
