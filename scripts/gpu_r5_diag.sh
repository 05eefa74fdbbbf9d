#!/bin/bash
# the one-wavefront diagonal kernel: time, phase profile (instrumented build), parity against the oracle on C5 / C5box
export TMPDIR=/tmp
mkdir -p gpurun_out/r5b
O=gpurun_out/r5b
python scripts/gpu_box_probe.py c2 c5 c5box 2>&1 | grep BOX
timeout 200 python bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline --stats 2>&1 | grep -v amdgpu.ids | cut -c1-300 | grep "cyc_\|n_newton\|n_ls\|n_kkt\|cycles total\|value" | grep -v " 0    0.0%"
for bx in 0 1; do timeout 300 python scripts/gpu_c5_check.py 200 ${NCHK:-512} $bx 2>&1 | tail -2; done | tee $O/check.txt
