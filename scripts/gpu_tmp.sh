export TMPDIR=/tmp
python scripts/gpu_box_probe.py c2 c5 c5box 2>&1 | grep BOX
for bx in 0 1; do timeout 300 python scripts/gpu_c5_check.py 200 1024 $bx 2>&1 | tail -1; done
timeout 300 python scripts/diag_kernel_sweep.py 2 60 2>&1 | grep "MISMATCH\|sweep seed" | cut -c1-400
