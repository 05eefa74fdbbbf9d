#!/bin/bash
# round 3: tests that changed + A/B of the in-tree library against build/dbg/lib_r3base.so on C2 / C5 / C5box
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "maros_meszaros_medium or vectors_in_hbm or full_shape_c5 or full_size_c2 or closest" 2>&1 | tail -15 > gpurun_out/r03_gputest_3.log
tail -3 gpurun_out/r03_gputest_3.log
for w in c2 c5 c5box; do
  WORKLOAD=$w STEPS=6 bash scripts/gpu_ab.sh build/dbg/lib_r3base.so 2>&1 | tee -a gpurun_out/r03_ab_ls.txt
done
