#!/bin/bash
# PMC passes (counters only with --kernel-trace, each in its own run) + a --stats trace
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --mpc-steps 0 > $R/gpurun_out/pmc_$tag.log 2>&1
  echo "pass $tag rc=$?"
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --mpc-steps 0 > $R/gpurun_out/trace.log 2>&1
echo "trace rc=$?"
cd $R; find gpurun_out -name "*.csv" | head -30
