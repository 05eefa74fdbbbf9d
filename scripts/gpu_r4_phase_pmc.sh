#!/bin/bash
# per-phase HBM traffic of the C2 solve: the instrumented library with one idempotent phase executed twice, under
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) -> r04_pmc_c2_waste_by_phase.json
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04_phase_pmc}
B=${2:-2048}
mkdir -p $O
cd /tmp
for ph in 0 1 2 3 4 5 6; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_${ph}_$ctr
    PQP_REPEAT_PHASE=$ph PQP_REPEAT_COUNT=2 timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_${ph}_$ctr -- python $R/scripts/gpu_phase_traffic.py $R/proxsuite_amd/csrc/libproxqp_hip_stats.so $B 3 > $O/pmc_${ph}_$ctr.log 2>&1
    grep '^{' $O/pmc_${ph}_$ctr.log | tail -1 > $O/phase_${ph}_$ctr.json
  done
done
cd $R
python scripts/phase_traffic_summary.py $O | tee $O/summary.txt
find $O -name "*.db" -delete 2>/dev/null
du -sh $O
