#!/bin/bash
# round 4, first GPU session: parity of the changed kernels, A/B of the byte-saving variants at B = 2048 and 8192,
# wider workgroups per QP at small launches, and the per-phase HBM traffic attribution (phase repeated under PMC).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r04_call1
mkdir -p $O
echo "== parity subset" | tee $O/summary.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "full_size_c2 or launch_size or schur_factor or primal_ldlt or state_machine" 2>&1 | tail -5 | tee -a $O/summary.txt
one() { # lib batch extra-env...
  local lib=$1 batch=$2; shift 2
  env "$@" PQP_HIP_LIBRARY=$lib timeout 300 python bench.py --workload c2 --batch $batch --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline --mpc-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$(basename $lib)', '$batch', '$*', round(j['value']), '%.3f ms' % j['ms_per_step'], 'kernel %.3f' % j['roofline']['kernel_ms'], 'first %.3f' % j['first_solve_ms'], 'lpt %.0f' % j['dispatch']['value_lpt_repeated_batch'], 'unsolved', j['unsolved'], 'kkt %.2e' % j['max_kkt_residual'])"
}
echo "== A/B at B = 2048 (3 interleaved rounds)" | tee -a $O/summary.txt
for i in 1 2 3; do
  for v in build/dbg/lib_r4base.so proxsuite_amd/csrc/libproxqp_hip.so build/dbg/lib_notouch.so; do
    one $v 2048 A=1 | tee -a $O/summary.txt
  done
done
echo "== A/B at B = 8192" | tee -a $O/summary.txt
for v in build/dbg/lib_r4base.so proxsuite_amd/csrc/libproxqp_hip.so build/dbg/lib_notouch.so; do
  STEPS=4 one $v 8192 A=1 | tee -a $O/summary.txt
done
echo "== wider workgroups per QP at small launches (PQP_FORCE_NT)" | tee -a $O/summary.txt
for bsz in 64 256; do
  one proxsuite_amd/csrc/libproxqp_hip.so $bsz A=1 | tee -a $O/summary.txt
  one proxsuite_amd/csrc/libproxqp_hip.so $bsz PQP_FORCE_NT=512 | tee -a $O/summary.txt
  one proxsuite_amd/csrc/libproxqp_hip.so $bsz PQP_FORCE_NT=1024 | tee -a $O/summary.txt
done
echo "== per-phase traffic (instrumented library, phase repeated twice; FETCH_SIZE / WRITE_SIZE in separate passes)" | tee -a $O/summary.txt
cd /tmp
for ph in 0 1 2 3 4 5 6; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    if [ $ctr = WRITE_SIZE ] && [ $ph != 0 ] && [ $ph != 4 ] && [ $ph != 6 ]; then continue; fi
    rm -rf $O/pmc_${ph}_$ctr
    PQP_REPEAT_PHASE=$ph PQP_REPEAT_COUNT=2 timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_${ph}_$ctr -- python $R/scripts/gpu_phase_traffic.py $R/proxsuite_amd/csrc/libproxqp_hip_stats.so 2048 3 > $O/pmc_${ph}_$ctr.log 2>&1
    grep '^{' $O/pmc_${ph}_$ctr.log | tail -1 > $O/phase_${ph}_$ctr.json
  done
done
cd $R
python scripts/phase_traffic_summary.py $O | tee -a $O/summary.txt
# keep the merge small: drop the raw profiler trees except the counter CSVs
find $O -name "*.db" -delete 2>/dev/null
du -sh $O | tee -a $O/summary.txt
