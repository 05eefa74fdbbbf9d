#!/bin/bash
# The C5 part of scripts/gpu_final.sh alone (after a change to pqp_diag.hpp): bench lines, kernel-trace summary, PMC passes.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(cd $(dirname $0)/.. && pwd)
TAG=${1:-r05}
mkdir -p $R/gpurun_out/final
cd $R
for w in c5 c5box; do
  timeout 300 python bench.py --workload $w --steps 5 --warmup 1 --stats > gpurun_out/final/${TAG}_bench_$w.log 2>&1
  tail -1 gpurun_out/final/${TAG}_bench_$w.log > gpurun_out/final/${TAG}_bench_$w.json
done
cd /tmp
rm -rf $R/gpurun_out/final/trace_c5
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/trace_c5 -- python $R/bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline --mpc-steps 0 > $R/gpurun_out/final/${TAG}_trace_bench_c5.log 2>&1
cd $R
f=$(find gpurun_out/final/trace_c5 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f gpurun_out/final/${TAG}_kernel_stats_c5.csv
scripts/gpu_pmc_traffic.sh c5 c5box 2>&1 | tail -3
for w in c5 c5box; do python - $w $TAG <<'PY'
import json, sys
w, tag = sys.argv[1:3]
j = json.loads(open('gpurun_out/final/%s_bench_%s.json' % (tag, w)).read())
r = j['roofline']
print(w, round(j['value']), '%.3f ms' % j['ms_per_step'], 'kernel %.3f' % r['kernel_ms'], 'unsolved', j['unsolved'], 'kkt %.2e' % j['max_kkt_residual'], 'cpu', round(j['cpu_baseline']['value']), j['max_abs_delta_vs_cpu'])
PY
done
