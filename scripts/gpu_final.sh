#!/bin/bash
# End-of-round measurement set (run on the GPU box): bench lines of every BASELINE config, the
# rocprofv3 kernel-trace --stats summary of the default bench command, PMC traffic per config.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(cd $(dirname $0)/.. && pwd)
TAG=${1:-r02}
mkdir -p $R/gpurun_out/final
cd $R
python bench.py --steps 10 --warmup 2 --stats > gpurun_out/final/${TAG}_bench_c2.log 2>&1
tail -1 gpurun_out/final/${TAG}_bench_c2.log > gpurun_out/final/${TAG}_bench_c2.json
for w in c1 c4 c5 c5box; do
  python bench.py --workload $w --steps 5 --warmup 1 --stats > gpurun_out/final/${TAG}_bench_$w.log 2>&1
  tail -1 gpurun_out/final/${TAG}_bench_$w.log > gpurun_out/final/${TAG}_bench_$w.json
done
python bench.py --workload backend --steps 5 --warmup 1 --mpc-steps 0 --cpu-sample 256 > gpurun_out/final/${TAG}_bench_backend_primal_ldlt.log 2>&1
PQP_BENCH_ONE_DEVICE=1 PQP_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --mpc-steps 0 > gpurun_out/final/${TAG}_bench_2ranks_one_gpu.log 2>&1
# RCCL with one rank (the "nccl" process group on one GPU): pack kernel + all_gather_into_tensor on ROCm tensors
MASTER_ADDR=127.0.0.1 MASTER_PORT=29547 python bench.py --gpus 1 --rccl-single-rank --steps 3 --warmup 1 --no-cpu-baseline --mpc-steps 0 > gpurun_out/final/${TAG}_bench_rccl_single_rank.log 2>&1
# BASELINE.json configs[2] at N = 1 (the 16 384 QPs of the 8-GPU configuration in one launch of one handle), every QP against the CPU path
python bench.py --gpus 1 --total-batch 16384 --steps 5 --warmup 1 --mpc-steps 0 --cpu-sample 16384 > gpurun_out/final/${TAG}_bench_c3_1gpu.log 2>&1
tail -1 gpurun_out/final/${TAG}_bench_c3_1gpu.log > gpurun_out/final/${TAG}_bench_c3_1gpu.json
# the in-process multi-device path (pqp_multi_*), shards mapped onto the one GPU of the box
PQP_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --inprocess --steps 5 --warmup 1 > gpurun_out/final/${TAG}_bench_inprocess_2shards_one_gpu.log 2>&1
PQP_BENCH_ONE_DEVICE=1 python bench.py --gpus 4 --inprocess --steps 5 --warmup 1 > gpurun_out/final/${TAG}_bench_inprocess_4shards_one_gpu.log 2>&1
PQP_BENCH_ONE_DEVICE=1 python bench.py --gpus 8 --inprocess --total-batch 16384 --steps 3 --warmup 1 > gpurun_out/final/${TAG}_bench_inprocess_c3_8shards_one_gpu.log 2>&1
cd /tmp
rm -rf $R/gpurun_out/final/trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/trace -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --mpc-steps 0 > $R/gpurun_out/final/${TAG}_trace_bench.log 2>&1
cd $R
f=$(find gpurun_out/final/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f gpurun_out/final/${TAG}_kernel_stats.csv
# the same for the one-wavefront diagonal kernel (the dominant kernel of the C5 bench line)
cd /tmp
rm -rf $R/gpurun_out/final/trace_c5
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/trace_c5 -- python $R/bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline --mpc-steps 0 > $R/gpurun_out/final/${TAG}_trace_bench_c5.log 2>&1
cd $R
f=$(find gpurun_out/final/trace_c5 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f gpurun_out/final/${TAG}_kernel_stats_c5.csv
[ -z "$SKIP_PMC" ] && scripts/gpu_pmc_traffic.sh c2 c1 c4 c5 c5box 2>&1 | tail -6
for w in c2 c1 c4 c5 c5box; do python - $w $TAG <<'PY'
import json, sys
w, tag = sys.argv[1:3]
j = json.loads(open('gpurun_out/final/%s_bench_%s.json' % (tag, w)).read())
r = j['roofline']
print(w, round(j['value']), '%.3f ms' % j['ms_per_step'], 'unsolved', j['unsolved'], 'kkt %.2e' % j['max_kkt_residual'],
      'engineMB %.2f' % (r.get('engine_bytes_per_qp', 0) / 1e6), 'frac %.3f' % (r.get('frac') or 0), 'cpu', round(j['cpu_baseline']['value']))
PY
done
