#!/bin/bash
# PMC comparison of builds of the library: scripts/gpu_pmc_ab.sh tag1=lib1.so tag2=new ...
# (counters only with --kernel-trace, each group in its own run; summary per kernel on stdout)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W=${WORKLOAD:-c2}
mkdir -p $R/gpurun_out
cd /tmp
for spec in "$@"; do
  tag=${spec%%=*}; lib=${spec#*=}
  if [ "$lib" = new ]; then unset PQP_HIP_LIBRARY; else export PQP_HIP_LIBRARY=$R/$lib; fi
  for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    p=$(echo $pass | cut -d' ' -f1)
    rm -rf $R/gpurun_out/pmcab_${tag}_$p
    timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmcab_${tag}_$p -- python $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --mpc-steps 0 > $R/gpurun_out/pmcab_${tag}_$p.log 2>&1
  done
done
cd $R
python - "$@" <<'PY'
import csv, glob, sys, collections
for spec in sys.argv[1:]:
    tag = spec.split('=')[0]
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob('gpurun_out/pmcab_%s_*/**/*counter_collection.csv' % tag, recursive=True):
        for row in csv.DictReader(open(f)):
            if 'pqp_solve_kernel' not in row.get('Kernel_Name', ''):
                continue
            k = row['Counter_Name']
            acc[k][0] += float(row['Counter_Value'])
            acc[k][1] += 1
    print('==', tag)
    # per launch = sum over the dispatch's rows / number of launches (rows per launch vary by counter dimension)
    for k in sorted(acc):
        print('  %-24s total %.6g  rows %d' % (k, acc[k][0], acc[k][1]))
PY
