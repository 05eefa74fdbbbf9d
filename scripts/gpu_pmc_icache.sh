#!/bin/bash
# instruction-cache behaviour of the solve kernels (every solve kernel is one fully inlined ~350 KB function; the instruction
# cache of a CU pair holds 64 KB): SQC_ICACHE_* and the average instruction-fetch latency, per BASELINE configuration
#   scripts/gpu_pmc_icache.sh [c1 c2 c5 ...]   -> gpurun_out/pmc_icache_<w>.json
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/gpurun_out
cd /tmp
WL="$@"; [ -z "$WL" ] && WL="c1 c2 c5"
for w in $WL; do
  for pass in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"; do
    p=$(echo $pass | cut -d' ' -f1)
    rm -rf $R/gpurun_out/pmci_${w}_$p
    timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmci_${w}_$p -- python $R/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --mpc-steps 0 > $R/gpurun_out/pmci_${w}_$p.log 2>&1
  done
done
cd $R
python - $WL <<'PY'
import csv, glob, json, sys, collections
for w in sys.argv[1:]:
    acc = collections.defaultdict(list)
    for f in glob.glob('gpurun_out/pmci_%s_*/**/*counter_collection.csv' % w, recursive=True):
        per = collections.defaultdict(float)
        for row in csv.DictReader(open(f)):
            k = row.get('Kernel_Name', '')
            if 'pqp_solve_kernel' not in k and 'pqp_diag_kernel' not in k:
                continue
            per[(row['Counter_Name'], row['Dispatch_Id'])] += float(row['Counter_Value'])
        for (c, _), v in per.items():
            acc[c].append(v)
    med = lambda v: sorted(v)[len(v) // 2]
    m = {c: med(v) for c, v in acc.items() if v}
    out = {"workload": w, "per_launch": m}
    if m.get('SQC_ICACHE_REQ'):
        out["icache_miss_rate"] = m.get('SQC_ICACHE_MISSES', 0) / m['SQC_ICACHE_REQ']
    if m.get('SQ_IFETCH'):
        out["ifetch_level_over_ifetch"] = m.get('SQ_IFETCH_LEVEL', 0) / m['SQ_IFETCH']
    if m.get('SQ_WAVE_CYCLES'):
        out["wait_inst_any_over_wave_cycles"] = m.get('SQ_WAIT_INST_ANY', 0) / m['SQ_WAVE_CYCLES']
        out["insts_per_wave_cycle"] = (m.get('SQ_INSTS_VALU', 0) + m.get('SQ_INSTS_SALU', 0)) / m['SQ_WAVE_CYCLES']
    json.dump(out, open('gpurun_out/pmc_icache_%s.json' % w, 'w'), indent=1)
    print(w, json.dumps({k: v for k, v in out.items() if k != "per_launch"}), json.dumps(m))
PY
