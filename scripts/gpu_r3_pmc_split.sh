#!/bin/bash
# HBM traffic of the C2 solve by PHASE: the two-kernel solve (PQP_SPLIT_SOLVE=1) under rocprofv3 --pmc
# FETCH_SIZE / WRITE_SIZE (separate passes, bytes = 1024 * (2 * FETCH + WRITE)), per kernel name.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/gpurun_out
cd /tmp
for split in 1 0; do
for pass in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum"; do
  p=$(echo $pass | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmcs${split}_$p
  PQP_SPLIT_SOLVE=$split timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmcs${split}_$p -- python $R/bench.py --workload c2 --steps 2 --warmup 1 --no-cpu-baseline --mpc-steps 0 > $R/gpurun_out/pmcs${split}_$p.log 2>&1
done
done
cd $R
python - <<'PY'
import csv, glob, json, collections
out = {}
for split in (1, 0):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('gpurun_out/pmcs%d_*/**/*counter_collection.csv' % split, recursive=True):
        per = collections.defaultdict(float)
        for row in csv.DictReader(open(f)):
            k = row.get('Kernel_Name', '')
            if 'pqp_solve' not in k:
                continue
            name = 'prepare' if ', 1>(' in k and 'part' in k else ('iterate' if 'part' in k else 'fused')
            per[(name, row['Counter_Name'], row['Dispatch_Id'])] += float(row['Counter_Value'])
        for (name, c, _), v in per.items():
            acc[name][c].append(v)
    for name, cs in acc.items():
        mean = {c: sum(v) / len(v) for c, v in cs.items()}
        rec = dict(mean)
        if 'FETCH_SIZE' in mean and 'WRITE_SIZE' in mean:
            rec['hbm_bytes_per_launch'] = 1024.0 * (2 * mean['FETCH_SIZE'] + mean['WRITE_SIZE'])
            rec['read_bytes'] = 2048.0 * mean['FETCH_SIZE']
            rec['write_bytes'] = 1024.0 * mean['WRITE_SIZE']
        out['split%d_%s' % (split, name)] = rec
json.dump(out, open('gpurun_out/r03_pmc_c2_by_phase.json', 'w'), indent=1)
for k, v in out.items():
    print(k, {a: ('%.3g' % b) for a, b in v.items()})
PY
