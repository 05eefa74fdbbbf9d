#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / L2 requests of the C2 solve kernel for the product library and an A/B partner built by
# proxsuite_amd._build.build_tu_variant into build/ab/ (here: the two-pivots-per-barrier Gauss-Jordan of
# profiles/r05_ab_gj_two_pivots.txt; the sources of that variant are not in the tree -- the record is).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for lib in proxsuite_amd/csrc/libproxqp_hip.so build/ab/libproxqp_hip_gjpairs.so; do
 for pass in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  p=$(echo $pass | cut -d' ' -f1)
  d=/tmp/pm_$(basename $lib .so)_$p
  PQP_HIP_LIBRARY=$R/$lib timeout 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $d -- python $R/bench.py --workload c2 --steps 2 --warmup 1 --no-cpu-baseline --mpc-steps 0 > $d.log 2>&1
  python - $d $lib "$pass" <<'PY'
import csv,glob,sys,collections
d,lib,p=sys.argv[1:4]
per=collections.defaultdict(float)
for f in glob.glob(d+'/**/*counter_collection.csv',recursive=True):
    for row in csv.DictReader(open(f)):
        if 'pqp_solve_kernel' in row['Kernel_Name']:
            per[(row['Counter_Name'],row['Dispatch_Id'])]+=float(row['Counter_Value'])
by=collections.defaultdict(list)
for (c,i),v in per.items(): by[c].append(v)
for c,v in by.items():
    v=sorted(v); print("PMC",lib.split('/')[-1],c,"median",v[len(v)//2],"n",len(v))
PY
 done
done
