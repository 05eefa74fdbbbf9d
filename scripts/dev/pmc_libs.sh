#!/bin/bash
# PMC counters of every pqp_* kernel for several builds of the library on C2 (wave kernel forced).
#   scripts/dev/pmc_libs.sh out.txt lib1.so lib2.so ...
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd $(dirname $0)/../.. && pwd)
OUT=$R/gpurun_out/$1; shift
: > $OUT
cd /tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  rm -rf /tmp/pmc_$tag; mkdir -p /tmp/pmc_$tag
  for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
    p=$(echo $pass | cut -d' ' -f1)
    PQP_HIP_LIBRARY=$R/$lib PQP_DENSE_KERNEL=wave PQP_AB_MODES=wave timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmc_$tag/$p -- python $R/scripts/gpu_dwave_ab.py 2048 3 > /tmp/pmc_$tag/$p.log 2>&1
  done
  python - $tag >> $OUT <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmc_%s/**/*counter_collection.csv' % tag, recursive=True):
    per = collections.defaultdict(float)
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0]
        per[(k, row['Counter_Name'], row['Dispatch_Id'])] += float(row['Counter_Value'])
    for (k, c, _), v in per.items():
        acc[k][c].append(v)
for k, cs in acc.items():
    if 'pqp_' not in k or 'setup' in k:
        continue
    m = {c: sorted(v)[len(v) // 2] for c, v in cs.items()}
    print("== %s  %s" % (tag, k))
    print("   " + "  ".join("%s=%.4g" % (c, v) for c, v in sorted(m.items())))
    if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
        print("   HBM bytes per launch = 1024 (2 FETCH + WRITE) = %.3f GB (read %.3f, write %.3f)" % (1024 * (2 * m['FETCH_SIZE'] + m['WRITE_SIZE']) * 1e-9, 2048 * m['FETCH_SIZE'] * 1e-9, 1024 * m['WRITE_SIZE'] * 1e-9))
    if 'SQ_INSTS_VALU' in m:
        print("   per QP: VALU %.0f SALU %.0f" % (m['SQ_INSTS_VALU'] / 2048, m.get('SQ_INSTS_SALU', 0) / 2048))
    if 'SQ_INSTS_MFMA' in m:
        print("   per QP: MFMA %.0f LDS %.0f VMEM_RD %.0f VMEM_WR %.0f" % (m['SQ_INSTS_MFMA'] / 2048, m['SQ_INSTS_LDS'] / 2048, m['SQ_INSTS_VMEM_RD'] / 2048, m['SQ_INSTS_VMEM_WR'] / 2048))
    if 'SQ_WAIT_INST_ANY' in m:
        print("   waves waiting %.2f of their cycles, executing %.2f" % (m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES'], m['SQ_ACTIVE_INST_ANY'] / m['SQ_WAVE_CYCLES']))
PY
done
cat $OUT
