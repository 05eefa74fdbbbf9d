#!/bin/bash
# kernel-trace + PMC summary of the one-wavefront dense kernel and its prologue on C2 (2048 QPs), one MI355X.
#   scripts/gpu_dwave_prof.sh [tag]   -> gpurun_out/dwave_prof_<tag>.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(cd $(dirname $0)/.. && pwd)
TAG=${1:-x}
OUT=$R/gpurun_out/dwave_prof_$TAG.txt
mkdir -p $R/gpurun_out
cd /tmp
: > $OUT
rm -rf $R/gpurun_out/dwp_*
[ -z "$DWP_ONLY" ] && PQP_AB_MODES=wave timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dwp_trace -- python $R/scripts/gpu_dwave_ab.py 2048 4 > $R/gpurun_out/dwp_trace.log 2>&1
tail -4 $R/gpurun_out/dwp_trace.log >> $OUT
for f in $(find $R/gpurun_out/dwp_trace -name '*kernel_stats.csv'); do echo "== $f" >> $OUT; head -8 $f >> $OUT; done
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_ANY SQ_WAVES"; do
  p=$(echo $pass | cut -d' ' -f1)
  [ -n "$DWP_ONLY" ] && case " $DWP_ONLY " in *" $p "*) ;; *) continue ;; esac
  PQP_AB_MODES=wave timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/dwp_$p -- python $R/scripts/gpu_dwave_ab.py 2048 3 > $R/gpurun_out/dwp_$p.log 2>&1
done
cd $R
python - >> $OUT <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/dwp_*/**/*counter_collection.csv', recursive=True):
    per = collections.defaultdict(float)
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0]
        per[(k, row['Counter_Name'], row['Dispatch_Id'])] += float(row['Counter_Value'])
    for (k, c, _), v in per.items():
        acc[k][c].append(v)
for k, cs in acc.items():
    if 'pqp_' not in k:
        continue
    print("==", k)
    for c, v in sorted(cs.items()):
        v = sorted(v)
        print("   %-28s median %.4g  (n=%d)" % (c, v[len(v) // 2], len(v)))
    m = {c: sorted(v)[len(v) // 2] for c, v in cs.items()}
    if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
        print("   HBM bytes per launch = 1024 (2 FETCH + WRITE) = %.3f GB" % (1024 * (2 * m['FETCH_SIZE'] + m['WRITE_SIZE']) * 1e-9))
    if 'SQ_INSTS_VALU' in m:
        print("   VALU instructions per QP (2048): %.0f ; SALU %.0f" % (m['SQ_INSTS_VALU'] / 2048, m.get('SQ_INSTS_SALU', 0) / 2048))
    if 'SQ_WAIT_INST_ANY' in m and 'SQ_WAVE_CYCLES' in m:
        print("   waves waiting %.2f of their cycles; VALU active %.2f" % (m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES'], m.get('SQ_ACTIVE_INST_VALU', 0) / m['SQ_WAVE_CYCLES']))
PY
cat $OUT
