#!/bin/bash
# HBM traffic per QP (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes) of library builds on C2:
#   B=2048 scripts/gpu_traffic_libs.sh out_tag lib1.so lib2.so ...
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
B=${B:-2048}
cd /tmp
for lib in "$@"; do
  t=$(basename $lib .so)
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/${t}_$ctr
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/${t}_$ctr -- python $R/scripts/gpu_phase_traffic.py $R/$lib $B 3 > $O/${t}_$ctr.log 2>&1
  done
done
cd $R
python - $O $B "$@" <<'PY' | tee $O/summary.txt
import csv, glob, sys, collections, os
O, B = sys.argv[1], int(sys.argv[2])
for lib in sys.argv[3:]:
    t = os.path.basename(lib)[:-3]
    v = {}
    dur = []
    for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
        per = collections.defaultdict(float)
        for f in glob.glob('%s/%s_%s/**/*counter_collection.csv' % (O, t, ctr), recursive=True):
            for row in csv.DictReader(open(f)):
                if 'pqp_solve' in row.get('Kernel_Name', '') and row['Counter_Name'] == ctr:
                    per[int(row['Dispatch_Id'])] += float(row['Counter_Value'])
        v[ctr] = [x for _, x in sorted(per.items())]
        for f in glob.glob('%s/%s_%s/**/*kernel_trace.csv' % (O, t, ctr), recursive=True):
            for row in csv.DictReader(open(f)):
                if 'pqp_solve' in row.get('Kernel_Name', ''):
                    dur.append((float(row['End_Timestamp']) - float(row['Start_Timestamp'])) * 1e-6)
    if not v['FETCH_SIZE'] or not v['WRITE_SIZE']:
        print(t, 'no counters'); continue
    rd, wr = 2048.0 * v['FETCH_SIZE'][-1] / B, 1024.0 * v['WRITE_SIZE'][-1] / B
    print('%-22s read %.2f MB/QP  write %.2f MB/QP  total %.2f MB/QP   kernel ms (profiled) %s' % (t, rd / 1e6, wr / 1e6, (rd + wr) / 1e6, ' '.join('%.2f' % d for d in dur)))
PY
find $O -name "*.db" -delete 2>/dev/null
