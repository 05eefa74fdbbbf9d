#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on a known byte count (see scripts/microbench/hbm_calib.hip)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
$R/scripts/microbench/hbm_calib > $R/gpurun_out/calib_plain.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/calib_$c -- $R/scripts/microbench/hbm_calib > $R/gpurun_out/calib_$c.log 2>&1
  echo "calib $c rc=$?"
done
cd $R
cat gpurun_out/calib_plain.log
python - <<'PY'
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/calib_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(c, k, "per-launch mean", sum(v) / len(v), "n", len(v))
PY
