#!/bin/bash
# matrix-core utilisation of the solve kernel (C2 and C4): counters only, one pass each
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\bSQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\b" | sort -u | tr '\n' ' '; echo
for w in c2 c4; do
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmcm_$w -- python $R/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmcm_$w.log 2>&1
  echo "pass $w rc=$?"
done
cd $R
python - <<'PY'
import csv,glob,collections
for w in ("c2","c4"):
    acc=collections.defaultdict(list)
    for f in glob.glob("gpurun_out/pmcm_%s/**/*counter_collection.csv"%w,recursive=True):
        for r in csv.DictReader(open(f)):
            if "pqp_solve_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(w, {k: sum(v)/len(v) for k,v in sorted(acc.items())})
PY
