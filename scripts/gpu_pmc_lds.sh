#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\bSQ_[A-Z_0-9]*LDS[A-Z_0-9]*\b|\bSQ_INST_CYCLES_[A-Z]*\b|\bSQ_ACTIVE_INST_[A-Z]*\b|\bSQ_THREAD_CYCLES_VALU\b|\bSQ_VALU_MFMA_BUSY_CYCLES\b" | sort -u > $R/gpurun_out/lds_counters.txt
cat $R/gpurun_out/lds_counters.txt | tr '\n' ' '; echo
for pass in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmcl_$tag -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmcl_$tag.log 2>&1
  echo "pass $tag rc=$?"; tail -2 $R/gpurun_out/pmcl_$tag.log | cut -c1-300
done
cd $R
python - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmcl_*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "pqp_solve_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print("%-26s %.4g"%(k,sum(v)/len(v)))
PY
