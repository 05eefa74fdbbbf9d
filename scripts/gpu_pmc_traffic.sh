#!/bin/bash
# HBM traffic and stall summary of the solve kernel per BASELINE.json configuration, collected as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3
# --pmc passes (with --kernel-trace only), bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE) -- the gfx950
# correction calibrated on this access pattern in profiles/r01_hbm_counter_calibration.txt.
#   scripts/gpu_pmc_traffic.sh [c2 c1 c4 c5 c5box]      -> gpurun_out/pmc_traffic_<w>.json (+ raw CSVs)
# Copy the JSON files into profiles/ (scripts/merge_pmc_traffic.py) to make bench.py report them.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/gpurun_out
cd /tmp
WL="$@"; [ -z "$WL" ] && WL="c2 c1 c4 c5 c5box"
for w in $WL; do
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"; do
    p=$(echo $pass | cut -d' ' -f1)
    # PMC_QUICK=1: the traffic passes and the issue / wait pass only (three of the six)
    [ -n "$PMC_QUICK" ] && case $p in FETCH_SIZE|WRITE_SIZE|SQ_WAVE_CYCLES) ;; *) continue ;; esac
    rm -rf $R/gpurun_out/pmct_${w}_$p
    timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmct_${w}_$p -- python $R/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --mpc-steps 0 > $R/gpurun_out/pmct_${w}_$p.log 2>&1
  done
done
cd $R
python - $WL <<'PY'
import csv, glob, json, sys, collections
for w in sys.argv[1:]:
    acc = collections.defaultdict(list)          # counter -> one value per launch (summed over its dimension rows)
    dur = []
    dur_k = collections.defaultdict(list)
    names = set()
    import os
    newest = {}
    for f in glob.glob('gpurun_out/pmct_%s_*/**/*counter_collection.csv' % w, recursive=True):
        d = f.split('/')[1]
        if d not in newest or os.path.getmtime(f) > os.path.getmtime(newest[d]):
            newest[d] = f  # (gpurun_out accumulates the runs of earlier sessions: the latest one of each pass)
    SOLVE = ('pqp_solve_kernel', 'pqp_diag_kernel', 'pqp_dwave_kernel', 'pqp_prologue_kernel')
    # a solve of the one-wavefront dense path is TWO kernels (factorisation prologue + iteration): counters per kernel,
    # a launch = the sum of the per-kernel medians
    by_kernel = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in newest.values():
        per = collections.defaultdict(float)
        for row in csv.DictReader(open(f)):
            kn = row.get('Kernel_Name', '')
            if not any(k in kn for k in SOLVE):
                continue
            names.add(kn)
            per[(kn, row['Counter_Name'], row['Dispatch_Id'])] += float(row['Counter_Value'])
        for (kn, c, _), v in per.items():
            by_kernel[kn][c].append(v)
    traces = glob.glob('gpurun_out/pmct_%s_FETCH_SIZE/**/*kernel_trace.csv' % w, recursive=True)
    for f in sorted(traces, key=os.path.getmtime)[-1:]:
        for row in csv.DictReader(open(f)):
            kn = row.get('Kernel_Name', '')
            if any(k in kn for k in SOLVE):
                dur_k[kn].append((float(row['End_Timestamp']) - float(row['Start_Timestamp'])) * 1e-6)
    # per launch: the MEDIAN over the launches of the run -- the steady-state dirty re-solve of an init-ed batch that the
    # timed region of bench.py consists of (the launches that follow a fresh init rewrite the equilibrated matrices and
    # move a third more: they are the first solve, reported separately by the bench line, not the step that is priced)
    def med(v):
        v = sorted(v)
        return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])
    mean = collections.defaultdict(float)
    per_kernel = {}
    for kn, cs in by_kernel.items():
        per_kernel[kn] = {c: med(v) for c, v in cs.items() if v}
        for c, v in per_kernel[kn].items():
            mean[c] += v
        acc.update({c: v for c, v in cs.items()})
    mean = dict(mean)
    dur = [sum(sum(v) / len(v) for v in dur_k.values())] if dur_k else []
    out = {"workload": w, "kernel": sorted(names), "launches_per_counter": {kn: {c: len(v) for c, v in cs.items()} for kn, cs in by_kernel.items()},
           "per_launch": mean}
    if len(per_kernel) > 1:
        out["per_kernel"] = {kn: dict(v, **({"hbm_bytes": 1024.0 * (2.0 * v['FETCH_SIZE'] + v['WRITE_SIZE'])} if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v else {}),
                                       **({"ms_under_profiler": sum(dur_k[kn]) / len(dur_k[kn])} if dur_k.get(kn) else {}))
                             for kn, v in per_kernel.items()}
    if 'FETCH_SIZE' in mean and 'WRITE_SIZE' in mean:
        out["hbm_bytes_per_launch"] = 1024.0 * (2.0 * mean['FETCH_SIZE'] + mean['WRITE_SIZE'])
        out["formula"] = "1024 * (2 * FETCH_SIZE + WRITE_SIZE), separate --pmc passes (profiles/r01_hbm_counter_calibration.txt); per_launch = median over the launches of the run"
        if len(per_kernel) == 1:
            out["hbm_bytes_every_launch"] = [1024.0 * (2.0 * a + b) for a, b in zip(acc['FETCH_SIZE'], acc['WRITE_SIZE'])]
    if dur:
        out["kernel_ms_under_profiler"] = sum(dur) / len(dur)
    if 'SQ_WAVE_CYCLES' in mean and mean['SQ_WAVE_CYCLES']:
        out["wait_any_over_wave_cycles"] = mean.get('SQ_WAIT_ANY', 0) / mean['SQ_WAVE_CYCLES']
        out["valu_active_over_wave_cycles"] = mean.get('SQ_ACTIVE_INST_VALU', 0) / mean['SQ_WAVE_CYCLES']
    if mean.get('SQ_INSTS_VALU') and dur:
        # vector-ALU issue: a wave64 instruction occupies its SIMD's 16 lanes for 4 cycles; 1024 SIMDs; SQ_INSTS_VALU counts
        # instructions per wavefront summed over the device.  Against the kernel's duration at the shader clock the
        # calibration kernels imply (~1.94 GHz under load, profiles/r05_box_calibration.txt)
        cyc = (sum(dur) / len(dur)) * 1e-3 * 1.94e9
        out["valu_insts_per_launch"] = mean['SQ_INSTS_VALU']
        out["valu_issue_frac_of_peak"] = mean['SQ_INSTS_VALU'] * 4.0 / (1024.0 * cyc)
        out["valu_issue_note"] = "SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x kernel duration x 1.94 GHz)"
    if mean.get('TCC_HIT_sum') is not None and mean.get('TCC_MISS_sum') is not None:
        t = mean['TCC_HIT_sum'] + mean['TCC_MISS_sum']
        out["l2_hit_rate"] = mean['TCC_HIT_sum'] / t if t else None
    json.dump(out, open('gpurun_out/pmc_traffic_%s.json' % w, 'w'), indent=1)
    print(w, json.dumps({k: out.get(k) for k in ("hbm_bytes_per_launch", "kernel_ms_under_profiler", "wait_any_over_wave_cycles", "l2_hit_rate")}))
PY
