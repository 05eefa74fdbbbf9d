"""Per-phase HBM traffic of the C2 solve from the runs of scripts/gpu_r4_call1.sh: counter bytes of the solve kernel with
phase k executed twice minus the plain run, against the same difference of the engine-byte counter."""
import csv, glob, json, sys, collections
O = sys.argv[1]
NAMES = {0: "plain", 1: "line search", 2: "KKT residual", 3: "KKT solve (first of a step)", 4: "Schur re-factorisation",
         5: "global residuals", 6: "primal block + Z / G"}
def counter(ph, ctr):
    per = collections.defaultdict(float)
    for f in glob.glob('%s/pmc_%d_%s/**/*counter_collection.csv' % (O, ph, ctr), recursive=True):
        for row in csv.DictReader(open(f)):
            if 'pqp_solve_kernel' in row.get('Kernel_Name', '') and row['Counter_Name'] == ctr:
                per[row['Dispatch_Id']] += float(row['Counter_Value'])
    v = sorted(per.items(), key=lambda kv: int(kv[0]))
    return [x for _, x in v]
out = {}
base = None
for ph in range(7):
    try:
        j = json.load(open('%s/phase_%d_FETCH_SIZE.json' % (O, ph)))
    except Exception as e:
        print("phase", ph, "missing:", e); continue
    f, w = counter(ph, 'FETCH_SIZE'), counter(ph, 'WRITE_SIZE')
    if not f:
        print("phase", ph, "no counters"); continue
    if not w:  # phases that write nothing to HBM were only profiled for reads: the plain run's writes stand in
        w = counter(0, 'WRITE_SIZE')
    B = j["B"]
    # the first launch of a fresh batch differs (first touch); use the last one
    rd, wr = 1024.0 * 2.0 * f[-1] / B, 1024.0 * w[-1] / B
    rec = {"phase": NAMES[ph], "read_bytes_per_qp": rd, "write_bytes_per_qp": wr, "traffic_per_qp": rd + wr,
           "engine_bytes_per_qp": j["bytes_engine"], "kernel_ms": j["kernel_ms"][-1], "unsolved": j["unsolved"], "iter_sum": j["iter_sum"]}
    if ph == 0:
        base = rec
    else:
        rec["delta_traffic"] = rec["traffic_per_qp"] - base["traffic_per_qp"]
        rec["delta_engine"] = rec["engine_bytes_per_qp"] - base["engine_bytes_per_qp"]
        rec["traffic_over_engine"] = rec["delta_traffic"] / rec["delta_engine"] if rec["delta_engine"] else None
        rec["same_iterates"] = rec["iter_sum"] == base["iter_sum"]
    out[ph] = rec
json.dump(out, open('%s/waste_by_phase.json' % O, 'w'), indent=1)
if base:
    print("plain: traffic %.2f MB/QP (read %.2f, write %.2f), engine %.2f MB/QP, ratio %.3f" % (
        base["traffic_per_qp"] / 1e6, base["read_bytes_per_qp"] / 1e6, base["write_bytes_per_qp"] / 1e6,
        base["engine_bytes_per_qp"] / 1e6, base["traffic_per_qp"] / base["engine_bytes_per_qp"]))
    acc_t = acc_e = 0.0
    for ph in range(1, 7):
        if ph in out:
            r = out[ph]
            acc_t += r["delta_traffic"]; acc_e += r["delta_engine"]
            print("%-28s traffic %.2f MB/QP  engine %.2f MB/QP  ratio %s  excess %.2f MB/QP  (same iterates: %s)" % (
                r["phase"], r["delta_traffic"] / 1e6, r["delta_engine"] / 1e6,
                "%.2f" % r["traffic_over_engine"] if r["traffic_over_engine"] else "-",
                (r["delta_traffic"] - r["delta_engine"]) / 1e6, r["same_iterates"]))
    print("rest (Schur edits, right-hand sides, update, prologue / epilogue vectors): traffic %.2f MB/QP engine %.2f MB/QP" % (
        (base["traffic_per_qp"] - acc_t) / 1e6, (base["engine_bytes_per_qp"] - acc_e) / 1e6))
