#!/usr/bin/env python
"""profiles/r06_pmc_c2_insts_by_phase.json: the per-phase budget of a C2 solve on the one-wavefront dense path (VERDICT r5
item 1(i)) from what the instrumented build and the PMC passes measured --
  * device cycles per phase, mean per QP, on a full device (2048 QPs: profiles/r06_dwave_phases.txt, section v2) and on an idle
    one (256 QPs, one wavefront per CU: profiles/r06_dwave_phases_B256.txt),
  * executed instructions of the two kernels of a launch (rocprofv3 --pmc, profiles/r06_pmc_dwave_final.txt),
  * the algorithmic minimum of each phase: its multiply-adds / 64 lanes, from the event counts of the same run.
The one-wavefront kernel has no PQP_REPEAT_PHASE differencing (a wavefront's phases are not idempotent in registers): the
instruction counters are per KERNEL, the split over phases is by device cycles.
    python scripts/phase_budget_r06.py"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def phases(path, section=None):
    txt = open(path).read()
    if section:
        txt = txt[txt.index(section):]
    out = {}
    for m in re.finditer(r"^(CYC_\w+|N_\w+|BYTES_ENGINE|FLOPS_FACT|WALL_TICKS)\s+mean\s+([\d.]+)", txt, re.M):
        out.setdefault(m.group(1).lower(), float(m.group(2)))
    return out


def counters(path, tag, kernel):
    txt = open(path).read()
    blk = txt[txt.index("== %s  void %s" % (tag, kernel)):]
    line = blk.splitlines()[1]
    return {k: float(v) for k, v in re.findall(r"(\w+)=([\d.e+]+)", line)}


full = phases(os.path.join(ROOT, "profiles", "r06_dwave_phases_final.txt"))
idle = phases(os.path.join(ROOT, "profiles", "r06_dwave_phases_final_B256.txt"))
pm = os.path.join(ROOT, "profiles", "r06_pmc_dwave_final.txt")
cw = counters(pm, "libproxqp_hip", "pqp_dwave_kernel<2>")
cp = counters(pm, "libproxqp_hip", "pqp_prologue_kernel<256>")
n, ne, ni = 100, 50, 100
newton, facts = full["n_newton"], full["n_schur_fact"]
r = ne + full["n_active_final"]  # slots of the dual block at the end of a solve (a little fewer on the way)
# multiply-adds per occurrence
fma = {
    "kkt_solve (W, Z_J twice, W_S twice, W^T; incl. the Schur solve)": newton * (n * n + 2 * r * n + r * r),
    "kkt_residual (H_s from its lower triangle, A_s, C_s: both products of each from one pass)": newton * (n * n + 2 * ne * n + 2 * ni * n),
    "global residuals (H_s x, A_s x / A_s^T y, C_s x / C_s^T z per outer iteration)": 8 * (n * n + 2 * ne * n + 2 * ni * n),
    "schur factorisation (LDL^T + inverse of the r x r block, r ~ 85)": facts * 2 * r ** 3 / 3,
    "schur append / delete (one row of W_S per change)": (full["n_append"] + full["n_delete"]) * r * r,
    "prologue kernel (LDL^T + inverse of H_s + rho I, Z = L^-1 B^T, G = Z^T D^-1 Z)": 2 * n ** 3 / 3 + n * n * (ne + ni) / 2 + (ne + ni) ** 2 * n / 2,
}
cyc_map = {
    "kkt_solve (W, Z_J twice, W_S twice, W^T; incl. the Schur solve)": ("cyc_kkt_solve", "cyc_solve_ldlt"),
    "kkt_residual (H_s from its lower triangle, A_s, C_s: both products of each from one pass)": ("cyc_residual",),
    "global residuals (H_s x, A_s x / A_s^T y, C_s x / C_s^T z per outer iteration)": ("cyc_global_res",),
    "schur factorisation (LDL^T + inverse of the r x r block, r ~ 85)": ("cyc_f_load", "cyc_f_update", "cyc_f_writeback", "cyc_s_gather"),
    "schur append / delete (one row of W_S per change)": ("cyc_schur",),
    "prologue kernel (LDL^T + inverse of H_s + rho I, Z = L^-1 B^T, G = Z^T D^-1 Z)": ("cyc_factor_h",),
    "line search (no matrix)": ("cyc_linesearch",),
    "active-set bookkeeping, right-hand sides, certificates, iterate update (no matrix)": ("cyc_newton_misc", "cyc_zg"),
}
rows = {}
for name, keys in cyc_map.items():
    rows[name] = {"cycles_per_qp_full_device": sum(full.get(k, 0.0) for k in keys),
                  "share_of_a_wavefronts_cycles_full_device": sum(full.get(k, 0.0) for k in keys) / full["cyc_total"],
                  "cycles_per_qp_idle_device": sum(idle.get(k, 0.0) for k in keys),
                  "share_idle_device": sum(idle.get(k, 0.0) for k in keys) / idle["cyc_total"]}
    if name in fma:
        rows[name]["multiply_adds_per_qp"] = fma[name]
        rows[name]["minimum_wave_instructions_per_qp"] = fma[name] / 64.0
out = {
    "workload": "C2: 2048 x (100, 50, 100), eps_abs 1e-9, NO_INITIAL_GUESS; one launch = pqp_prologue_kernel<256> + pqp_dwave_kernel<2>",
    "events_per_qp": {k: full[k] for k in ("n_newton", "n_schur_fact", "n_kkt_solves", "n_append", "n_delete", "n_ls_breakpoints",
                                           "n_active_final", "bytes_engine")},
    "executed_instructions_per_qp": {
        "pqp_dwave_kernel<2>": {"valu": cw["SQ_INSTS_VALU"] / 2048, "salu": cw["SQ_INSTS_SALU"] / 2048, "mfma": cw["SQ_INSTS_MFMA"] / 2048,
                                "lds": cw["SQ_INSTS_LDS"] / 2048, "vmem_read": cw["SQ_INSTS_VMEM_RD"] / 2048,
                                "vmem_write": cw["SQ_INSTS_VMEM_WR"] / 2048,
                                "waves_waiting_frac": cw["SQ_WAIT_INST_ANY"] / cw["SQ_WAVE_CYCLES"],
                                "waves_executing_frac": cw["SQ_ACTIVE_INST_ANY"] / cw["SQ_WAVE_CYCLES"]},
        "pqp_prologue_kernel<256>": {"valu": cp["SQ_INSTS_VALU"] / 2048, "salu": cp["SQ_INSTS_SALU"] / 2048, "mfma": cp["SQ_INSTS_MFMA"] / 2048,
                                     "lds": cp["SQ_INSTS_LDS"] / 2048, "vmem_read": cp["SQ_INSTS_VMEM_RD"] / 2048,
                                     "vmem_write": cp["SQ_INSTS_VMEM_WR"] / 2048,
                                     "waves_waiting_frac": cp["SQ_WAIT_INST_ANY"] / cp["SQ_WAVE_CYCLES"]},
        "round_5_workgroup_kernel_for_comparison": {"valu": 911000, "salu": 256000, "lds": 110000, "vmem_read": 36000,
                                                    "source": "profiles/r05_pmc_c2.json (pqp_solve_kernel<256,4,1>, per workgroup-QP)"}},
    "minimum_wave_instructions_per_qp_all_matrix_phases": sum(v for v in fma.values()) / 64.0,
    "phases": rows,
    "reading": "The mat-vec phases (KKT solve, residuals, Schur edits) need ~%.0f k wavefront multiply-add instructions per QP and the "
               "factorisations (matrix cores) the equivalent of ~%.0f k; the iteration kernel executes 383 k vector "
               "instructions (911 k in the round-5 workgroup kernel): a row of a pass costs 3 instructions to fetch (two "
               "broadcasts of its descriptor fields, the load) and 4 to consume (two broadcasts of its coefficient, two "
               "multiply-adds) plus 17 / 16 matrix-core instructions where its row sum is wanted, so a pass runs at 2 "
               "useful of ~8 executed vector instructions; the rest is the line search, the active-set bookkeeping and 757 "
               "spilled registers.  On a FULL device the phases' shares follow their HBM bytes (the kernel moves 31.2 GB per "
               "launch at 4.9 TB/s); on an IDLE one (the tail of every launch) a batch of 16 rows costs one memory round "
               "trip (1.2 us) plus its row-sum reduction (0.5 us), ~64 batches per Newton step." % (
                   sum(v for k, v in fma.items() if "factorisation" not in k and "prologue" not in k) / 64e3,
                   sum(v for k, v in fma.items() if "factorisation" in k or "prologue" in k) / 64e3),
    "sources": ["profiles/r06_dwave_phases_final.txt", "profiles/r06_dwave_phases_final_B256.txt", "profiles/r06_pmc_dwave_final.txt"],
}
json.dump(out, open(os.path.join(ROOT, "profiles", "r06_pmc_c2_insts_by_phase.json"), "w"), indent=1)
print(json.dumps(out["executed_instructions_per_qp"]["pqp_dwave_kernel<2>"], indent=1))
for k, v in rows.items():
    print("%-95s full %5.1f %%  idle %5.1f %%  min instr %s" % (k[:95], 100 * v["share_of_a_wavefronts_cycles_full_device"],
          100 * v["share_idle_device"], ("%.0f" % v["minimum_wave_instructions_per_qp"]) if "minimum_wave_instructions_per_qp" in v else "-"))
