"""Every solve kernel is one fully inlined function whose register allocation decides its speed, and an edit anywhere in
pqp_solver.hpp / pqp_block.hpp can shift the allocation of a kernel that never executes the edited code (round 4: a
never-executed routine cost the C2 kernel 12 %, unrelated dense-path edits cost C5 5 %).  The expectations below are
FROZEN (tests/golden/kernel_resources_expected.json, written by `python -m proxsuite_amd._build --freeze` after a change
has been measured on the GPU): a build whose kernels drift from them fails HERE, on the CPU, at build time, instead of
costing throughput silently."""
import json
import os

import pytest

from proxsuite_amd import _build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXPECTED = os.path.join(ROOT, "tests", "golden", "kernel_resources_expected.json")

# allowed drift per field (absolute, relative): a few registers of noise between compiler runs are not a regression
TOL = {"VGPRs": (4, 0.0), "AGPRs": (8, 0.0), "VGPRs_Spill": (12, 0.05), "ScratchSize": (48, 0.05), "SGPRs_Spill": (40, 0.05),
       "Occupancy": (0, 0.0)}


def _record():
    _build.build_hip()  # no-op when the library is newer than its sources; the record is that build's
    rec = _build.kernel_resources()
    if not rec:
        pytest.skip("no kernel-resource record of the product build in build/obj/default (library prebuilt elsewhere)")
    return rec


def test_every_kernel_matches_its_frozen_resources():
    rec = _record()
    exp = json.load(open(EXPECTED))
    missing = sorted(k for k in exp if k.startswith("pqp_") and k not in rec)
    assert not missing, "kernels of the frozen record that the build no longer produces: %s" % missing
    new = sorted(k for k in rec if k.startswith("pqp_") and k not in exp)
    assert not new, "kernels without a frozen expectation (python -m proxsuite_amd._build --freeze): %s" % new
    drift = []
    for k, e in exp.items():
        for f, (ab, rel) in TOL.items():
            if f in e and abs(rec[k].get(f, 0) - e[f]) > max(ab, rel * abs(e[f])):
                drift.append("%s %s: %s -> %s" % (k, f, e[f], rec[k].get(f)))
    assert not drift, "register allocation drifted from the frozen record:\n  " + "\n  ".join(drift)


def test_headline_kernel_budget():
    rec = _record()
    c2 = rec["pqp_solve_kernel<256,4,1>"]
    assert c2["VGPRs"] <= 128 and c2["Occupancy"] == 4  # four workgroups of four wavefronts per CU
    # a loop that the unroller leaves rolled puts the register tile it indexes into scratch memory: 492 -> 1072 B/lane took
    # a C2 launch from 8.1 to 27.6 ms, 60 GB of scratch stores reaching HBM (profiles/r05_ab_gj_two_pivots.txt)
    assert c2["ScratchSize"] <= 560, c2
    for k, v in rec.items():  # kernels that must not spill a single vector register
        if k in ("pqp_solve_kernel<256,1,1>", "pqp_solve_kernel<256,2,1>", "pqp_solve_kernel<256,2,2>", "pqp_solve_kernel<256,1,0>"):
            assert v["VGPRs_Spill"] == 0 and v["ScratchSize"] == 0, (k, v)


def test_no_register_array_lives_in_scratch_memory():
    """Scratch beyond what the spilled registers need means a private array that the compiler could not keep in registers
    (a loop left rolled by the unroller's size limit indexes it dynamically): the two-pivot Gauss-Jordan experiment of round 5
    had 464 B of it with NO spilled register, and the C2 launch went from 8.1 to 27.6 ms
    (profiles/r05_ab_gj_two_pivots.txt).  Spill slots are 4 bytes per register and are shared, never more."""
    rec = _record()
    bad = {k: (v["ScratchSize"], v["VGPRs_Spill"]) for k, v in rec.items()
           if k.startswith("pqp_") and v.get("ScratchSize", 0) > 4 * v.get("VGPRs_Spill", 0) + 16}
    assert not bad, "kernels with private arrays in scratch memory (ScratchSize, VGPRs_Spill): %s" % bad
