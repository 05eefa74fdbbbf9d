"""`-m gpu`: regression guard on the headline kernel.  A fifth of its throughput hangs on three internal
`-mllvm` code-generation switches (proxsuite_amd/_build.py) that a toolchain change could alter silently:
the C2 solve (2048 random dense QPs, n=100 n_eq=50 n_in=100, index order) must stay within the margin (7 %) of the
kernel time recorded in profiles/perf_guard.json -- the median box of the pool --, so must C4 and C5, and the build's
own record of the kernels' registers (profiles/r04_kernel_resources.json, written by __graft_entry__.build) must be there."""
import json
import os

import pytest

from proxsuite_amd import _native as N
from proxsuite_amd._ctypes_defs import InitialGuess

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c2_kernel_time_within_margin(randqp):
    guard = json.load(open(os.path.join(ROOT, "profiles", "perf_guard.json")))
    lib = N.load()
    B, n, ne, ni = 2048, 100, 50, 100
    m = randqp.dense_strongly_convex_qp_batch(B, n, ne, ni, 0.15, 1e-2)
    b = N.Batch(B, n, ne, ni, lib=lib)
    b.set_all_settings(eps_abs=1e-9, eps_rel=0.0, initial_guess=int(InitialGuess.NO_INITIAL_GUESS))
    b.init(-1, m.H, m.g, m.A, m.b, m.C, m.l, m.u)
    b.solve()  # first touch of the workspace
    ms = []
    for _ in range(5):
        b.solve()
        ms.append(b.last_solve_ms)
    x, y, z, se, si, info = b.results()
    assert all(info[i].status == 0 for i in range(B))
    b.close()
    best = min(ms)
    limit = (1.0 + guard["margin"]) * guard["c2_kernel_ms"]
    assert best <= limit, "C2 solve kernel %.3f ms > %.3f ms (recorded %.3f ms + %.0f %%): %s" % (
        best, limit, guard["c2_kernel_ms"], 100 * guard["margin"], guard["source"])


def test_c4_kernel_time_within_margin(randqp):
    """BASELINE.json configs[3] (512 x (512, 200, 400)): the LDS-tiled Z / G build of the wide kernels must stay in place"""
    guard = json.load(open(os.path.join(ROOT, "profiles", "perf_guard.json")))
    B, n, ne, ni = 512, 512, 200, 400
    m = randqp.dense_strongly_convex_qp_batch(B, n, ne, ni, 0.15, 1e-2)
    b = N.Batch(B, n, ne, ni, lib=N.load())
    b.set_all_settings(eps_abs=1e-9, eps_rel=0.0, initial_guess=int(InitialGuess.NO_INITIAL_GUESS))
    b.init(-1, m.H, m.g, m.A, m.b, m.C, m.l, m.u)
    b.solve()
    ms = []
    for _ in range(3):
        b.solve()
        ms.append(b.last_solve_ms)
    infos = b.infos()
    assert all(infos[i].status == 0 for i in range(B))
    b.close()
    best = min(ms)
    limit = (1.0 + guard["margin"]) * guard["c4_kernel_ms"]
    assert best <= limit, "C4 solve kernel %.3f ms > %.3f ms (recorded %.3f ms): %s" % (best, limit, guard["c4_kernel_ms"], guard["c4_source"])


def test_c5_kernel_time_within_margin(randqp):
    """the structured configuration (BASELINE.json configs[4]: 4096 x (200, 0, 200), diagonal Hessian, C = I), whose time is
    the line search's: the bracketing evaluation and the dedicated kernel must stay in place"""
    import parity_cases as pc
    from proxsuite_amd._ctypes_defs import HessianType
    guard = json.load(open(os.path.join(ROOT, "profiles", "perf_guard.json")))
    B, dim = 4096, 200
    H, g, Cm, l, u = pc.c5_models(randqp, B, dim)
    b = N.Batch(B, dim, 0, dim, hessian_type=int(HessianType.Diagonal), lib=N.load())
    b.set_all_settings(eps_abs=1e-9, eps_rel=0.0, initial_guess=int(InitialGuess.NO_INITIAL_GUESS))
    b.init(-1, H, g, None, None, Cm, l, u)
    b.solve()
    ms = []
    for _ in range(5):
        b.solve()
        ms.append(b.last_solve_ms)
    x, y, z, se, si, info = b.results()
    assert all(info[i].status == 0 for i in range(B))
    b.close()
    best = min(ms)
    limit = (1.0 + guard["margin"]) * guard["c5_kernel_ms"]
    assert best <= limit, "C5 solve kernel %.3f ms > %.3f ms (recorded %.3f ms): %s" % (best, limit, guard["c5_kernel_ms"], guard["c5_source"])


def test_kernel_resource_record_exists():
    res = json.load(open(os.path.join(ROOT, "profiles", "r04_kernel_resources.json")))
    c2 = res["pqp_solve_kernel<256,4,1>"]
    assert c2["VGPRs"] <= 128 and c2["Occupancy"] == 4  # four workgroups of four wavefronts per CU
    assert "VGPRs_Spill" in c2 and "ScratchSize" in c2
    diag = res["pqp_solve_kernel<256,2,2>"]  # the diagonal-structure kernel: nothing spilled
    assert diag["VGPRs_Spill"] == 0 and diag["ScratchSize"] == 0
