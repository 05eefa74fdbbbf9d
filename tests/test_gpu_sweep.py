"""`-m gpu`: the randomised robustness sweep as a test (tests/parity_cases.py::case_random_sweep): random shapes x
{box constraints, Dense / Diagonal Hessian, the three DenseBackend values} x {cold solve, update(g) + warm
re-solve on the restored, edited Schur factor}.  0 failures; every SOLVED QP carries the oracle's Info
counters; the infeasible instances whose two sides end with different non-SOLVED statuses stay a small
fraction of the unsolved ones (measured: 16 of 660 over three seeds, profiles/r03_random_sweep.log).  Every third shape runs
the PDAL merit function: full gate on the QPs whose two sides walk the same path (276 of 280), status + solution on the rest."""
import pytest

import parity_cases as pc
from proxsuite_amd import _native as N

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_random_sweep(oracle, randqp, seed):
    import warnings
    r = pc.case_random_sweep(N.load(), oracle, randqp, seed, 60)
    print("sweep seed", seed, r)
    # (in the warnings summary, i.e. in the tail of a quiet run: the parity residue stays visible -- VERDICT r4 item 7)
    warnings.warn(UserWarning("random sweep seed %d: PDAL paths forked on %d of %d QPs (full gate on the others); %d solved with the "
                              "oracle's Info, %d infeasible alike, %d status forks among the unsolved"
                              % (seed, r["pdal_forked"], r["pdal_same_path"] + r["pdal_forked"], r["solved"], r["unsolved_alike"], r["forks"])))
    assert r["failures"] == 0, r
    assert r["info_mismatch"] == 0, r
    assert r["solved"] >= 150, r
    assert r["forks"] <= max(3, 0.08 * (r["unsolved_alike"] + r["forks"])), r
    # PDAL merit function: the full gate (x to XYZ_TOL, Info counters equal) wherever both sides walk the same path --
    # measured 82 / 88 / 78 QPs of 82 / 90 / 78 --, status + solution on the few whose path forks at a breakpoint tie
    assert r["pdal_same_path"] >= 60 and r["pdal_forked"] <= 0.05 * (r["pdal_same_path"] + r["pdal_forked"]), r


def test_random_sweep_large_shapes(oracle, randqp):
    """the same sweep with n in 150 .. 420 (up to ~1300 constraint rows with boxes): the 512- and 1024-thread kernels
    and the vectors-in-HBM kernel on random data, warm re-solves on edited factors included"""
    r = pc.case_random_sweep(N.load(), oracle, randqp, 23, 16, n_range=(150, 420))
    print("sweep large", r)
    assert r["failures"] == 0, r
    assert r["info_mismatch"] == 0, r
    assert r["solved"] >= 40, r
    assert r["forks"] <= 3, r
    assert r["pdal_same_path"] >= 20 and r["pdal_forked"] <= 4, r  # (measured 28 and 2)


def test_random_sweep_dense_wave_kernel(oracle, randqp, monkeypatch):
    """the sweep with the one-wavefront dense kernel forced (PQP_DENSE_KERNEL=wave: every launch of its signature -- dense
    Hessian, no box, PrimalDualLDLT / Automatic, n, n_eq, n_in <= 128 -- whatever its size), n in 2 .. 60: cold solves, warm
    re-solves on the restored, edited Schur factor, infeasible instances, PDAL shapes -- same gates as above"""
    monkeypatch.setenv("PQP_DENSE_KERNEL", "wave")
    r = pc.case_random_sweep(N.load(), oracle, randqp, 31, 60, n_range=(2, 60))
    print("sweep wave", r)
    assert r["failures"] == 0, r
    assert r["info_mismatch"] == 0, r
    assert r["solved"] >= 120, r
    assert r["forks"] <= max(3, 0.08 * (r["unsolved_alike"] + r["forks"])), r
    assert r["pdal_forked"] <= max(2, 0.05 * (r["pdal_same_path"] + r["pdal_forked"])), r
