"""`-m "not gpu"`: a short leg of the randomised sweep on the CPU emulator of the device code (small shapes
only: the emulator runs one fiber per GPU thread)."""
import os
import sys

import numpy as np
import pytest

import parity_cases as pc
from proxsuite_amd import _native as N

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))


@pytest.mark.parametrize("seed", [5, 10, 11])  # (seeds whose infeasible instances give up quickly: the emulator is slow)
def test_random_sweep_small(oracle, randqp, monkeypatch, seed):
    import build as emu_build
    lib = N.NativeLib(emu_build.build())
    # shrink the shapes the sweep draws: n < 24 (the generator is seeded, the cap only rescales the draw)
    real = np.random.default_rng

    class Small:
        def __init__(self, seed):
            self.r = real(seed)
            self.first = True

        def integers(self, lo, hi=None, *a, **k):
            v = self.r.integers(lo, hi, *a, **k)
            if hi == 120:  # the `n` draw
                v = 1 + int(v) % 23
            return v

        def __getattr__(self, name):
            return getattr(self.r, name)

    monkeypatch.setattr(np.random, "default_rng", lambda seed: Small(seed))
    r = pc.case_random_sweep(lib, oracle, randqp, seed, 7, verbose=True)
    assert r["failures"] == 0 and r["info_mismatch"] == 0, r
    assert r["solved"] + r["unsolved_alike"] + r["forks"] == 7 * 3 * 2, r
    assert r["pdal_forked"] <= 2, r  # (PDAL shapes: the full gate on every QP whose two sides walk the same path)


def test_random_sweep_dense_wave_kernel(oracle, randqp, monkeypatch):
    """a short leg of the sweep with the one-wavefront dense kernel forced on every launch of its signature (csrc/pqp_dwave.hpp)"""
    import build as emu_build
    lib = N.NativeLib(emu_build.build())
    monkeypatch.setenv("PQP_DENSE_KERNEL", "wave")
    r = pc.case_random_sweep(lib, oracle, randqp, 5, 4, verbose=True, n_range=(2, 16))
    assert r["failures"] == 0 and r["info_mismatch"] == 0, r
    assert r["solved"] + r["unsolved_alike"] + r["forks"] == 4 * 3 * 2, r
    assert r["pdal_forked"] <= 2, r
