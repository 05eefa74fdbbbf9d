"""Randomised robustness sweep (tests/parity_cases.py::case_random_sweep) from the command line:
python scripts/gpu_sweep2.py <seed> <count> [library.so] ; PQP_SWEEP_N=lo,hi picks the range of n (default 1,120)"""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from proxsuite_amd import _native as N
from proxsuite_amd.utils import random_qp as R
from oracle import oracle as O
import parity_cases as pc

lib = N.NativeLib(sys.argv[3]) if len(sys.argv) > 3 else N.load()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
import os
nr = tuple(int(v) for v in os.environ.get('PQP_SWEEP_N', '1,120').split(','))
only = os.environ.get('PQP_SWEEP_ONLY')
r = pc.case_random_sweep(lib, O, R, seed, count, n_range=nr, only=None if only is None else int(only))
print("sweep2: %d shapes x 3 QPs x 2 phases, %d failures, %d QPs unsolved alike in the oracle, %d infeasible QPs "
      "ending with different non-SOLVED statuses, %d SOLVED QPs compared of which %d with different Info counters, %.1f s"
      % (count, r["failures"], r["unsolved_alike"], r["forks"], r["solved"], r["info_mismatch"], r["seconds"]))
