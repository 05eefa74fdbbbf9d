"""Development harness: the one-wavefront dense kernel (PQP_DENSE_KERNEL=wave) on the CPU emulator against the oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
os.environ["PQP_DENSE_KERNEL"] = sys.argv[1] if len(sys.argv) > 1 else "wave"
import numpy as np
import build as emu_build
from proxsuite_amd import _native as N
from proxsuite_amd.utils import random_qp as R
from oracle import oracle as O
import parity_cases as pc
from proxsuite_amd._ctypes_defs import InitialGuess

lib = N.NativeLib(emu_build.build())
shapes = [(10, 2, 3, 4), (30, 7, 9, 4), (33, 8, 11, 3), (12, 0, 9, 3), (9, 5, 0, 3), (50, 25, 50, 2)]
if len(sys.argv) > 2:
    shapes = [tuple(int(v) for v in sys.argv[2].split(","))]
for (n, ne, ni, B) in shapes:
    t0 = time.time()
    try:
        pc.case_random_batch(lib, O, R, n, ne, ni, B)
        print("ok", (n, ne, ni, B), "%.1fs" % (time.time() - t0), flush=True)
    except AssertionError as e:
        print("FAIL", (n, ne, ni, B), e, flush=True)

# which kernel ran? (launch configuration + a counter only the one-wavefront kernel raises at these sizes)
m = R.dense_strongly_convex_qp_batch(2, 20, 5, 8, 0.15, 1e-2)
b = N.Batch(2, 20, 5, 8, lib=lib)
pc.settings_all(b, eps_abs=1e-9, eps_rel=0, initial_guess=int(InitialGuess.NO_INITIAL_GUESS))
b.init(-1, m.H, m.g, m.A, m.b, m.C, m.l, m.u)
b.solve()
print("launch_config", b.launch_config(), "stats[ST_N_SCHUR_BLOCKED]", b.stats()[:, 23])
