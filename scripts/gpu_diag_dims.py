"""QPs/s of the one-wavefront diagonal kernel by dimension (1 / 2 / 4 register slots per vector: dim <= 64 / 128 / 256):
  python scripts/gpu_diag_dims.py [B]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_cases as pc
from proxsuite_amd import _native as N
from proxsuite_amd.utils import random_qp as randqp
from proxsuite_amd._ctypes_defs import HessianType, InitialGuess

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for dim in (32, 64, 65, 128, 129, 200, 256):
    H, g, Cm, l, u = pc.c5_models(randqp, B, dim)
    b = N.Batch(B, dim, 0, dim, hessian_type=int(HessianType.Diagonal))
    b.set_all_settings(eps_abs=1e-9, eps_rel=0.0, initial_guess=int(InitialGuess.NO_INITIAL_GUESS))
    b.init(-1, H, g, None, None, Cm, l, u)
    b.solve()
    ms = []
    for _ in range(6):
        b.solve()
        ms.append(b.last_solve_ms)
    infos = b.infos()
    nt, lds = b.launch_config()
    print("dim %3d  threads %d lds %5d B  kernel %.3f ms  %.2f M QPs/s  unsolved %d" % (
        dim, nt, lds, min(ms), B / min(ms) / 1e3, sum(1 for i in range(B) if infos[i].status != 0)), flush=True)
    b.close()
