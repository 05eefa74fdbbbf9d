"""one shape of a random sweep stream with a forced dense kernel: python scripts/dev/sweep_one.py <wave|workgroup> <seed> <it> <lo> <hi>"""
import os, sys, json
sys.path.insert(0, "."); sys.path.insert(0, "tests")
os.environ["PQP_DENSE_KERNEL"] = sys.argv[1]
import parity_cases as pc
from proxsuite_amd import _native as N
from proxsuite_amd.utils import random_qp as randqp
from oracle import oracle
seed, it, lo, hi = (int(v) for v in sys.argv[2:6])
r = pc.case_random_sweep(N.load(), oracle, randqp, seed, it + 1, verbose=True, n_range=(lo, hi), only=it)
print(sys.argv[1], "seed", seed, "it", it, json.dumps(r))
