import os, sys, json
sys.path.insert(0, "."); sys.path.insert(0, "tests")
os.environ["PQP_DENSE_KERNEL"] = "wave"
import parity_cases as pc
from proxsuite_amd import _native as N
from proxsuite_amd.utils import random_qp as randqp
from oracle import oracle
lib = N.load()
for seed, rng in ((41, (2, 64)), (42, (40, 128)), (43, (2, 128)), (44, (90, 128))):
    r = pc.case_random_sweep(lib, oracle, randqp, seed, 50, verbose=False, n_range=rng)
    print("seed", seed, rng, json.dumps(r), flush=True)
