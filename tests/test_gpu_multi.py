"""`-m gpu`: asynchronous solve, host-resident results and the in-process multi-device batch (pqp_multi_*) on the
MI355X through the C-ABI of libproxqp_hip.so.  The box has one GPU: the shards of the multi-device batch are logical
shards on device 0 (G = 2 and G = 4), each with its own stream -- bit-exact against the single-handle solve."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import multi_cases as mc
from proxsuite_amd import _native as N

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    return N.load()


def _dev_alloc(count):
    import torch
    return torch.zeros(count, dtype=torch.float64, device="cuda:0")


def test_async_solve_and_host_results(lib, randqp):
    mc.case_async_and_host_results(lib, randqp)
    mc.case_async_and_host_results(lib, randqp, n=100, ne=50, ni=100, B=24)


def test_two_handles_in_flight(lib, randqp):
    mc.case_two_handles_in_flight(lib, randqp)
    mc.case_two_handles_in_flight(lib, randqp, n=60, ne=20, ni=40, B=300)


@pytest.mark.parametrize("G", [2, 4])
def test_multi_device_batch_is_bit_exact(lib, randqp, G):
    mc.case_multi(lib, randqp, [0] * G, gather_alloc=_dev_alloc)
    mc.case_multi(lib, randqp, [0] * G, n=100, ne=50, ni=100, B=64 + G - 1, gather_alloc=_dev_alloc)


def test_multi_verbose_trace(lib, randqp):
    mc.case_multi_verbose_trace(lib, randqp, [0, 0, 0])


def test_multi_errors(lib):
    mc.case_multi_errors(lib)


def test_bench_inprocess_mode():
    """bench.py --gpus 2 --inprocess (shards on device 0): one JSON line, every QP solved, sharded == single handle"""
    env = dict(os.environ, PQP_BENCH_ONE_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--inprocess", "--batch", "256",
                        "--steps", "2", "--warmup", "1"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["unsolved"] == 0 and line["max_kkt_residual"] <= 1e-9
    assert line["gather_bit_exact"] and line["sharded_equals_single_handle_bit_for_bit"]
