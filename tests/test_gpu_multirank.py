"""`-m gpu`: the N > 1 path of bench.py on the one-GPU test box -- two ranks sharing device 0 (RCCL refuses
that, so the process group runs on gloo): each rank solves its own shard, the DEVICE pack kernel builds the
payload of the path's only collective, the gathered batch is checked bit for bit on every rank.  The RCCL
all_gather_into_tensor itself needs two GPUs and runs in the driver's multi-GPU tier."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("extra", [[], ["--total-batch", "512"]])
def test_two_ranks_on_one_device(extra):
    env = dict(os.environ, PQP_BENCH_ONE_DEVICE="1", PQP_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "256", "--no-cpu-baseline", "--mpc-steps", "0"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["unsolved"] == 0 and line["max_kkt_residual"] <= 1e-9
    assert line["scaling"] == ("strong" if extra else "weak")
    assert line["config"]["batch_per_gpu"] == 256
    d = line["distributed"]
    assert d["world_size"] == 2 and len(d["per_rank"]) == 2 and all(p["qps"] > 0 for p in d["per_rank"])
    assert line["gather_bit_exact_on_every_rank"] is True and line["gather_ms"] > 0
    assert "pack kernel" in line["gather"]


def test_rccl_single_rank():
    """RCCL itself as far as one GPU allows: bench.py --gpus 1 --rccl-single-rank initialises the "nccl" process group
    (RCCL on ROCm) with one rank and sends the path's only collective through it -- the device pack kernel, then
    all_gather_into_tensor on ROCm tensors -- and checks the gathered batch bit for bit (the N-rank form of the same
    call runs in the driver's multi-GPU tier; reference parallel/qp_solve.hpp:41-59 has no exchange at all)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "PQP_BENCH_BACKEND", "PQP_BENCH_ONE_DEVICE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--rccl-single-rank", "--steps", "2", "--warmup", "1",
           "--batch", "256", "--no-cpu-baseline", "--mpc-steps", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["unsolved"] == 0 and line["max_kkt_residual"] <= 1e-9
    d = line["distributed"]
    assert d["world_size"] == 1 and d["backend"] == "nccl" and d["rccl_version"], d
    assert "RCCL" in line["gather"] and line["gather_ms"] > 0
    assert line["gather_bit_exact_on_every_rank"] is True
