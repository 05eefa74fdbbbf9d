"""N>1 path on CPU: two gloo processes, each solving its shard of one batch (on the SIMT emulator
build of the device code -- test only) and all-gathering the solutions.  The gathered batch must
equal the single-process solve bit for bit (QPs are independent; reference
test/src/parallel_qp_solve.cpp:74-76 checks parallel == serial the same way)."""
import os
import socket
import sys

import numpy as np
import pytest

from proxsuite_amd.sharding import shard_range

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_range_partitions():
    for total in (0, 1, 5, 16, 2048, 16385):
        for world in (1, 2, 3, 8):
            cuts = [shard_range(total, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == total
            assert all(cuts[r][1] == cuts[r + 1][0] for r in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _worker(rank, world, port, total, shape, emu_path, out_dir):
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(HERE))
    from proxsuite_amd import _native as N
    from proxsuite_amd._ctypes_defs import InitialGuess
    from proxsuite_amd.sharding import ShardedBatch
    from proxsuite_amd.utils import random_qp as R
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HIPEMU_THREADS"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, ne, ni = shape
        lib = N.NativeLib(emu_path)
        sb = ShardedBatch(total, n, ne, ni, rank, world, device=0, lib=lib)
        m = R.dense_strongly_convex_qp_batch(sb.local, n, ne, ni, 0.15, 1e-2, seed0=sb.first)
        for i in range(sb.local):
            s = sb.batch.settings(i)
            s.eps_abs, s.eps_rel, s.initial_guess = 1e-9, 0.0, int(InitialGuess.NO_INITIAL_GUESS)
        sb.batch.init(-1, m.H, m.g, m.A, m.b, m.C, m.l, m.u)
        sb.solve()
        # the device-side pack kernel (here: on the emulator, into a host buffer) must produce the
        # rows the host path builds
        buf = np.zeros((sb.local, sb.width))
        sb.batch.pack_results(buf, 0, sb.local)
        lx, ly, lz, lst, lit = sb.local_results()
        assert np.array_equal(buf[:, :n], lx) and np.array_equal(buf[:, n:n + ne], ly)
        assert np.array_equal(buf[:, n + ne:-2], lz)
        assert np.array_equal(buf[:, -2].astype(np.int64), lst) and np.array_equal(buf[:, -1].astype(np.int64), lit)
        x, y, z, status, iters = sb.gather()
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), x=x, y=y, z=z, status=status, iters=iters)
    finally:
        dist.destroy_process_group()


def test_two_rank_shards_equal_single_process(tmp_path, randqp):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(HERE, "emu"))
    import build as emu_build
    from proxsuite_amd import _native as N
    from proxsuite_amd._ctypes_defs import InitialGuess
    emu = str(emu_build.build())
    total, shape = 5, (10, 3, 4)  # odd total: shards of 3 and 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, total, shape, emu, str(tmp_path)), nprocs=2, join=True)
    # single-process reference of the same batch
    n, ne, ni = shape
    m = randqp.dense_strongly_convex_qp_batch(total, n, ne, ni, 0.15, 1e-2, seed0=0)
    b = N.Batch(total, n, ne, ni, lib=N.NativeLib(emu))
    for i in range(total):
        st = b.settings(i)
        st.eps_abs, st.eps_rel, st.initial_guess = 1e-9, 0.0, int(InitialGuess.NO_INITIAL_GUESS)
    b.init(-1, m.H, m.g, m.A, m.b, m.C, m.l, m.u)
    b.solve()
    x, y, z, _, _, info = b.results()
    for r in range(2):
        d = np.load(tmp_path / ("rank%d.npz" % r))
        assert np.array_equal(d["x"], x) and np.array_equal(d["y"], y) and np.array_equal(d["z"], z)
        assert list(d["status"]) == [info[i].status for i in range(total)] == [0] * total
        assert list(d["iters"]) == [info[i].iter for i in range(total)]
