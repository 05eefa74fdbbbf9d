"""Multi-GPU form of `solve_in_parallel`: one process per GPU, the batch sharded by QP.

QPs are independent (reference parallel/qp_solve.hpp:33-37: no shared state, no reduction), so
rank r of W owns the contiguous range `shard_range(B, r, W)` and solves it on its own GPU with
no exchange during the solve.  The only collective of the path is the optional final gather of
the solutions (x, y, z, status) for consumers that want the whole batch on every rank:
`torch.distributed.all_gather` -- RCCL over xGMI when the tensors live on the GPUs (backend
"nccl"), gloo on the host (CPU tests).  Consumers that read results per shard skip it.
"""
from __future__ import annotations

import numpy as np


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """[first, last) of the QPs owned by `rank`: contiguous, sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    base, extra = divmod(int(total), int(world))
    first = rank * base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


class ShardedBatch:
    """The local shard of a global batch of `total` QPs of one size, as a native Batch.

    `arrays` passed to init/update are the LOCAL slices ([local_B, ...]); use `.first/.last` to cut
    them out of global arrays.  `lib` is only for tests (emulator build)."""

    def __init__(self, total, n, n_eq, n_in, rank, world, device=None, lib=None, **kw):
        import os
        from . import _native
        self.total, self.rank, self.world = int(total), int(rank), int(world)
        self.first, self.last = shard_range(total, rank, world)
        self.local = self.last - self.first
        if device is None:
            # one process per GPU of the node: the LOCAL rank picks the device (the global rank is
            # only right on a single node)
            device = int(os.environ.get("LOCAL_RANK", rank))
        self.device = int(device)
        self.batch = _native.Batch(max(self.local, 1), n, n_eq, n_in, device=self.device, lib=lib, **kw)
        self.n, self.n_eq, self.n_c = self.batch.n, self.batch.n_eq, self.batch.n_c

    def solve(self):
        if self.local:
            self.batch.solve(0, self.local)

    def local_results(self):
        x, y, z, se, si, info = self.batch.results(-1)
        k = self.local
        status = np.array([info[i].status for i in range(k)], dtype=np.int64)
        iters = np.array([info[i].iter for i in range(k)], dtype=np.int64)
        return x[:k], y[:k], z[:k], status, iters

    @property
    def width(self):
        return self.n + self.n_eq + self.n_c + 2

    def gather_device(self, group=None, via_host=False):
        """The path's one collective, device-resident end to end: a device kernel packs
        (x, y, z, status, iter) of the local shard into one fp64 buffer on this rank's GPU
        (`pqp_batch_pack_results`), `all_gather_into_tensor` moves it over RCCL / xGMI, and the
        result -- a [world * per][width] ROCm tensor, shards padded to `per` rows -- stays on the
        GPU.  No host copy, no numpy.

        `via_host=True` is the rehearsal form for a box with ONE GPU shared by all ranks (RCCL refuses two
        ranks on one device): the same pack kernel, then the packed buffer crosses the process group on
        the host (gloo) and the gathered batch goes back to the device."""
        import torch
        import torch.distributed as dist
        dev = torch.device("cuda", self.device)
        per = -(-self.total // self.world)  # shards padded to equal length
        buf = torch.zeros((per, self.width), dtype=torch.float64, device=dev)
        if self.local:
            with torch.cuda.device(dev):
                self.batch.pack_results(buf, 0, self.local, stream=torch.cuda.current_stream(dev).cuda_stream)
        if via_host:
            host = torch.empty((self.world * per, self.width), dtype=torch.float64)
            dist.all_gather_into_tensor(host, buf.cpu(), group=group)
            return host.to(dev), per
        out = torch.empty((self.world * per, self.width), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(out, buf, group=group)
        return out, per

    def split(self, full, per):
        """(x, y, z, status, iter) views of a gathered [world * per][width] array in global QP order."""
        rows = []
        for r in range(self.world):
            f, l = shard_range(self.total, r, self.world)
            rows.append(full[r * per: r * per + (l - f)])
        cat = np.concatenate if isinstance(full, np.ndarray) else __import__("torch").cat
        full = cat(rows, 0)
        n, ne, nc = self.n, self.n_eq, self.n_c
        as_int = (lambda a: a.astype(np.int64)) if isinstance(full, np.ndarray) else (lambda a: a.long())
        return (full[:, :n], full[:, n:n + ne], full[:, n + ne:n + ne + nc], as_int(full[:, -2]), as_int(full[:, -1]))

    def gather(self, group=None, device="cpu"):
        """all_gather of (x, y, z, status, iter) over the process group: every rank returns the
        full-batch arrays in global QP order (numpy).  `device="cuda"`: the device-resident path
        (`gather_device`), one D2H copy of the gathered batch at the very end; `device="cpu"`: host
        buffers over gloo (CPU tests; the emulator build has no device memory)."""
        import torch
        import torch.distributed as dist
        if str(device).startswith("cuda"):
            out, per = self.gather_device(group)
            return self.split(out.cpu().numpy(), per)
        x, y, z, status, iters = self.local_results()
        per = -(-self.total // self.world)
        buf = torch.zeros((per, self.width), dtype=torch.float64)
        if self.local:
            packed = np.concatenate([x, y, z, status[:, None].astype(np.float64), iters[:, None].astype(np.float64)],
                                    axis=1)
            buf[: self.local] = torch.from_numpy(packed)
        out = torch.empty((self.world * per, self.width), dtype=torch.float64)
        dist.all_gather_into_tensor(out, buf, group=group)
        return self.split(out.numpy(), per)
