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
        from . import _native
        self.total, self.rank, self.world = int(total), int(rank), int(world)
        self.first, self.last = shard_range(total, rank, world)
        self.local = self.last - self.first
        self.batch = _native.Batch(max(self.local, 1), n, n_eq, n_in, device=rank if device is None else device,
                                   lib=lib, **kw)
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

    def gather(self, group=None, device="cpu"):
        """all_gather of (x, y, z, status, iter) over the process group: every rank returns the
        full-batch arrays in global QP order.  One fused fp64 buffer per rank -> one collective."""
        import torch
        import torch.distributed as dist
        x, y, z, status, iters = self.local_results()
        width = self.n + self.n_eq + self.n_c + 2
        per = -(-self.total // self.world)  # shards padded to equal length for all_gather
        buf = torch.zeros((per, width), dtype=torch.float64, device=device)
        if self.local:
            packed = np.concatenate([x, y, z, status[:, None].astype(np.float64), iters[:, None].astype(np.float64)],
                                    axis=1)
            buf[: self.local] = torch.from_numpy(packed).to(device)
        out = [torch.empty_like(buf) for _ in range(self.world)]
        dist.all_gather(out, buf, group=group)
        rows = []
        for r, t in enumerate(out):
            f, l = shard_range(self.total, r, self.world)
            rows.append(t[: l - f].cpu().numpy())
        full = np.concatenate(rows, axis=0)
        n, ne, nc = self.n, self.n_eq, self.n_c
        return (full[:, :n], full[:, n:n + ne], full[:, n + ne:n + ne + nc],
                full[:, -2].astype(np.int64), full[:, -1].astype(np.int64))
