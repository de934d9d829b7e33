"""Process-group plumbing for one-process-per-GPU runs.

torch.distributed (gloo) is used ONLY for rendezvous: sharing the RCCL unique id, barriers and
the max-over-ranks timing reduction.  The data-path collective (all-gather of the scored
shards) is RCCL inside libugvc_mi355x.so (csrc/comm.hip); on CPU-only test runs the same
reassembly is exercised with a gloo all_gather of padded columns.
Environment: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as set by
`python -m torch.distributed.run`.
"""
from __future__ import annotations

import os

import numpy as np

from . import schema as S
from . import shard


class Group:
    """world == 1: no torch import at all."""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self._dist = None
        if self.world > 1:
            import datetime

            import torch.distributed as dist
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world,
                                        timeout=datetime.timedelta(minutes=20))
            self._dist = dist

    def barrier(self):
        if self._dist is not None:
            self._dist.barrier()

    def broadcast_bytes(self, payload: bytes | None, src: int = 0) -> bytes:
        if self._dist is None:
            return payload
        obj = [payload if self.rank == src else None]
        self._dist.broadcast_object_list(obj, src=src)
        return obj[0]

    def max_float(self, x: float) -> float:
        if self._dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t[0])

    def sum_float(self, x: float) -> float:
        if self._dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return float(t[0])

    def allgather_results_host(self, local: S.FilterResult, n_total: int) -> S.FilterResult:
        """Host-side (gloo) equivalent of the RCCL all-gather: padded equal-size columns, rank
        order concatenation.  Used by the CPU tests of the sharding logic."""
        if self._dist is None:
            return local
        import torch
        cap = shard.shard_cap(n_total, self.world)
        b = shard.shard_bounds(n_total, self.world)
        counts = [int(b[r + 1] - b[r]) for r in range(self.world)]
        padded = shard.pad_result(local, cap)
        parts = []
        for col in (padded.tree_score, padded.filter, padded.flags):
            t = torch.from_numpy(np.ascontiguousarray(col))
            outs = [torch.empty_like(t) for _ in range(self.world)]
            self._dist.all_gather(outs, t)
            parts.append([o.numpy() for o in outs])
        per_rank = [S.FilterResult(parts[0][r], parts[1][r], parts[2][r]) for r in range(self.world)]
        return shard.reassemble(per_rank, counts)

    def close(self):
        if self._dist is not None and self._dist.is_initialized():
            self._dist.destroy_process_group()
            self._dist = None
