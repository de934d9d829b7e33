"""Multi-GPU partitioning of a sorted callset.

Every variant is independent given read-only side tables, so ranks take contiguous,
equal-count (+-1) slices of the sorted variant list; rank-order concatenation of the result
columns is callset order, so the RCCL all-gather needs no permutation.  The reference
parallelises the neighbouring steps per contig (`--n_jobs`, docs/run_comparison_pipeline.md:81;
HDF5 keyed per chromosome, docs/train_models_pipeline.md:58-59); equal counts balance better
than contigs on 8 GPUs (chr1 is 8 % of the genome, chr21 1.5 %).
"""
from __future__ import annotations

import numpy as np

from . import schema as S


def shard_bounds(n: int, world: int) -> np.ndarray:
    """world+1 row boundaries; shard r is rows [b[r], b[r+1]); sizes differ by at most 1."""
    if world < 1:
        raise ValueError("world must be >= 1")
    base, extra = divmod(n, world)
    sizes = np.full(world, base, dtype=np.int64)
    sizes[:extra] += 1
    return np.concatenate([[0], np.cumsum(sizes)])


def shard_cap(n: int, world: int) -> int:
    """Padded shard length used as the all-gather count (max shard size)."""
    return int(-(-n // world))


def shard_of(vt: S.VariantTable, rank: int, world: int) -> S.VariantTable:
    b = shard_bounds(vt.n, world)
    return vt.slice(int(b[rank]), int(b[rank + 1]))


def reassemble(parts: list, counts: list) -> S.FilterResult:
    """Concatenate padded per-rank result columns (what the all-gather produces) in rank order."""
    ts = np.concatenate([p.tree_score[:c] for p, c in zip(parts, counts)])
    fl = np.concatenate([p.filter[:c] for p, c in zip(parts, counts)])
    fg = np.concatenate([p.flags[:c] for p, c in zip(parts, counts)])
    return S.FilterResult(ts, fl, fg)


def pad_result(r: S.FilterResult, cap: int) -> S.FilterResult:
    def pad(a):
        out = np.zeros(cap, dtype=a.dtype)
        out[: a.size] = a
        return out
    return S.FilterResult(pad(r.tree_score), pad(r.filter), pad(r.flags))
