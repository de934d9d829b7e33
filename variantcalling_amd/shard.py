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


def shard_bounds(n: int, world: int, contig: np.ndarray | None = None, tol: float = 0.01) -> np.ndarray:
    """world+1 row boundaries; shard r is rows [b[r], b[r+1]); sizes differ by at most 1.

    With the callset's (sorted) `contig` column an interior cut is SNAPPED to a contig change when one lies close enough that
    no shard grows beyond (1 + tol) of the equal share ("snap shard cuts to contig / interval boundaries when
    that costs < 1 % imbalance"): the rank behind the cut then starts on a contig's first row, the rank before it ends on a
    contig's last, and neither carries a contig boundary in the middle of a workgroup's rows there.  (On a 24-contig genome
    cut 8 ways a cut finds such a neighbour about once in twenty: the equal-count cut stays the rule, and every rank must
    cope with a boundary anywhere in its shard - tests/test_shard_slices.py.)  Every rank computes the same bounds from the
    same column; a rank that holds only its slice of the records (the tool's part reader) uses the plain equal-count cut."""
    if world < 1:
        raise ValueError("world must be >= 1")
    base, extra = divmod(n, world)
    sizes = np.full(world, base, dtype=np.int64)
    sizes[:extra] += 1
    b = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    if contig is None or world == 1 or n == 0:
        return b
    contig = np.asarray(contig)
    if contig.shape != (n,):
        raise ValueError("contig must have one entry per row")
    change = np.flatnonzero(contig[1:] != contig[:-1]) + 1           # first rows of the 2nd, 3rd, ... contig
    if change.size == 0:
        return b
    share = n / world
    slack = int(tol * share)                                          # rows a shard may grow by
    out = b.copy()
    for r in range(1, world):
        k = int(np.searchsorted(change, b[r]))
        cand = [int(change[j]) for j in (k - 1, k) if 0 <= j < change.size]
        cand = [c for c in cand if abs(c - int(b[r])) <= slack and out[r - 1] < c < b[r + 1]]
        if not cand:
            continue
        c = min(cand, key=lambda x: abs(x - int(b[r])))
        # both neighbours stay within (1 + tol) of the equal share (the previous cut may have moved already)
        if c - out[r - 1] <= share * (1 + tol) + 1 and b[r + 1] - c <= share * (1 + tol) + 1:
            out[r] = c
    return out


def shard_cap(n: int, world: int, bounds: np.ndarray | None = None) -> int:
    """Padded shard length used as the all-gather count: the largest shard, rounded up to 256 rows - every rank's slot of the
    three gather buffers (f32, u8, u8 at rank * cap elements) then starts on a 256-byte boundary, whatever RCCL's copy kernels
    prefer, and the scoring pass writes its result columns to aligned bases.  `bounds`: snapped cuts (shard_bounds with the
    contig column) - the largest shard is then read off them."""
    largest = int(-(-n // world)) if bounds is None else int(np.max(np.diff(np.asarray(bounds))))
    return int(-(-max(largest, 1) // 256) * 256)


def shard_of(vt: S.VariantTable, rank: int, world: int, snap: bool = False) -> S.VariantTable:
    b = shard_bounds(vt.n, world, vt.contig if snap else None)
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


def _slice_track(tr: S.IntervalTrack, lo: np.ndarray, hi: np.ndarray, shift: np.ndarray, touched: np.ndarray, pad: int) -> S.IntervalTrack:
    """Rows of a per-contig sorted interval table that can matter to positions [lo[c], hi[c]) of every touched contig
    (overlap within `pad`, plus one row of halo on either side), shifted into the sliced coordinates."""
    n_contigs = tr.contig_ptr.size - 1
    starts, ends, ptr = [], [], [0]
    for c in range(n_contigs):
        a, b = int(tr.contig_ptr[c]), int(tr.contig_ptr[c + 1])
        kept = 0
        if touched[c] and b > a:
            s, e = tr.starts[a:b], tr.ends[a:b]
            first = int(np.searchsorted(e, lo[c] - pad, side="left"))          # ends ascend per contig (checked at upload)
            last = int(np.searchsorted(s, hi[c] + pad, side="right"))
            first, last = max(first - 1, 0), min(last + 1, b - a)
            if last > first:
                starts.append(s[first:last].astype(np.int64) - shift[c])
                ends.append(e[first:last].astype(np.int64) - shift[c])
                kept = last - first
        ptr.append(ptr[-1] + kept)
    st = np.concatenate(starts).astype(np.int32) if starts else np.zeros(0, np.int32)
    en = np.concatenate(ends).astype(np.int32) if ends else np.zeros(0, np.int32)
    return S.IntervalTrack(st, en, np.asarray(ptr, np.int32), tr.name)


def _run_end(seq: np.ndarray, b: int, clen: int, block: int = 1 << 16) -> int:
    """First index >= b whose base differs from seq[b - 1] (the end of the run a cut at b falls into), vectorised: a cut
    inside a multi-megabase N run or a centromere costs a few block compares, not a Python loop per base."""
    if b <= 0 or b >= clen:
        return min(max(b, 0), clen)
    base = seq[b - 1]
    while b < clen:
        chunk = seq[b: b + block]
        diff = np.flatnonzero(chunk != base)
        if diff.size:
            return b + int(diff[0])
        b += chunk.size
    return clen


def slice_context(ref: S.Reference, runs, tracks: list, blacklist, mine: S.VariantTable, margin: int = 64, pad: int = 128,
                  hpol_dist: int | None = None):
    """What ONE rank needs of the resident tables to score its shard `mine`: per touched contig the
    reference bases [min_pos - margin, max_pos + longest allele + margin) - extended to the end of the homopolymer run
    the cut would fall into, so every run a variant of the shard can see is whole (every feature looks FORWARD from a
    call for its run; backwards only the 5-base motif and the GC window, which the margin covers) - the overlapping part of each interval
    table (+ one row of halo) and the blacklist keys inside, all shifted into the sliced contigs' coordinates; untouched
    contigs keep their index with length 0.  Returns (ref, runs, tracks, blacklist, variants) with results identical to
    scoring `mine` against the full tables (tests/test_host_logic.py on the oracle, tests/test_gpu_parity.py on the GPU).
    A 3.1 Gb genome becomes ~0.4 Gb per rank at 8 ranks."""
    if hpol_dist is not None:
        # `--hpol_filter_length_dist L D`: a run D bases beyond the shard's last call still marks it - the table pad
        # follows the distance the engine is configured with instead of assuming it is small
        pad = max(pad, int(hpol_dist) + 2)
    n_contigs = ref.n_contigs
    lo = np.zeros(n_contigs, np.int64)
    hi = np.zeros(n_contigs, np.int64)
    touched = np.zeros(n_contigs, bool)
    if mine.n:
        c = mine.contig.astype(np.int64)
        first = np.flatnonzero(np.r_[True, c[1:] != c[:-1]])
        last = np.r_[first[1:], c.size] - 1
        reach = mine.pos.astype(np.int64) + np.maximum(mine.ref_len, mine.alt_len).astype(np.int64)
        for f, l in zip(first, last):
            cc = int(c[f])
            clen = ref.contig_len(cc)
            a = max(int(mine.pos[f]) - 1 - margin, 0)
            b = min(int(reach[f:l + 1].max()) + margin, clen)
            seq = ref.codes[int(ref.contig_off[cc]): int(ref.contig_off[cc + 1])]
            b = _run_end(seq, b, clen)                                             # finish the run the cut falls into
            b = min(b + margin, clen)                                              # ... and keep the motif behind it
            lo[cc], hi[cc], touched[cc] = a, b, True
    parts = [ref.codes[int(ref.contig_off[cc]) + int(lo[cc]): int(ref.contig_off[cc]) + int(hi[cc])] for cc in range(n_contigs)]
    off = np.concatenate([[0], np.cumsum([p.size for p in parts])]).astype(np.int64)
    ref_s = S.Reference(np.concatenate(parts) if parts else np.zeros(0, np.uint8), off, list(ref.names))
    shift = lo
    runs_s = _slice_track(runs, lo, hi, shift, touched, pad) if runs is not None else None
    tracks_s = [_slice_track(t, lo, hi, shift, touched, pad) for t in (tracks or [])]
    bl_s = None
    if blacklist is not None:
        kc = (blacklist >> np.uint64(32)).astype(np.int64)
        kp = (blacklist & np.uint64(0xFFFFFFFF)).astype(np.int64)
        inside = (kc < n_contigs)
        kcc = np.minimum(kc, n_contigs - 1)
        inside &= touched[kcc] & (kp > lo[kcc]) & (kp <= hi[kcc])
        bl_s = ((kc[inside].astype(np.uint64) << np.uint64(32)) | (kp[inside] - lo[kc[inside]]).astype(np.uint64)).astype(np.uint64)
    kw = {col: np.ascontiguousarray(getattr(mine, col)) for col in mine.COLS}
    kw["pos"] = (mine.pos.astype(np.int64) - lo[mine.contig.astype(np.int64)]).astype(np.int32)
    mine_s = S.VariantTable(alleles=mine.alleles, **kw)
    return ref_s, runs_s, tracks_s, bl_s, mine_s
