"""BED / Picard interval_list -> `schema.IntervalTrack`; blacklist files -> sorted u64 locus keys.

`--runs_file`, `--annotate_intervals` (docs/filter_variants_pipeline.md:30-33,45-46), `--runs_intervals`
"bed/interval_list" (docs/train_models_pipeline.md:62-63), `--blacklist` (docs/filter_variants_pipeline.md:34-35).
Intervals are sorted per contig and overlapping / nested ones merged, so starts and ends are both
non-decreasing - the precondition the engine checks at upload (include/ugvc_mi355x.h).  Host logic only."""
from __future__ import annotations

import gzip
import os
import pickle

import numpy as np

from ..schema import IntervalTrack


def _open_text(path: str):
    with open(path, "rb") as fh:
        magic = fh.read(2)
    return gzip.open(path, "rt") if magic == b"\x1f\x8b" else open(path, "rt")


def track_from_arrays(contig: np.ndarray, starts: np.ndarray, ends: np.ndarray, n_contigs: int,
                      name: str = "", merge: bool = True) -> IntervalTrack:
    contig, starts, ends = np.asarray(contig), np.asarray(starts, dtype=np.int64), np.asarray(ends, dtype=np.int64)
    # (a BED file is sorted as a rule: one vectorised check instead of a three-key sort of 3 M rows - 0.25 s of the tool's first stage)
    if contig.size > 1:
        # (comparisons, not differences: a decreasing UNSIGNED contig column - the variant table's is uint16 - wraps to a large
        # positive difference and would pass for sorted)
        c0, c1, s0, s1 = contig[:-1], contig[1:], starts[:-1], starts[1:]
        in_order = bool(np.all((c1 > c0) | ((c1 == c0) & ((s1 > s0) | ((s1 == s0) & (ends[1:] >= ends[:-1]))))))
    else:
        in_order = True
    if not in_order:
        order = np.lexsort((ends, starts, contig))
        contig, starts, ends = contig[order], starts[order], ends[order]
    keep = ends > starts
    if not keep.all():
        contig, starts, ends = contig[keep], starts[keep], ends[keep]
    if merge and starts.size:
        # running maximum of the ends inside each contig; a new merged interval opens where a start
        # lies beyond everything seen so far (bedtools-merge semantics, book-ended intervals stay apart)
        big = np.int64(1) << 40
        cm = np.maximum.accumulate(ends + contig.astype(np.int64) * big) - contig.astype(np.int64) * big
        new = np.ones(starts.size, dtype=bool)
        new[1:] = (contig[1:] != contig[:-1]) | (starts[1:] > cm[:-1])
        first = np.flatnonzero(new)                              # (the groups are runs of rows: a segmented maximum)
        contig, starts, ends = contig[new], starts[new], np.maximum.reduceat(ends, first)
    ptr = np.searchsorted(contig, np.arange(n_contigs + 1)).astype(np.int32)
    return IntervalTrack(starts.astype(np.int32), ends.astype(np.int32), ptr, name)


def read_intervals(path: str, contig_names: list, merge: bool = True) -> IntervalTrack:
    """BED (0-based half-open) or interval_list (`@` header, 1-based inclusive)."""
    idx = {n: i for i, n in enumerate(contig_names)}
    c, s, e = [], [], []
    one_based = path.endswith(".interval_list")
    with _open_text(path) as fh:
        for line in fh:
            if not line.strip() or line[0] in "#@" or line.startswith(("track", "browser")):
                if line.startswith("@"):
                    one_based = True
                continue
            f = line.rstrip("\n").split("\t")
            if len(f) < 3:
                f = line.split()
            if f[0] not in idx:
                continue                   # contig not in the reference: cannot annotate any variant
            c.append(idx[f[0]])
            s.append(int(f[1]) - (1 if one_based else 0))
            e.append(int(f[2]))
    stem = os.path.basename(path)
    for suf in (".gz", ".bed", ".interval_list"):
        if stem.endswith(suf):
            stem = stem[: -len(suf)]
    return track_from_arrays(np.array(c, dtype=np.int64), np.array(s, dtype=np.int64), np.array(e, dtype=np.int64),
                             len(contig_names), stem, merge)


def write_bed(path: str, track: IntervalTrack, contig_names: list) -> None:
    with open(path, "w") as fh:
        for c, name in enumerate(contig_names):
            for i in range(int(track.contig_ptr[c]), int(track.contig_ptr[c + 1])):
                fh.write(f"{name}\t{int(track.starts[i])}\t{int(track.ends[i])}\n")


def keys_from_loci(contig_idx, pos) -> np.ndarray:
    k = (np.asarray(contig_idx, dtype=np.uint64) << np.uint64(32)) | np.asarray(pos, dtype=np.uint64)
    return np.unique(k)


def read_blacklist(path: str, contig_names: list) -> np.ndarray:
    """Cohort false-positive loci as sorted unique keys contig<<32|pos.  Accepted: .npz/.npy of keys or
    (contig, pos); .bed (every 1-based position start+1..end); .pkl holding an iterable of (chrom, pos),
    a dict of such iterables, or a pandas DataFrame / index with chrom and pos; .h5 / .hdf (`--blacklist
    cohort_fp.h5`, docs/train_models_pipeline.md:88): every pandas frame / series of the file that carries loci as a
    (chrom, pos) MultiIndex or as chrom and pos columns (io/h5.py).  The reference's own blacklist pickle holds objects
    of a class that lives in the absent submodule: that is reported, not guessed."""
    idx = {n: i for i, n in enumerate(contig_names)}
    if path.endswith(".npy"):
        return np.unique(np.load(path).astype(np.uint64))
    if path.endswith(".npz"):
        z = np.load(path)
        if "keys" in z:
            return np.unique(z["keys"].astype(np.uint64))
        return keys_from_loci(z["contig"], z["pos"])
    if path.endswith((".bed", ".bed.gz")):
        t = read_intervals(path, contig_names, merge=True)
        out = []
        for c in range(len(contig_names)):
            for i in range(int(t.contig_ptr[c]), int(t.contig_ptr[c + 1])):
                p = np.arange(int(t.starts[i]) + 1, int(t.ends[i]) + 1, dtype=np.uint64)
                out.append((np.uint64(c) << np.uint64(32)) | p)
        return np.unique(np.concatenate(out)) if out else np.zeros(0, np.uint64)
    if path.endswith((".pkl", ".pickle")):
        try:
            with open(path, "rb") as fh:
                obj = pickle.load(fh)
        except (ImportError, AttributeError):
            # the reference's blacklist pickle holds objects of an absent class: read its data (legacy_pickle.py)
            from .. import legacy_pickle
            got = legacy_pickle.find_loci(legacy_pickle.load(path))
            if not got:
                raise ValueError(f"{path}: no (chrom, pos) loci found in the pickle")
            got = [(idx[str(c)], int(p_)) for c, p_ in got if str(c) in idx]
            if not got:
                return np.zeros(0, np.uint64)
            a = np.array(got, dtype=np.int64)
            return keys_from_loci(a[:, 0], a[:, 1])
        loci = []

        def walk(o):
            if hasattr(o, "columns") and "chrom" in getattr(o, "columns", []) and "pos" in o.columns:
                loci.extend(zip(o["chrom"], o["pos"]))
            elif hasattr(o, "index") and hasattr(o.index, "names") and list(o.index.names)[:2] == ["chrom", "pos"]:
                loci.extend(o.index.tolist())
            elif isinstance(o, dict):
                for v in o.values():
                    walk(v)
            elif isinstance(o, (list, tuple, set, frozenset)):
                for v in o:
                    if isinstance(v, tuple) and len(v) >= 2 and isinstance(v[0], str):
                        loci.append(v[:2])
                    else:
                        walk(v)
            else:
                raise ValueError(f"{path}: cannot interpret a {type(o).__name__} as blacklist loci")
        walk(obj)
        loci = [(idx[c], int(p)) for c, p in loci if c in idx]
        if not loci:
            return np.zeros(0, np.uint64)
        a = np.array(loci, dtype=np.int64)
        return keys_from_loci(a[:, 0], a[:, 1])
    if path.endswith((".h5", ".hdf", ".hdf5")):
        from . import h5
        chroms, poss = [], []
        with h5.H5File(path) as f:
            keys = [k for k in f.keys() if f[k].attrs.get("pandas_type") in ("frame", "series")]
        for k in keys:
            fr = h5.read_hdf(path, k)
            if "chrom" in fr and "pos" in fr:
                chroms.append(np.asarray(fr["chrom"], dtype=object)); poss.append(np.asarray(fr["pos"]))
            elif isinstance(fr.index, list) and len(fr.index) >= 2 and list(fr.index_names or [])[:2] in (["chrom", "pos"], [None, None]):
                chroms.append(np.asarray(fr.index[0], dtype=object)); poss.append(np.asarray(fr.index[1]))
        if not chroms:
            raise ValueError(f"{path}: no frame or series with (chrom, pos) loci among the keys {keys}")
        c = np.concatenate(chroms)
        p_ = np.concatenate(poss).astype(np.int64)
        known = np.array([x in idx for x in c], dtype=bool)
        if not known.any():
            return np.zeros(0, np.uint64)
        return keys_from_loci(np.array([idx[x] for x in c[known]], dtype=np.int64), p_[known])
    raise ValueError(f"{path}: unsupported blacklist format (use .h5 / .pkl of (chrom, pos) loci, .bed, .npy or .npz)")
