"""Shared host plumbing of the two pipelines: load the side files named on the command line and
configure one GPU context.  Host logic only; the compute is `engine.Engine` (libugvc_mi355x.so)."""
from __future__ import annotations

import logging
import os

import numpy as np

from .. import schema as S
from ..io import bed

logger = logging.getLogger("ugvc")


def load_side_tables(reference_file, runs_file, annotate_intervals, blacklist_file, also=None, on_ready=None):
    """Reference, homopolymer runs, annotation tracks, blacklist - read CONCURRENTLY (the native readers release the GIL;
    each is threaded itself, but none keeps a large host busy alone: FASTA encode, four interval files and a 100 MB BGZF are
    0.3-0.6 s one after the other).  The interval files need the contig names: with a `.fai` beside the FASTA they start at
    once, otherwise when the FASTA has been read.  `also`: {name: f(contig_names)} extra readers that need only the names
    (the callset VCF of filter_variants_pipeline) - their results come back as a dict in fifth place.  `on_ready(kind, index,
    table)` (kind "reference" | "runs" | "track" | "blacklist"; round 6): called on the reader's thread the moment a table is
    there - filter_variants_pipeline hands it to the GPU context's thread, so that the uploads run under the slowest reader
    instead of behind all of them."""
    from concurrent.futures import ThreadPoolExecutor
    from ..io import vcf_native                        # threaded native readers (libugvc_vcf.so); io.fasta / io.bed are their references
    if len(annotate_intervals or []) > S.MAX_TRACKS:
        raise ValueError(f"at most {S.MAX_TRACKS} --annotate_intervals files are supported")
    also = also or {}
    import time
    seconds = {}

    def timed(name, f, kind=None, index=0):
        def g(*a):
            t0 = time.perf_counter()
            try:
                out = f(*a)
            finally:
                seconds[name] = time.perf_counter() - t0
            if on_ready is not None and kind is not None:
                on_ready(kind, index, out)
            return out
        return g
    load_side_tables.last_seconds = seconds                  # (read by tools/bench_pipeline.py: which reader the stage waits for)
    with ThreadPoolExecutor(max_workers=8) as pool:
        f_ref = pool.submit(timed("reference", vcf_native.read_fasta, "reference"), reference_file)
        names = vcf_native.read_fasta_names(reference_file) if os.path.exists(reference_file + ".fai") else None
        if names is None:
            names = f_ref.result().names
        # homopolymer runs are disjoint by nature; book-ended runs of different bases must stay separate
        f_runs = pool.submit(timed("runs", vcf_native.read_intervals, "runs"), runs_file, names, False) if runs_file else None
        f_tracks = [pool.submit(timed(f"track {os.path.basename(p)}", vcf_native.read_intervals, "track", k), p, names, True)
                    for k, p in enumerate(annotate_intervals or [])]
        f_bl = pool.submit(timed("blacklist", bed.read_blacklist, "blacklist"), blacklist_file, names) if blacklist_file else None
        f_also = {k: pool.submit(timed(k, f), names) for k, f in also.items()}
        ref = f_ref.result()
        if list(ref.names) != list(names):
            raise ValueError(f"{reference_file}.fai does not list the contigs of {reference_file}")
        if ref.n_contigs > 65535:
            # the contig column is u16; production hg38 has 3 366 contigs
            # (test/resources/unit/vcfbed/test_vcftools/header.txt), nothing is dropped below this bound
            raise ValueError(f"{reference_file}: {ref.n_contigs} contigs; the engine indexes at most 65535")
        runs = f_runs.result() if f_runs else None
        tracks = [f.result() for f in f_tracks]
        bl = f_bl.result() if f_bl else None
        extra = {k: f.result() for k, f in f_also.items()}
    return (ref, runs, tracks, bl, extra) if also else (ref, runs, tracks, bl)


def cg_insertion_mask(vt: S.VariantTable) -> np.ndarray:
    """`--blacklist_cg_insertions` "Should CCG/GGC insertions be filtered out?"
    (docs/filter_variants_pipeline.md:36-37).  BUILDER-DEFINED reading (the body is in the absent
    submodule): an insertion whose inserted bases are exactly CCG or GGC."""
    out = np.zeros(vt.n, dtype=bool)
    ins = np.flatnonzero((vt.alt_len == vt.ref_len + 3) & (vt.ref_len == 1))
    if ins.size:
        off = vt.alt_off[ins].astype(np.int64)
        # the three inserted bases as one base-8 number (codes 0..4): CCG / GGC
        tail = (vt.alleles[off + 1].astype(np.int64) << 6) | (vt.alleles[off + 2].astype(np.int64) << 3) | vt.alleles[off + 3]
        ccg, ggc = (int(x[0]) << 6 | int(x[1]) << 3 | int(x[2]) for x in (S.encode_bases("CCG"), S.encode_bases("GGC")))
        out[ins] = (tail == ccg) | (tail == ggc)
    return out


def check_model_tracks(forests, n_tracks: int, tool: str):
    want = max((f.n_features for f in forests if f is not None), default=S.N_BASE_FEATURES)
    if want > S.N_BASE_FEATURES + n_tracks:
        raise ValueError(f"{tool}: the model was trained with {want - S.N_BASE_FEATURES} annotation track(s) "
                         f"(--annotate_intervals), {n_tracks} given")
