"""Debug aid: repeated ugvc_filter_variants calls on the 5 M callset; where do two calls differ, and which one is right."""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from variantcalling_amd import model_io, synth  # noqa: E402
from variantcalling_amd.engine import Engine, configure  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
cs = synth.make_callset(n)
forests = model_io.load_models(os.path.join(ROOT, "tests", "golden", "synth_rf_v1.npz"))["rf_model_ignore_gt_incl_hpol_runs"]
with Engine(0) as eng:
    # smaller callsets first, as the test session does (scratch buffers grow from call to call)
    small = synth.make_callset(60_000, genome_len=30_000_000, n_contigs=5, seed=1234)
    configure(eng, small.ref, small.runs, small.tracks, small.blacklist, forests)
    eng.filter_variants(small.variants)
    configure(eng, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
    runs = [eng.filter_variants(cs.variants) for _ in range(4)]
for k in range(1, 4):
    for what in ("tree_score", "filter", "flags"):
        a, b = getattr(runs[0], what), getattr(runs[k], what)
        bad = np.flatnonzero(a != b)
        if bad.size:
            print(f"call 0 vs call {k}: {bad.size} {what} differ, rows {bad[:12]} ... {bad[-3:]}")
bad = np.flatnonzero(runs[0].tree_score != runs[1].tree_score)
if bad.size:
    lo, hi = int(bad.min()), int(bad.max()) + 1
    hi = min(hi, lo + 200_000)
    exp = O.filter_variants(cs.variants.slice(lo, hi), cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
    for k in range(4):
        w = np.flatnonzero(runs[k].tree_score[lo:hi] != exp.tree_score)
        print(f"call {k}: {w.size} rows of [{lo}, {hi}) differ from the oracle; first {w[:8] + lo}; indel? {(cs.variants.ref_len[lo:hi][w[:8]] != cs.variants.alt_len[lo:hi][w[:8]])}")
else:
    print("all four calls agree")
