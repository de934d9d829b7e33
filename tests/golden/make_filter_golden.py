"""Mint tests/golden/filter_golden_v1.npz: oracle outputs on the edge-case table over real hg38
(chr1:1-5,000,000 + chr20 100 kb) with the frozen synthetic RF model.  Guards both the oracle
(against drift) and the kernel.  Usage: python tests/golden/make_filter_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
import edge_cases as E  # noqa: E402
from conftest import real_chr1_reference  # noqa: E402
from oracle import oracle as O  # noqa: E402
from variantcalling_amd import model_io  # noqa: E402

ref = real_chr1_reference()
vt = E.edge_table(ref)
runs, tracks = E.simple_tracks(ref)
bl = np.unique(np.concatenate([vt.keys()[::7], vt.keys()[::11] + np.uint64(1)]))
forests = model_io.load_models(os.path.join(HERE, "synth_rf_v1.npz"))["rf_model_ignore_gt_incl_hpol_runs"]
res = O.filter_variants(vt, ref, runs, tracks, bl, forests, hpol_len=8, hpol_dist=12)
X = O.featurize(vt, ref, runs, tracks, "TGCA", 8, 12)["X"]
dst = os.path.join(HERE, "filter_golden_v1.npz")
np.savez_compressed(dst, tree_score=res.tree_score, filter=res.filter, flags=res.flags, X=X,
                    pos=vt.pos, contig=vt.contig)
print(dst, os.path.getsize(dst), vt.n)
