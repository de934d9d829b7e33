"""Mint tests/golden/reference_run_v1.npz by RUNNING the reference's own in-tree code here.

Runs only in the build container (needs /root/reference).  Three pieces of the hot path have
their body in-tree and are importable once the I/O libraries
they pull in (pysam, simppl - not installed, no network) are replaced by inert stand-ins:

  * ugvc/pipelines/vcfbed/calibrate_bridging_snvs.py  `run(argv)` end to end: the stand-in
    `pysam.VariantFile` yields record objects built from our SoA edge-case table over the REAL
    hg38 slice, `pysam.FastaFile.fetch` slices that sequence, and the output "file" records
    which records the reference un-filtered (PASS added, QUAL set).  -> pins oracle/bridging.py
    and the ugvc_bridging_snvs kernel.
  * ugvc/utils/stats_utils.py  multinomial_likelihood[_ratio] on random count vectors
    -> pins oracle/stats.py and the ugvc_sec_likelihood_ratio kernel;
    precision_recall_curve (FN-aware) on random score/label sets -> pins oracle/evaluate.py.
  * ugvc/utils/math_utils.py  phred / unphred.

Nothing of the reference is copied: this script only calls it and stores inputs + outputs.
Usage: python tests/golden/make_reference_goldens.py
"""
import argparse
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, "/root/reference")

import edge_cases as E  # noqa: E402
from conftest import real_chr1_reference  # noqa: E402
from variantcalling_amd import schema as S  # noqa: E402

# ---- inert stand-ins for the I/O libraries the reference module imports -------------------
_state = {}


class _Filter:
    def __init__(self, keys):
        self._k = list(keys)

    def keys(self):
        return list(self._k)

    def add(self, k):
        self._k.append(k)


class _Record:
    def __init__(self, idx, chrom, pos, ref, alt, is_pass, qual, ad, dp, bg_ad, bg_dp):
        self.idx, self.chrom, self.pos, self.ref, self.alts = idx, chrom, pos, ref, (alt,)
        self.filter = _Filter(["PASS"] if is_pass else ["RefCall"])
        self.qual = qual
        self.samples = [dict(AD=ad, DP=dp, BG_AD=bg_ad, BG_DP=bg_dp)]


class _VariantFile:
    def __init__(self, path, mode="r", header=None):
        self.mode, self.header = mode, header or "header"
        if mode == "w":
            _state["written"] = []

    def __iter__(self):
        return iter(_state["records"])

    def write(self, rec):
        _state["written"].append(rec)

    def close(self):
        pass


class _FastaFile:
    def __init__(self, path):
        pass

    def fetch(self, contig, start, end):
        # pysam: 0-based half-open; an end past the contig is clipped; a negative start raises
        # (records that would trigger it are excluded from the golden set - the oracle's
        # behaviour there is BUILDER-DEFINED and tested separately).
        assert start >= 0
        return _state["seqs"][contig][start:end]


pysam = types.ModuleType("pysam")
pysam.VariantFile, pysam.FastaFile = _VariantFile, _FastaFile
pysam.tabix_index = lambda *a, **k: None
sys.modules["pysam"] = pysam
simppl = types.ModuleType("simppl")
simppl_cli = types.ModuleType("simppl.cli")
simppl_cli.get_parser = lambda name, doc: argparse.ArgumentParser(prog=name, description=doc)
simppl.cli = simppl_cli
sys.modules["simppl"], sys.modules["simppl.cli"] = simppl, simppl_cli

import logging  # noqa: E402

from ugvc.pipelines.vcfbed import calibrate_bridging_snvs as ref_cbs  # noqa: E402
from ugvc.utils import math_utils as ref_math  # noqa: E402
from ugvc.utils import stats_utils as ref_stats  # noqa: E402

logging.getLogger("ugvc").setLevel(logging.ERROR)

out = {}

# ---- a12: calibrate_bridging_snvs.run on the real hg38 slice ------------------------------
ref = real_chr1_reference()
vt = E.edge_table(ref, seed=9, n_random=6000)
rng = np.random.default_rng(4)
n = vt.n
is_pass = rng.random(n) < 0.3
ad_alt = rng.integers(0, 40, n).astype(np.int32)
bg_ad = rng.integers(0, 6, n).astype(np.int32)
bg_dp = rng.integers(0, 40, n).astype(np.int32)
table = np.frombuffer(b"NACGT", dtype=np.uint8)
_state["seqs"] = {name: table[ref.codes[ref.contig_off[c]: ref.contig_off[c + 1]]].tobytes().decode()
                  for c, name in enumerate(ref.names)}
H_MAX = 5
usable = (vt.pos.astype(np.int64) - H_MAX - 1 >= 0) & (vt.dp > 0)     # pysam would raise / ZeroDivisionError
records = []
for i in np.flatnonzero(usable):
    r = S.decode_bases(vt.alleles[vt.ref_off[i]: vt.ref_off[i] + vt.ref_len[i]])
    a = S.decode_bases(vt.alleles[vt.alt_off[i]: vt.alt_off[i] + vt.alt_len[i]])
    records.append(_Record(int(i), ref.names[int(vt.contig[i])], int(vt.pos[i]), r, a, bool(is_pass[i]),
                           float(vt.qual[i]), (int(vt.ad_ref[i]), int(ad_alt[i])), int(vt.dp[i]),
                           (0, int(bg_ad[i])), int(bg_dp[i])))
combos = [(2, 0), (3, 0), (5, 1), (4, 0)]
out["bridging_usable"] = usable
out["bridging_combos"] = np.array(combos, dtype=np.int32)
out["bridging_is_pass"], out["bridging_ad_alt"] = is_pass, ad_alt
out["bridging_bg_ad"], out["bridging_bg_dp"] = bg_ad, bg_dp
for h, edge in combos:
    for rec in records:                      # fresh filter/qual state for every run
        rec.filter = _Filter(["PASS"] if is_pass[rec.idx] else ["RefCall"])
        rec.qual = float(vt.qual[rec.idx])
    _state["records"] = records
    ref_cbs.run(["calibrate_bridging_snvs", "--vcf", "in.vcf", "--reference", "ref.fa", "--output", "out.vcf",
                 "--min_query_hmer_size", str(h), "--min_distance_from_edge", str(edge)])
    unf = np.zeros(n, dtype=bool)
    hm = np.zeros(n, dtype=bool)
    for rec in _state["written"]:
        unf[rec.idx] = (not is_pass[rec.idx]) and "PASS" in rec.filter.keys() and rec.qual == 20
        hm[rec.idx] = ref_cbs.is_homopolymer_snp(
            _Record(rec.idx, rec.chrom, rec.pos, rec.ref, rec.alts[0], bool(is_pass[rec.idx]),
                    float(vt.qual[rec.idx]), None, None, None, None),
            _FastaFile(""), h, 5, edge)[0]
    out[f"bridging_hm_{h}_{edge}"] = hm
    out[f"bridging_unfiltered_{h}_{edge}"] = unf
    print(f"bridging h={h} edge={edge}: records {len(records)}, hmer SNPs {int(hm.sum())}, un-filtered {int(unf.sum())}")

# ---- a8: SEC statistic on random count vectors ---------------------------------------------
rng = np.random.default_rng(3)
A = rng.integers(0, 60, size=(2000, 5)).astype(np.int32)
Ex = rng.integers(0, 400, size=(2000, 5)).astype(np.int32)
A[:50] = 0                                 # all-zero observations
Ex[25:75] = 0                              # all-zero expectations
lik = np.zeros(A.shape[0])
ratio = np.zeros(A.shape[0])
for i in range(A.shape[0]):
    lik[i], ratio[i] = ref_stats.multinomial_likelihood_ratio(list(A[i]), list(Ex[i]))
out["sec_actual"], out["sec_expected"], out["sec_lik"], out["sec_ratio"] = A, Ex, lik, ratio
tabs = [rng.integers(0, 50, size=int(rng.integers(2, 6))).tolist() for _ in range(200)]
ns = rng.integers(1, 300, size=200)
out["scale_tables"] = np.array([t + [-1] * (5 - len(t)) for t in tabs], dtype=np.int64)
out["scale_n"] = ns
out["scale_out"] = np.array([list(map(int, ref_stats.scale_contingency_table(t, int(k)))) + [-1] * (5 - len(t))
                             for t, k in zip(tabs, ns)], dtype=np.int64)

# ---- a13: FN-aware precision/recall curve ---------------------------------------------------
for case, (m, fn_frac, min_cls) in enumerate([(5000, 0.05, 20), (300, 0.2, 1), (40, 0.0, 20), (2000, 0.5, 100)]):
    labels = (rng.random(m) < 0.7).astype(np.int64)
    scores = np.round(np.clip(rng.normal(0.4 + 0.3 * labels, 0.2), 0, 1), 3)     # ties on purpose
    fn = (rng.random(m) < fn_frac) & (labels == 1)
    scores[fn] = -1
    p, r, f1, thr = ref_stats.precision_recall_curve(labels, scores, fn, pos_label=1, min_class_counts_to_output=min_cls)
    out[f"pr{case}_labels"], out[f"pr{case}_scores"], out[f"pr{case}_fn"] = labels, scores, fn
    out[f"pr{case}_min_cls"] = np.int64(min_cls)
    out[f"pr{case}_precision"], out[f"pr{case}_recall"], out[f"pr{case}_f1"], out[f"pr{case}_thr"] = p, r, f1, thr
    print(f"pr case {case}: {m} calls -> {p.size} curve points, max f1 {f1.max() if f1.size else float('nan'):.4f}")
fpv = rng.integers(0, 1000, size=100)
tpv = rng.integers(0, 1000, size=100)
fpv[:3] = 0
tpv[:3] = 0
out["prec_fp"], out["prec_tp"] = fpv, tpv
out["prec_out"] = np.array([ref_stats.get_precision(int(a), int(b)) for a, b in zip(fpv, tpv)])
out["rec_out"] = np.array([ref_stats.get_recall(int(a), int(b)) for a, b in zip(fpv, tpv)])
out["f1_out"] = np.array([ref_stats.get_f1(float(a), float(b)) for a, b in zip(out["prec_out"], out["rec_out"])])

# ---- math_utils -----------------------------------------------------------------------------
q = rng.random(64) * 60
out["phred_in"] = 10 ** (-q / 10)
out["phred_out"] = ref_math.phred(out["phred_in"])
out["unphred_in"] = q
out["unphred_out"] = ref_math.unphred(list(q))

dst = os.path.join(HERE, "reference_run_v1.npz")
np.savez_compressed(dst, **out)
print(dst, os.path.getsize(dst))
