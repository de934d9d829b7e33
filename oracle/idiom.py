"""CPU ORACLE (reference idiom, baseline B0) - TEST INFRASTRUCTURE ONLY.

The same featurize -> lookup -> score -> FILTER path as oracle.py, written the way the
reference writes it: a pandas DataFrame of Python objects, per-row `DataFrame.apply` with
string slicing on a Python reference sequence, set lookups, scikit-learn `predict_proba`,
per-record output loop.  Patterns followed: VCF->frame
(/root/reference/ugvc/reports/report_wo_gt.ipynb:1207-1210), per-record reference fetch and
run counting (ugvc/pipelines/vcfbed/calibrate_bridging_snvs.py:9-66,110-128), featurizer call
order (ugvc/pipelines/run_no_gt_report.py:92-94,307-314).  PARITY UNPINNED (see oracle.py).

It exists (i) as an independent second restatement the vectorised oracle is tested
against, and (ii) as the single-process CPU baseline BASELINE.md calls B0.
"""
from __future__ import annotations

import numpy as np
import pandas as pd

from variantcalling_amd import schema as S      # containers only (named arrays); every number comes from spec.py

from . import spec as P

from . import oracle as O


class PyFasta:
    """Minimal stand-in for pyfaidx.Fasta over the in-memory reference: fa[chrom][a:b] -> str."""

    class _Seq:
        def __init__(self, s):
            self.s = s

        def __getitem__(self, k):
            if isinstance(k, slice):
                a = 0 if k.start is None else k.start
                b = len(self.s) if k.stop is None else k.stop
                if b <= 0:
                    return ""
                return self.s[max(a, 0):b]
            return self.s[k] if 0 <= k < len(self.s) else ""

        def __len__(self):
            return len(self.s)

    def __init__(self, ref: S.Reference):
        self.ref = ref
        self.seqs = {}

    def __getitem__(self, c):
        if c not in self.seqs:            # contigs are materialised as Python strings on first use
            table = np.frombuffer(b"NACGT", dtype=np.uint8)
            lo, hi = int(self.ref.contig_off[c]), int(self.ref.contig_off[c + 1])
            self.seqs[c] = self._Seq(table[self.ref.codes[lo:hi]].tobytes().decode())
        return self.seqs[c]


def table_to_frame(vt: S.VariantTable) -> pd.DataFrame:
    """get_vcf_df-shaped frame: one Python object per cell."""
    rows = []
    pool = vt.alleles
    for i in range(vt.n):
        ref = P.decode_bases(pool[vt.ref_off[i]: vt.ref_off[i] + vt.ref_len[i]])
        alt = P.decode_bases(pool[vt.alt_off[i]: vt.alt_off[i] + vt.alt_len[i]])
        rows.append([int(vt.contig[i]), int(vt.pos[i]), ref, (ref, alt), float(vt.qual[i]),
                     float(vt.sor[i]), int(vt.dp[i]), (int(vt.ad_ref[i]), int(vt.ad_alt[i])),
                     int(vt.gq[i])])
    return pd.DataFrame(rows, columns=["chrom", "pos", "ref", "alleles", "qual", "sor", "dp", "ad", "gq"])


def hmer_length(seq, start_point):
    idx = start_point
    base = seq[start_point]
    if base == "":
        return 0
    while seq[idx] == base:
        idx += 1
    return idx - start_point


def classify_indel(df):
    df["indel"] = df["alleles"].apply(lambda x: len({len(y) for y in x}) > 1)

    def classify(x):
        if not x["indel"]:
            return None
        if len(x["ref"]) < max(len(y) for y in x["alleles"]):
            return "ins"
        return "del"

    df["indel_classify"] = df.apply(classify, axis=1, result_type="reduce")
    df["indel_length"] = df.apply(lambda x: max(abs(len(y) - len(x["ref"])) for y in x["alleles"]), axis=1)
    return df


def is_hmer_indel(df, fa):
    def _is_hmer(rec):
        if not rec["indel"]:
            return (0, None)
        chrom = fa[rec["chrom"]]
        if rec["indel_classify"] == "ins":
            alt = [x for x in rec["alleles"] if x != rec["ref"]][0][1:]
            if len(set(alt)) != 1:
                return (0, None)
            if chrom[rec["pos"]] != alt[0]:
                return (0, None)
            return (hmer_length(chrom, rec["pos"]), alt[0])
        del_seq = rec["ref"][1:]
        if len(set(del_seq)) != 1:
            return (0, None)
        if chrom[rec["pos"] + len(rec["ref"]) - 1] != del_seq[0]:
            return (0, None)
        return (len(del_seq) + hmer_length(chrom, rec["pos"] + len(rec["ref"]) - 1), del_seq[0])

    res = df.apply(_is_hmer, axis=1, result_type="reduce")
    df["hmer_indel_length"] = [x[0] for x in res]
    df["hmer_indel_nuc"] = [x[1] for x in res]
    return df


def _pad(s, size, left):
    return ("N" * (size - len(s)) + s) if left else (s + "N" * (size - len(s)))


def get_motif_around(df, size, fa):
    def _motif(rec):
        chrom = fa[rec["chrom"]]
        pos = rec["pos"]
        if rec["indel"] and rec["hmer_indel_length"] > 0:
            h = rec["hmer_indel_length"]
            l, r = chrom[pos - size: pos], chrom[pos + h: pos + h + size]
        elif rec["indel"]:
            n = len(rec["ref"])
            l, r = chrom[pos - size: pos], chrom[pos + n - 1: pos + n - 1 + size]
        else:
            l, r = chrom[pos - size - 1: pos - 1], chrom[pos: pos + size]
        # BUILDER-DEFINED edge rule: clipped bases read as N
        return _pad(l, size, True), _pad(r, size, False)

    res = df.apply(_motif, axis=1, result_type="reduce")
    df["left_motif"] = [x[0] for x in res]
    df["right_motif"] = [x[1] for x in res]
    return df


def get_gc_content(df, window, fa):
    def _gc(rec):
        chrom = fa[rec["chrom"]]
        beg = rec["pos"] - int(window / 2)
        seq = chrom[beg: beg + window]
        if len(seq) == 0:
            return 0.0
        seq_gc = seq.replace("A", "").replace("T", "")
        return float(len(seq_gc)) / len(seq)

    df["gc_content"] = df.apply(_gc, axis=1)
    return df


def generate_key_from_sequence(sequence, flow_order):
    if any(x not in "ACGT" for x in sequence):
        raise ValueError("non-standard nucleotide")
    key, pos, s = [], 0, 0
    while pos < len(sequence):
        base = flow_order[s % len(flow_order)]
        h = 0
        while pos + h < len(sequence) and sequence[pos + h] == base:
            h += 1
        key.append(h)
        pos += h
        s += 1
    return np.array(key)


def annotate_cycle_skip(df, flow_order):
    def _css(rec):
        if rec["indel"] or len(rec["alleles"]) > 2:
            return "NA"
        alt = [y for y in rec["alleles"] if y != rec["ref"]][0]
        try:
            kr = generate_key_from_sequence(rec["left_motif"] + rec["ref"] + rec["right_motif"], flow_order)
            ka = generate_key_from_sequence(rec["left_motif"] + alt + rec["right_motif"], flow_order)
        except ValueError:
            return "non-skip"
        if len(kr) != len(ka):
            return "cycle-skip"
        d = kr != ka
        if np.any(kr[d] == 0) or np.any(ka[d] == 0):
            return "possible-cycle-skip"
        return "non-skip"

    df["cycleskip_status"] = df.apply(_css, axis=1)
    return df


NUC = {None: 0, "N": 0, "A": 1, "C": 2, "G": 3, "T": 4}
CLS = {None: 0, "ins": 1, "del": 2}
CSS = {n: i for i, n in enumerate(P.CSS_NAMES)}


def motif_code(m):
    v = 0
    for ch in m:
        v = v * 5 + NUC[ch]
    return v


def filter_variants_idiom(vt, ref, runs, tracks, blacklist, sk_models, flow_order="TGCA",
                          hpol_len=10, hpol_dist=10, mark_hpol=True, fasta=None):
    """End-to-end B0 path.  `sk_models` = [sklearn estimator per group] (or FlatForest, which is
    then evaluated with oracle.forest_predict).  Returns (FilterResult, frame)."""
    fa = fasta or PyFasta(ref)
    df = table_to_frame(vt)
    df = classify_indel(df)
    df = is_hmer_indel(df, fa)
    df = get_motif_around(df, P.MOTIF_SIZE, fa)
    df = get_gc_content(df, P.GC_WINDOW, fa)
    df = annotate_cycle_skip(df, flow_order)
    contig = vt.contig
    if runs is not None:
        ins, close = O.hmer_run_flags(runs, contig, vt.pos, hpol_len, hpol_dist)
    else:
        ins = close = np.zeros(vt.n, dtype=bool)
    df["inside_hmer_run"], df["close_to_hmer_run"] = ins, close
    for t, tr in enumerate(tracks):
        df[f"track{t}"] = O.inside_track(tr, contig, vt.pos)
    bl = set(int(k) for k in blacklist) if blacklist is not None else set()
    df["blacklst"] = [((int(c) << 32) | int(p)) in bl for c, p in zip(df["chrom"], df["pos"])]

    # feature_prepare: strings -> numbers, f32 matrix in schema.feature_names order
    X = np.zeros((vt.n, P.N_BASE_FEATURES + len(tracks)), dtype=np.float32)
    X[:, 0] = df["qual"]; X[:, 1] = df["sor"]; X[:, 2] = df["dp"]
    X[:, 3] = df["ad"].apply(lambda a: a[0]); X[:, 4] = df["ad"].apply(lambda a: a[1])
    X[:, 5] = [np.float32(a[1]) / np.float32(d) if d > 0 else np.float32(0) for a, d in zip(df["ad"], df["dp"])]
    X[:, 6] = df["gq"]
    X[:, 7] = df["indel_classify"].map(lambda c: CLS[c])
    X[:, 8] = df["indel_length"]; X[:, 9] = df["hmer_indel_length"]
    X[:, 10] = df["hmer_indel_nuc"].map(lambda c: NUC[c])
    X[:, 11] = df["left_motif"].map(motif_code); X[:, 12] = df["right_motif"].map(motif_code)
    X[:, 13] = df["gc_content"].astype(np.float32)
    X[:, 14] = df["cycleskip_status"].map(lambda c: CSS[c])
    X[:, 15] = df["inside_hmer_run"]; X[:, 16] = df["close_to_hmer_run"]
    for t in range(len(tracks)):
        X[:, 17 + t] = df[f"track{t}"]
    group = np.where(~df["indel"], 0, np.where(df["hmer_indel_length"] > 0, 1, 2))

    score = np.zeros(vt.n, dtype=np.float32)
    flt = np.zeros(vt.n, dtype=np.uint8)
    for g, m in enumerate(sk_models):
        sel = np.where(group == g)[0]
        if sel.size == 0:
            continue
        if isinstance(m, S.FlatForest):
            p0, p1 = O.forest_predict(m, X[sel])
            if m.kind == P.MODEL_GBT:
                score[sel] = p1
                flt[sel] = np.where(p0 > np.float32(0), 0, 1)
                continue
        else:
            pp = m.predict_proba(X[sel])
            p0, p1 = pp[:, 0], pp[:, 1]
        score[sel] = p1.astype(np.float32)
        flt[sel] = np.where(p1 > p0, P.FILTER_PASS, P.FILTER_LOW_SCORE)

    # per-record write-back loop (pattern: calibrate_bridging_snvs.py:110-128)
    flags = np.zeros(vt.n, dtype=np.uint8)
    for i in range(vt.n):
        f = 0
        if mark_hpol and (df["inside_hmer_run"].iat[i] or df["close_to_hmer_run"].iat[i]):
            f |= P.FLAG_HPOL_RUN
        if df["blacklst"].iat[i]:
            f |= P.FLAG_COHORT_FP
        for t in range(len(tracks)):
            if df[f"track{t}"].iat[i]:
                f |= 1 << (P.FLAG_TRACK0_SHIFT + t)
        flags[i] = f
    return S.FilterResult(score, flt, flags), df, X
