"""CPU ORACLE - TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

Normative numpy restatement of the post-GATK filter hot path
(featurize -> interval/blacklist lookup -> tree-ensemble score -> FILTER), i.e. steps 2-4
of `ugvc filter_variants_pipeline`.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module.

PARITY UNPINNED.  The reference implementation of this path lives in the un-vendored
submodule Ultimagen/ugbio-utils (packages ugbio_core / ugbio_filtering /
ugbio_comparison; /root/reference/.gitmodules:1-4, directory empty; no pinned revision:
/root/reference/.github/workflows/ci.yml:21-32) and all of its fixtures are un-pulled
git-LFS pointers (/root/reference/.gitattributes:1-19).  Each function below therefore
cites the in-tree evidence it follows (call sites, docs, sibling code, VCF header) and marks
every remaining choice BUILDER-DEFINED.  What IS pinned against the reference's own
known-answer tests: the SEC statistic (stats.py, test/unit/utils/test_stats_utils.py),
phred helpers (test/unit/utils/test_math_utils.py) and `is_homopolymer_snp`
(ugvc/pipelines/vcfbed/calibrate_bridging_snvs.py:9-66, restated in bridging.py).
Tree-ensemble scoring is pinned against scikit-learn `predict_proba`/`predict` (bit-exact).

Coordinates: `pos` is the 1-based VCF POS; g0 = contig_off[contig] + pos - 1 is the
0-based index of that base in the concatenated reference.  The reference indexes pyfaidx
sequences with 0-based slices of the 1-based `pos` (e.g. `chrom[pos]` is the base AFTER the
variant's first base); those expressions are kept verbatim in the comments.
"""
from __future__ import annotations

import numpy as np

from variantcalling_amd import schema as S      # containers only (named arrays); every number comes from spec.py

from . import spec as P

MOTIF = P.MOTIF_SIZE
GCW = P.GC_WINDOW


# --------------------------------------------------------------------------- reference access
def _fetch(ref: S.Reference, contig: np.ndarray, idx: np.ndarray) -> np.ndarray:
    """Base codes at global indices `idx`; positions outside the variant's contig read as N.

    BUILDER-DEFINED edge rule (pyfaidx clips slices at contig ends; we pad with N)."""
    lo = ref.contig_off[contig]
    hi = ref.contig_off[contig.astype(np.int64) + 1]
    ok = (idx >= lo) & (idx < hi)
    out = np.zeros(idx.shape, dtype=np.uint8)
    out[ok] = ref.codes[idx[ok]]
    return out


def _run_length_forward(ref: S.Reference, contig: np.ndarray, start: np.ndarray) -> np.ndarray:
    """`hmer_length(seq, start_point)`: number of consecutive bases equal to seq[start], going
    right, unbounded except by the contig end (semantics shared with the
    flow-key definition used at ugvc/scripts/collect_hpol_table.py:99-115)."""
    hi = ref.contig_off[contig.astype(np.int64) + 1]
    base = _fetch(ref, contig, start)
    n = np.zeros(start.shape, dtype=np.int64)
    act = (start >= ref.contig_off[contig]) & (start < hi)
    k = 0
    while act.any():
        n[act] += 1
        k += 1
        nxt = start + k
        act = act & (nxt < hi)
        idx = np.where(act)[0]
        same = ref.codes[nxt[idx]] == base[idx]
        act[idx[~same]] = False
    return n


def _motif_code(bases: np.ndarray) -> np.ndarray:
    """[n, MOTIF] base codes -> base-5 integer, first base most significant (N=0)."""
    w = 5 ** np.arange(MOTIF - 1, -1, -1, dtype=np.int64)
    return (bases.astype(np.int64) * w[None, :]).sum(axis=1).astype(np.int32)


def motif_to_str(code: int) -> str:
    out = []
    for k in range(MOTIF):
        out.append(P.CODE_TO_CHAR[(code // 5 ** (MOTIF - 1 - k)) % 5])
    return "".join(out)


# --------------------------------------------------------------------------- flow keys
def flow_key(seq: np.ndarray, flow_order: np.ndarray) -> np.ndarray | None:
    """`flow_based_read.generate_key_from_sequence(seq, flow_order)`: per-flow homopolymer
    calls of `seq` under a cyclic flow order (use: ugvc/scripts/collect_hpol_table.py:99-115;
    property test/system/test_collect_hpol_table.py:32-36: cumsum(key) maps flow -> base).
    Returns None for a sequence with a non-ACGT base (the reference raises ValueError, which
    annotate_cycle_skip turns into 'non-skip')."""
    if np.any(seq == 0):
        return None
    key = []
    p = 0
    s = 0
    n = len(seq)
    while p < n:
        b = flow_order[s % 4]
        h = 0
        while p + h < n and seq[p + h] == b:
            h += 1
        key.append(h)
        p += h
        s += 1
    return np.array(key, dtype=np.int64)


def cycle_skip_status(ref_seq: np.ndarray, alt_seq: np.ndarray, flow_order: np.ndarray) -> int:
    """X_CSS / `cycleskip_status` of one substitution (header.txt:3382;
    docs/filter_variants_pipeline.md:43-44; SURVEY.md App. A `annotate_cycle_skip`):
    keys of different length -> cycle-skip; same length and, where they differ, one of them
    is 0 -> possible-cycle-skip; else non-skip."""
    kr = flow_key(ref_seq, flow_order)
    ka = flow_key(alt_seq, flow_order)
    if kr is None or ka is None:
        return P.CSS_NON_SKIP
    if len(kr) != len(ka):
        return P.CSS_CYCLE_SKIP
    d = kr != ka
    if np.any(kr[d] == 0) or np.any(ka[d] == 0):
        return P.CSS_POSSIBLE
    return P.CSS_NON_SKIP


def _flow_keys_batch(seqs: np.ndarray, lens: np.ndarray, flow_order: np.ndarray):
    """Vectorised flow_key for padded sequences [n, L] (pad value 255).  Returns
    (keys [n, 4L] padded with -1, key_len [n]); rows holding an N get key_len -1."""
    n, L = seqs.shape
    keys = np.full((n, 4 * L), -1, dtype=np.int16)
    klen = np.zeros(n, dtype=np.int64)
    p = np.zeros(n, dtype=np.int64)
    act = lens > 0
    rows = np.arange(n)
    padded = np.concatenate([seqs, np.full((n, 1), 255, dtype=seqs.dtype)], axis=1)
    for s in range(4 * L):
        if not act.any():
            break
        b = flow_order[s % 4]
        h = np.zeros(n, dtype=np.int64)
        run = act.copy()
        while run.any():
            cur = padded[rows, np.minimum(p + h, L)]
            run = run & (p + h < lens) & (cur == b)
            h[run] += 1
        keys[act, s] = h[act]
        p[act] += h[act]
        klen[act] += 1
        act = act & (p < lens)
    has_n = ((seqs == 0) & (np.arange(L)[None, :] < lens[:, None])).any(axis=1)
    klen[has_n] = -1
    return keys, klen


# --------------------------------------------------------------------------- interval joins
def _inside_track(track: S.IntervalTrack, contig: np.ndarray, pos: np.ndarray):
    """`annotate_intervals` membership test (boolean column per BED stem:
    ugvc/reports/report_data_loader.py:94; docs/howto-callset-filter.md:116-120):
        s = searchsorted(starts, pos) - 1 ; e = searchsorted(ends, pos) ; inside = (s == e)
    evaluated per contig, BED coordinates compared with the VCF POS as they stand
    (start < pos <= end, i.e. the 1-based position lies in the 0-based half-open interval)."""
    n = pos.size
    s = np.zeros(n, dtype=np.int64)
    e = np.zeros(n, dtype=np.int64)
    for c in np.unique(contig):
        m = contig == c
        lo, hi = int(track.contig_ptr[c]), int(track.contig_ptr[c + 1])
        s[m] = np.searchsorted(track.starts[lo:hi], pos[m], side="left") - 1
        e[m] = np.searchsorted(track.ends[lo:hi], pos[m], side="left")
    return s, e


def inside_track(track: S.IntervalTrack, contig: np.ndarray, pos: np.ndarray) -> np.ndarray:
    s, e = _inside_track(track, contig, pos)
    return s == e


def hmer_run_flags(runs: S.IntervalTrack, contig: np.ndarray, pos: np.ndarray,
                   min_len: int, max_dist: int):
    """`close_to_hmer_run(df, runs_file, min_hmer_run_length=L, max_distance=D)` ->
    (inside_hmer_run, close_to_hmer_run): `--runs_file` + `--hpol_filter_length_dist L D`,
    "Length and distance to the hpol run to mark" (docs/filter_variants_pipeline.md:30-33);
    the HPOL_RUN tag is ignored by default in evaluation (evaluate_concordance.py:44-48).
    Runs shorter than L are dropped first; `close` looks at the nearest starts/ends on either
    side within the contig (|delta| < D) and excludes variants inside a run."""
    keep = (runs.ends - runs.starts) >= min_len
    starts, ends = runs.starts[keep], runs.ends[keep]
    ptr = np.concatenate([[0], np.cumsum(keep)])[runs.contig_ptr].astype(np.int32)
    n = pos.size
    inside = np.zeros(n, dtype=bool)
    close = np.zeros(n, dtype=bool)
    for c in np.unique(contig):
        m = np.where(contig == c)[0]
        lo, hi = int(ptr[c]), int(ptr[c + 1])
        if hi == lo:
            continue
        st, en = starts[lo:hi].astype(np.int64), ends[lo:hi].astype(np.int64)
        p = pos[m].astype(np.int64)
        s = np.searchsorted(st, p, side="left") - 1
        cd = np.abs(p - st[np.clip(s, 0, None)]) < max_dist
        cd |= np.abs(st[np.clip(s + 1, None, st.size - 1)] - p) < max_dist
        e = np.searchsorted(en, p, side="left")
        cd |= np.abs(p - en[np.clip(e - 1, 0, None)]) < max_dist
        cd |= np.abs(en[np.clip(e, None, en.size - 1)] - p) < max_dist
        ins = s == e
        inside[m] = ins
        close[m] = cd & ~ins
    return inside, close


def blacklist_hit(keys_sorted: np.ndarray, vkeys: np.ndarray) -> np.ndarray:
    """Membership of (chrom, pos) in the cohort false-positive locus set -> COHORT_FP
    (`--blacklist`, docs/filter_variants_pipeline.md:34-35; docs/howto-callset-filter.md:65)."""
    if keys_sorted.size == 0:
        return np.zeros(vkeys.shape, dtype=bool)
    i = np.searchsorted(keys_sorted, vkeys, side="left")
    i = np.minimum(i, keys_sorted.size - 1)
    return keys_sorted[i] == vkeys


# --------------------------------------------------------------------------- featurize
def featurize(vt: S.VariantTable, ref: S.Reference, runs: S.IntervalTrack | None,
              tracks: list, flow_order: str = "TGCA", hpol_len: int = 10, hpol_dist: int = 10) -> dict:
    """`annotate_concordance(df, fasta, ...)` (call site ugvc/pipelines/run_no_gt_report.py:314):
    classify_indel / is_hmer_indel / get_motif_around (run_no_gt_report.py:92-94, produced
    columns consumed at :133-143), gc_content + interval columns
    (ugvc/reports/report_data_loader.py:67-94), cycleskip_status, hmer-run flags.
    Returns a dict of per-variant integer/float columns plus the f32 feature matrix `X`."""
    n = vt.n
    contig = vt.contig
    c64 = contig.astype(np.int64)
    g0 = ref.contig_off[c64] + vt.pos.astype(np.int64) - 1
    ref_len = vt.ref_len.astype(np.int64)
    alt_len = vt.alt_len.astype(np.int64)
    ro = vt.ref_off.astype(np.int64)
    ao = vt.alt_off.astype(np.int64)
    pool = vt.alleles

    # classify_indel: indel <=> alleles of different length; ins if ref shorter (run_no_gt_report.py:92)
    indel = ref_len != alt_len
    classify = np.where(~indel, P.INDEL_NONE, np.where(ref_len < alt_len, P.INDEL_INS, P.INDEL_DEL))
    indel_length = np.abs(alt_len - ref_len)

    # is_hmer_indel:
    #   ins: alt[1:] is one repeated base b and fasta[chrom][pos] == b
    #        -> (hmer_length(fasta[chrom], pos), b)
    #   del: ref[1:] is one repeated base b and fasta[chrom][pos + len(ref) - 1] == b
    #        -> (len(ref[1:]) + hmer_length(fasta[chrom], pos + len(ref) - 1), b)
    hmer_len = np.zeros(n, dtype=np.int64)
    hmer_nuc = np.zeros(n, dtype=np.int64)
    for cls, off, ln in ((P.INDEL_INS, ao, alt_len), (P.INDEL_DEL, ro, ref_len)):
        m = np.where(classify == cls)[0]
        if m.size == 0:
            continue
        b = pool[off[m] + 1]
        mono = np.ones(m.size, dtype=bool)
        maxl = int(ln[m].max())
        for k in range(2, maxl):
            has = ln[m] > k
            mono[has] &= pool[off[m][has] + k] == b[has]
        start = g0[m] + 1 if cls == P.INDEL_INS else g0[m] + ref_len[m]
        nxt = _fetch(ref, contig[m], start)
        inb = (start >= ref.contig_off[c64[m]]) & (start < ref.contig_off[c64[m] + 1])
        ok = mono & inb & (nxt == b)
        mm = m[ok]
        run = _run_length_forward(ref, contig[mm], start[ok])
        hmer_len[mm] = run + (0 if cls == P.INDEL_INS else ref_len[mm] - 1)
        hmer_nuc[mm] = b[ok]
    is_h = indel & (hmer_len > 0)

    # get_motif_around(df, 5, fasta):
    #   snp:            chrom[pos-size-1 : pos-1], chrom[pos : pos+size]
    #   non-hmer indel: chrom[pos-size : pos],     chrom[pos+len(ref)-1 : pos+len(ref)-1+size]
    #   hmer indel:     chrom[pos-size : pos],     chrom[pos+hmer_len : pos+hmer_len+size]
    lstart = np.where(indel, g0 - (MOTIF - 1), g0 - MOTIF)
    rstart = np.where(~indel, g0 + 1, np.where(is_h, g0 + 1 + hmer_len, g0 + ref_len))
    ar = np.arange(MOTIF, dtype=np.int64)[None, :]
    lm_b = _fetch(ref, contig[:, None].repeat(MOTIF, 1), lstart[:, None] + ar)
    rm_b = _fetch(ref, contig[:, None].repeat(MOTIF, 1), rstart[:, None] + ar)
    left_motif = _motif_code(lm_b)
    right_motif = _motif_code(rm_b)

    # gc_content: beg = pos - window/2; seq = chrom[beg : beg + window];
    #             gc = len(seq without 'A','T') / len(seq)   (N counts with G/C - literal)
    arw = np.arange(GCW, dtype=np.int64)[None, :]
    widx = (g0 + 1 - GCW // 2)[:, None] + arw
    lo = ref.contig_off[c64][:, None]
    hi = ref.contig_off[c64 + 1][:, None]
    inb = (widx >= lo) & (widx < hi)
    wb = _fetch(ref, contig[:, None].repeat(GCW, 1), widx)
    gc_cnt = (inb & (wb != P.BASE_A) & (wb != P.BASE_T)).sum(axis=1)
    gc_len = inb.sum(axis=1)
    gc = np.where(gc_len > 0, gc_cnt / np.maximum(gc_len, 1), 0.0).astype(np.float32)

    # annotate_cycle_skip: only for non-indels; seq = left_motif + allele + right_motif
    css = np.full(n, P.CSS_NA, dtype=np.int64)
    fo = P.encode_bases(flow_order)
    sub = np.where(~indel)[0]
    if sub.size:
        L = int(ref_len[sub].max()) + 2 * MOTIF
        rs = np.full((sub.size, L), 255, dtype=np.uint8)
        as_ = np.full((sub.size, L), 255, dtype=np.uint8)
        rs[:, :MOTIF] = lm_b[sub]
        as_[:, :MOTIF] = lm_b[sub]
        al = ref_len[sub]
        for k in range(int(al.max())):
            has = al > k
            rs[has, MOTIF + k] = pool[ro[sub][has] + k]
            as_[has, MOTIF + k] = pool[ao[sub][has] + k]
        for k in range(MOTIF):
            rs[np.arange(sub.size), MOTIF + al + k] = rm_b[sub, k]
            as_[np.arange(sub.size), MOTIF + al + k] = rm_b[sub, k]
        lens = al + 2 * MOTIF
        kr, lr = _flow_keys_batch(rs, lens, fo)
        ka, la = _flow_keys_batch(as_, lens, fo)
        valid = (lr >= 0) & (la >= 0)
        differ_len = valid & (lr != la)
        d = kr != ka
        poss = valid & ~differ_len & ((d & ((kr == 0) | (ka == 0))).any(axis=1))
        st = np.full(sub.size, P.CSS_NON_SKIP, dtype=np.int64)
        st[poss] = P.CSS_POSSIBLE
        st[differ_len] = P.CSS_CYCLE_SKIP
        css[sub] = st

    # hmer-run proximity and interval tracks
    if runs is not None:
        inside_run, close_run = hmer_run_flags(runs, contig, vt.pos, hpol_len, hpol_dist)
    else:
        inside_run = np.zeros(n, dtype=bool)
        close_run = np.zeros(n, dtype=bool)
    track_bits = [inside_track(t, contig, vt.pos) for t in tracks]

    # vaf = ad_alt / dp in f32 (ugvc/reports/report_data_loader.py:24-28: vaf = ad / dp)
    dpf = vt.dp.astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        vaf = np.where(vt.dp > 0, vt.ad_alt.astype(np.float32) / np.where(vt.dp > 0, dpf, np.float32(1)),
                       np.float32(0)).astype(np.float32)

    group = np.where(~indel, P.GROUP_SNP, np.where(is_h, P.GROUP_HINDEL, P.GROUP_NON_HINDEL))
    F = P.N_BASE_FEATURES + len(tracks)
    X = np.zeros((n, F), dtype=np.float32)
    cols = [vt.qual, vt.sor, vt.dp, vt.ad_ref, vt.ad_alt, vaf, vt.gq, classify, indel_length,
            hmer_len, hmer_nuc, left_motif, right_motif, gc, css, inside_run, close_run] + track_bits
    for j, col in enumerate(cols):
        X[:, j] = np.asarray(col).astype(np.float32)
    return dict(indel=indel, indel_classify=classify, indel_length=indel_length,
                hmer_indel_length=hmer_len, hmer_indel_nuc=hmer_nuc, left_motif=left_motif,
                right_motif=right_motif, gc_content=gc, cycleskip_status=css,
                inside_hmer_run=inside_run, close_to_hmer_run=close_run, tracks=track_bits,
                vaf=vaf, group=group, X=X)


# --------------------------------------------------------------------------- tree ensembles
def forest_predict(f: S.FlatForest, X: np.ndarray):
    """Flat-table evaluation of one ensemble.  MODEL_RF: scikit-learn semantics - go left when
    f32 feature <= threshold, class fractions of the reached leaf summed over trees IN TREE
    ORDER in f64, divided by n_trees (`RandomForestClassifier.predict_proba`); class = argmax
    with ties to class 0 (`predict`).  MODEL_GBT: XGBoost semantics - go left when feature <
    threshold, f32 margins added in tree order starting from base_score, score =
    sigmoid(margin) in f32, class 1 iff margin > 0 (decided on the exactly reproducible margin,
    not on the rounded sigmoid).  Returns (p0, p1) f64 / (margin, score)."""
    n = X.shape[0]
    if f.kind == P.MODEL_RF:
        acc0 = np.zeros(n, dtype=np.float64)
        acc1 = np.zeros(n, dtype=np.float64)
    else:
        acc = np.full(n, np.float32(f.base_score), dtype=np.float32)
    rows = np.arange(n)
    for t in range(f.n_trees):
        idx = np.full(n, f.tree_root[t], dtype=np.int64)
        act = f.feature[idx] >= 0
        while act.any():
            ia = idx[act]
            x = X[rows[act], f.feature[ia]]
            if f.kind == P.MODEL_RF:
                go_left = x <= f.threshold[ia]
            else:
                go_left = x < f.threshold[ia]
            idx[act] = np.where(go_left, f.left[ia], f.right[ia])
            act = f.feature[idx] >= 0
        leaf = f.left[idx]
        if f.kind == P.MODEL_RF:
            acc0 += f.leaf_value[leaf, 0]
            acc1 += f.leaf_value[leaf, 1]
        else:
            acc = (acc + f.leaf_value[leaf, 0].astype(np.float32)).astype(np.float32)
    if f.kind == P.MODEL_RF:
        return acc0 / f.n_trees, acc1 / f.n_trees
    score = (np.float32(1) / (np.float32(1) + np.exp(-acc, dtype=np.float32))).astype(np.float32)
    return acc, score


def score(forests: list, X: np.ndarray, group: np.ndarray):
    """Per-group model application -> (tree_score f32, filter u8): TREE_SCORE and
    PASS / LOW_SCORE (docs/howto-callset-filter.md:61-65; model dict keyed by --model_name,
    one model per variant-type group: report_utils.py:508-538, header.txt:3381)."""
    n = X.shape[0]
    ts = np.zeros(n, dtype=np.float32)
    flt = np.zeros(n, dtype=np.uint8)
    for g, f in enumerate(forests):
        m = np.where(group == g)[0]
        if m.size == 0 or f is None:
            continue
        a, b = forest_predict(f, X[m])
        if f.kind == P.MODEL_RF:
            ts[m] = b.astype(np.float32)
            flt[m] = np.where(b > a, P.FILTER_PASS, P.FILTER_LOW_SCORE)
        else:
            ts[m] = b
            flt[m] = np.where(a > np.float32(0), P.FILTER_PASS, P.FILTER_LOW_SCORE)
    return ts, flt


def filter_variants(vt: S.VariantTable, ref: S.Reference, runs, tracks: list, blacklist,
                    forests: list, flow_order: str = "TGCA", hpol_len: int = 10,
                    hpol_dist: int = 10, mark_hpol: bool = True) -> S.FilterResult:
    """featurize -> lookup -> score -> FILTER/flags: what `filter_variants_pipeline.run` computes
    between reading and writing the VCF."""
    ft = featurize(vt, ref, runs, tracks, flow_order, hpol_len, hpol_dist)
    ts, flt = score(forests, ft["X"], ft["group"])
    flags = np.zeros(vt.n, dtype=np.uint8)
    if mark_hpol:
        flags |= np.where(ft["inside_hmer_run"] | ft["close_to_hmer_run"], P.FLAG_HPOL_RUN, 0).astype(np.uint8)
    if blacklist is not None:
        flags |= np.where(blacklist_hit(blacklist, (vt.contig.astype(np.uint64) << np.uint64(32)) | vt.pos.astype(np.uint64)), P.FLAG_COHORT_FP, 0).astype(np.uint8)
    for t, bits in enumerate(ft["tracks"]):
        flags |= (bits.astype(np.uint8) << np.uint8(P.FLAG_TRACK0_SHIFT + t)).astype(np.uint8)
    return S.FilterResult(ts, flt, flags)


# --------------------------------------------------------------------------- pileup tally (a11)
def pileup_tally(offsets: np.ndarray, obs: np.ndarray) -> dict:
    """BUILDER-DEFINED: the reference only READS FORMAT/AD, DP, SB, VAF
    and INFO/SOR (header.txt:3379,3391-3398; uses at calibrate_bridging_snvs.py:115-117,
    report_data_loader.py:24-28).  Per locus, tally read observations
    (allele 0 ref / 1 alt / 2 other, strand, base quality) into AD by strand, BQ sums, DP, VAF
    and GATK's StrandOddsRatio (added upstream by `gatk VariantAnnotator -A StrandOddsRatio`,
    docs/howto-callset-filter.md:79-85):
        SOR = ln(R + 1/R) + ln(min(a,b)/max(a,b)) - ln(min(c,d)/max(c,d)),
        R = (a*d)/(b*c), table [[refF,refR],[altF,altR]] + 1 in every cell (f64, stored f32)."""
    n = offsets.size - 1
    d = np.diff(offsets)
    locus = np.repeat(np.arange(n), d)
    allele = (obs & 3).astype(np.int64)
    strand = ((obs >> 2) & 1).astype(np.int64)
    bq = (obs >> 3).astype(np.int64)
    cls = allele * 2 + strand
    cnt = np.zeros((n, 8), dtype=np.int64)
    np.add.at(cnt, (locus, cls), 1)
    bqs = np.zeros((n, 4), dtype=np.int64)
    np.add.at(bqs, (locus, allele), bq)
    ad_ref = cnt[:, 0] + cnt[:, 1]
    ad_alt = cnt[:, 2] + cnt[:, 3]
    dp = d.astype(np.int64)
    a = cnt[:, 0] + 1.0
    b = cnt[:, 1] + 1.0
    c = cnt[:, 2] + 1.0
    dd = cnt[:, 3] + 1.0
    R = (a * dd) / (b * c)
    sor = np.log(R + 1.0 / R) + np.log(np.minimum(a, b) / np.maximum(a, b)) \
        - np.log(np.minimum(c, dd) / np.maximum(c, dd))
    with np.errstate(divide="ignore", invalid="ignore"):
        vaf = np.where(dp > 0, ad_alt.astype(np.float32) / np.maximum(dp, 1).astype(np.float32),
                       np.float32(0)).astype(np.float32)
    return dict(ref_fwd=cnt[:, 0].astype(np.int32), ref_rev=cnt[:, 1].astype(np.int32),
                alt_fwd=cnt[:, 2].astype(np.int32), alt_rev=cnt[:, 3].astype(np.int32),
                other=(cnt[:, 4] + cnt[:, 5]).astype(np.int32), dp=dp.astype(np.int32),
                bq_ref=bqs[:, 0].astype(np.int32), bq_alt=bqs[:, 1].astype(np.int32),
                ad_ref=ad_ref.astype(np.int32), ad_alt=ad_alt.astype(np.int32),
                vaf=vaf, sor=sor.astype(np.float32), sor64=sor)
