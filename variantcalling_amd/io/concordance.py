"""Concordance frames (the HDF5 tables of the reference's comparison / training / evaluation tools) <-> the SoA
variant table of the hot path.

Column set and types: the 25-column frame of
/root/reference/test/resources/unit/comparison/test_vcf_pipeline_utils/annotate_concordance_h5_input.hdf (chrom, pos,
ref, alleles, gt_ultima, gt_ground_truth, sync, call, base, indel, classify, classify_gt, filter, qual, sor, as_sor,
as_sorp, fs, vqsr_val, qd, dp, ad, tree_score, tlod, af); shape of the per-record values as `get_vcf_df` produces them
(/root/reference/ugvc/reports/report_wo_gt.ipynb:1207-1210: `alleles`, `ad`, `gt` are tuples, missing values None /
NaN).  File layout: one frame per contig key plus bookkeeping keys, read back with key="all" and a skip list
(/root/reference/ugvc/pipelines/evaluate_concordance.py:84-89).  Labels: `classify` / `classify_gt` in {tp, fp, fn}
(:96; report_utils.py:449-457); an "fn" row is a truth variant with no call and cannot be featurised.

Multi-allelic rows follow io/vcf.py (first ALT); rows on contigs the reference does not name are dropped."""
from __future__ import annotations

import numpy as np

from .. import schema as S
from . import h5

SKIP_KEYS_ALL = ("concordance", "scored_concordance", "input_args", "comparison_result", "labels", "training_set",
                 "optimal_recall_precision")


def _concat(frames):
    frames = [f for f in frames if f is not None]
    if not frames:
        return h5.Frame()
    cols = list(frames[0].keys())
    for f in frames[1:]:
        if list(f.keys()) != cols:
            raise ValueError(f"frames to concatenate differ in columns: {cols} vs {list(f.keys())}")
    out = h5.Frame()
    for c in cols:
        parts = [f[c] for f in frames]
        kinds = {p.dtype for p in parts}
        out[c] = np.concatenate([p.astype(object) for p in parts]) if len(kinds) > 1 and any(p.dtype == object for p in parts) \
            else np.concatenate(parts)
    return out


def read_concordance(path: str, key: str = "all", skip_keys=None, contigs=None) -> h5.Frame:
    """`read_hdf(path, key, skip_keys)` of the reference's h5 helpers: key "all" concatenates every frame of the file
    (in key order) but the skipped ones; `contigs` restricts that to the named keys (`--list_of_contigs_to_read`)."""
    if key != "all":
        return h5.read_hdf(path, key)
    skip = set(SKIP_KEYS_ALL if skip_keys is None else skip_keys)
    with h5.H5File(path) as f:
        keys = [k for k in f.keys() if k not in skip and f[k].attrs.get("pandas_type") == "frame"]
    if contigs:
        keys = [k for k in keys if k in set(contigs)]
    if not keys:
        raise KeyError(f"{path}: no frame left to read (keys skipped: {sorted(skip)})")
    return _concat([h5.read_hdf(path, k) for k in keys])


def _num(col, n, default=0.0):
    """Numeric view of a column that pandas may have left as object (None / NaN / tuples -> max)."""
    if col is None:
        return np.full(n, default, np.float64)
    a = np.asarray(col)
    if a.dtype != object:
        return np.nan_to_num(a.astype(np.float64), nan=default)
    out = np.full(n, default, np.float64)
    for i, x in enumerate(a):
        if x is None:
            continue
        if isinstance(x, (tuple, list, np.ndarray)):
            vals = [float(v) for v in x if v is not None]
            if vals:
                out[i] = max(vals)
        else:
            try:
                v = float(x)
            except (TypeError, ValueError):
                continue
            if v == v:
                out[i] = v
    return out


def frame_to_table(fr: h5.Frame, contig_names, is_mutect: bool = False, label_column: str = "classify"):
    """-> (VariantTable, rows, label).  `rows[k]` is the frame row behind table row k (the table is sorted by
    (contig, pos), stably; rows without a call or off the reference are left out); label: 1 tp / 0 fp / -1 other."""
    n = fr.n_rows
    for c in ("chrom", "pos", "ref", "alleles"):
        if c not in fr:
            raise KeyError(f"concordance frame has no {c!r} column (columns: {list(fr.keys())})")
    idx = {name: i for i, name in enumerate(contig_names)}
    chrom = np.asarray(fr["chrom"], dtype=object)
    uniq, inv = np.unique(chrom.astype(str), return_inverse=True) if n else (np.zeros(0, str), np.zeros(0, np.int64))
    contig = np.array([idx.get(c, -1) for c in uniq], dtype=np.int64)[inv] if n else np.zeros(0, np.int64)
    pos = _num(fr["pos"], n).astype(np.int64)
    ref, alleles = fr["ref"], fr["alleles"]
    # first ALT of every called row (None: no call - a missed truth variant - or a malformed row)
    alt = [a[1] if isinstance(a, (tuple, list, np.ndarray)) and len(a) > 1 and isinstance(a[1], str) and a[1] else None for a in alleles]
    ok = (contig >= 0) & np.fromiter((x is not None for x in alt), bool, n) & \
        np.fromiter((isinstance(r, str) and len(r) > 0 for r in ref), bool, n)
    rows = np.flatnonzero(ok)
    rows = rows[np.lexsort((rows, pos[rows], contig[rows]))]
    m = rows.size
    refs = [ref[i].encode() for i in rows]
    alts = [alt[i].encode() for i in rows]
    rl64 = np.fromiter((len(x) for x in refs), np.int64, m)
    al64 = np.fromiter((len(x) for x in alts), np.int64, m)
    if m and max(int(rl64.max()), int(al64.max())) > 65535:
        # the allele-length columns are u16 (the native VCF reader rejects such records as well): a wrapped length
        # would misalign every later allele offset
        bad = int(np.flatnonzero((rl64 > 65535) | (al64 > 65535))[0])
        raise ValueError(f"allele longer than 65535 bases at row {int(rows[bad])} ({int(max(rl64[bad], al64[bad]))} bases)")
    rl, al = rl64.astype(np.uint16), al64.astype(np.uint16)
    off = np.concatenate([[0], np.cumsum(rl.astype(np.int64) + al)])
    pool = np.frombuffer(b"".join(r + a for r, a in zip(refs, alts)), dtype=np.uint8) if m else np.zeros(0, np.uint8)

    def first_two(col):
        """(x[0], x[1]) of a tuple-valued column as two int arrays; missing / NaN -> 0."""
        a0, a1 = np.zeros(m, np.int32), np.zeros(m, np.int32)
        if col is None:
            return a0, a1
        for k, x in enumerate(col[rows] if isinstance(col, np.ndarray) else [col[i] for i in rows]):
            if isinstance(x, (tuple, list, np.ndarray)) and len(x):
                if x[0] is not None and x[0] == x[0]:
                    a0[k] = int(x[0])
                if len(x) > 1 and x[1] is not None and x[1] == x[1]:
                    a1[k] = int(x[1])
        return a0, a1

    adr, ada = first_two(fr.get("ad"))
    gt = np.zeros(m, np.uint8)
    g = fr.get("gt_ultima", fr.get("gt"))
    if g is not None:
        code = {(1, 1): 2}
        for k, x in enumerate(g[rows] if isinstance(g, np.ndarray) else [g[i] for i in rows]):
            if isinstance(x, (tuple, list, np.ndarray)):
                x = tuple(x)
                gt[k] = code.get(x, 1 if 1 in x else 0)
    tlod = _num(fr.get("tlod"), n)[rows]
    qual = (10.0 * tlod) if is_mutect else _num(fr.get("qual"), n)[rows]
    vt = S.VariantTable(
        contig=contig[rows].astype(np.uint16), pos=pos[rows].astype(np.int32), ref_len=rl, alt_len=al,
        ref_off=off[:-1].astype(np.uint32), alt_off=(off[:-1] + rl).astype(np.uint32), alleles=S._ASCII_TO_CODE[pool],
        qual=qual.astype(np.float32), sor=_num(fr.get("sor"), n)[rows].astype(np.float32),
        dp=_num(fr.get("dp"), n)[rows].astype(np.int32), ad_ref=adr, ad_alt=ada,
        gq=np.clip(_num(fr.get("gq"), n)[rows], 0, 255).astype(np.uint8), gt=gt)
    vt.validate()
    label = np.full(m, -1, np.int8)
    if label_column in fr:
        lab = np.asarray(fr[label_column], dtype=object)[rows]
        label[lab == "tp"] = 1
        label[lab == "fp"] = 0
    return vt, rows, label


def filter_strings(res: S.FilterResult, n_tracks: int = 0) -> np.ndarray:
    """FILTER column text of a scored table: PASS | [HPOL_RUN;][COHORT_FP;][LOW_SCORE] (docs/howto-callset-filter.md:61-65)."""
    table = np.empty(8, object)
    for code in range(8):
        tags = [t for bit, t in ((S.FLAG_HPOL_RUN, "HPOL_RUN"), (S.FLAG_COHORT_FP, "COHORT_FP"), (4, "LOW_SCORE")) if code & bit]
        table[code] = ";".join(tags) if tags else "PASS"
    code = (res.flags & np.uint8(S.FLAG_HPOL_RUN | S.FLAG_COHORT_FP)).astype(np.int64) | ((res.filter != S.FILTER_PASS).astype(np.int64) << 2)
    return table[code]


def table_to_frame(vt: S.VariantTable, contig_names, label=None, res: S.FilterResult | None = None) -> h5.Frame:
    """The call-side columns of a concordance frame for a (scored) table, indexed by (chrom, pos) as the reference's
    frames are."""
    n = vt.n
    chrom = np.array([contig_names[c] for c in vt.contig], dtype=object)
    ref, alleles, ad, gt = (np.empty(n, object) for _ in range(4))
    for i in range(n):
        r = S.decode_bases(vt.alleles[vt.ref_off[i]:vt.ref_off[i] + vt.ref_len[i]])
        a = S.decode_bases(vt.alleles[vt.alt_off[i]:vt.alt_off[i] + vt.alt_len[i]])
        ref[i], alleles[i] = r, (r, a)
        ad[i] = (int(vt.ad_ref[i]), int(vt.ad_alt[i]))
        gt[i] = {0: (0, 0), 1: (0, 1), 2: (1, 1)}[int(vt.gt[i])]
    cols = [("chrom", chrom), ("pos", vt.pos.astype(np.int64)), ("ref", ref), ("alleles", alleles), ("gt_ultima", gt),
            ("indel", vt.ref_len != vt.alt_len), ("qual", vt.qual.astype(np.float64)), ("sor", vt.sor.astype(np.float64)),
            ("dp", vt.dp.astype(np.float64)), ("ad", ad), ("gq", vt.gq.astype(np.float64))]
    if label is not None:
        lab = np.full(n, None, object)
        lab[np.asarray(label) == 1] = "tp"
        lab[np.asarray(label) == 0] = "fp"
        cols += [("classify", lab), ("classify_gt", lab.copy())]
    if res is not None:
        cols += [("filter", filter_strings(res)), ("tree_score", res.tree_score.astype(np.float64))]
    return h5.Frame(cols, index=[chrom, vt.pos.astype(np.int64)], index_names=["chrom", "pos"])
