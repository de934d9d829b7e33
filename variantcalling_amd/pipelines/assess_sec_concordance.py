"""assess_sec_concordance: what tagging systematic errors (SEC) does to a callset's accuracy against its ground truth.

Stands in for `ugbio_filtering.sec.assess_sec_concordance.run(argv)` (registered at /root/reference/ugvc/__main__.py:19,44).
Body and flags live in the absent submodule: the FLAGS AND THE TABLE BELOW ARE BUILDER-DEFINED; the parts that are the
reference's own are the inputs' vocabulary - the comparison frame of `run_comparison_pipeline` with its `classify` column
(tp / fp / fn, /root/reference/ugvc/reports/report_utils.py:415-505), a call carrying "SEC" in FILTER being read as
filter = "SEC" (report_utils.py:71-75,408-413) - and the accuracy arithmetic (`get_precision` / `get_recall` / `get_f1`,
/root/reference/ugvc/utils/stats_utils.py:76-138, as restated in variantcalling_amd/evaluate.py).

Inputs: the callset after correct_systematic_errors (FILTER may hold SEC, INFO/SEC_LR the ratio) and the concordance
HDF5 of the SAME calls before the correction (keys per contig or one frame; columns chrom / pos / classify and, if
present, indel / hmer_indel_length for the categories).  Host-side tool: it joins the two on (chrom, pos), counts - per
variant category - the true and false calls the SEC tag removes, and writes the accuracy with and without the tag:
`<prefix>.sec_concordance.csv`."""
from __future__ import annotations

import argparse
import csv
import logging
import sys

import numpy as np

logger = logging.getLogger("ugvc")


def get_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="assess_sec_concordance.py", description=run.__doc__)
    ap.add_argument("--input_file", help="VCF written by correct_systematic_errors", type=str, required=True)
    ap.add_argument("--concordance_h5_input", help="comparison HDF5 of the same calls (classify = tp / fp / fn)", type=str, required=True)
    ap.add_argument("--reference_file", help="Indexed reference FASTA file (contig names and order)", type=str, required=True)
    ap.add_argument("--output_prefix", help="Prefix of the report file", type=str, required=True)
    ap.add_argument("--dataset_key", help="key of the HDF5 frame to read", type=str, default="all")
    ap.add_argument("--classify_column", help="column of the frame holding tp / fp / fn", type=str, default="classify")
    return ap


def categories(indel: np.ndarray, hmer_len: np.ndarray) -> dict:
    """Row masks of the report's variant categories (names as in expected.out.stats.csv of evaluate_concordance)."""
    snp = ~indel
    return {"SNP": snp, "Indel": indel, "non-hmer Indel": indel & (hmer_len == 0), "hmer Indel <=4": indel & (hmer_len > 0) & (hmer_len <= 4),
            "hmer Indel >4": indel & (hmer_len > 4), "ALL": np.ones(indel.size, bool)}


def accuracy(tp: int, fp: int, fn: int) -> tuple:
    """(precision, recall, f1) by the reference's formulas (variantcalling_amd/evaluate.py: get_precision / get_recall / get_f1)."""
    from .. import evaluate as E
    p, r = float(E.get_precision(fp, tp)), float(E.get_recall(fn, tp))
    return p, r, float(E.get_f1(p, r))


def assess(classify: np.ndarray, sec: np.ndarray, indel: np.ndarray, hmer_len: np.ndarray) -> list:
    """Report rows from per-row arrays: classify in {"tp", "fp", "fn"}, sec = the call carries the SEC tag (False for fn rows)."""
    cls = np.asarray(classify).astype(str)
    tp, fp, fn = cls == "tp", cls == "fp", cls == "fn"
    rows = []
    for name, m in categories(indel, hmer_len).items():
        n_tp, n_fp, n_fn = int((tp & m).sum()), int((fp & m).sum()), int((fn & m).sum())
        tp_sec, fp_sec = int((tp & m & sec).sum()), int((fp & m & sec).sum())
        p0, r0, f0 = accuracy(n_tp, n_fp, n_fn)
        # a true call tagged SEC is lost (it becomes a false negative), a false call tagged SEC is removed
        p1, r1, f1 = accuracy(n_tp - tp_sec, n_fp - fp_sec, n_fn + tp_sec)
        rows.append(dict(group=name, tp=n_tp, fp=n_fp, fn=n_fn, tp_tagged_sec=tp_sec, fp_tagged_sec=fp_sec, precision=p0, recall=r0, f1=f0,
                         precision_after_sec=p1, recall_after_sec=r1, f1_after_sec=f1))
    return rows


def run(argv: list[str]):
    """Assess SEC against the ground truth: accuracy of the callset with and without the calls tagged SEC"""
    args = get_parser().parse_args(argv[1:])
    from ..io import concordance, vcf as pv, vcf_native
    names = vcf_native.read_fasta_names(args.reference_file)
    vcf = pv.read_vcf(args.input_file, names)
    vt = vcf.table
    # (orig_filter is in FILE order, the table in sorted order: vcf.order[k] = file row of table row k)
    tagged_sorted = np.array(["SEC" in vcf.orig_filter[int(j)].split(";") for j in vcf.order], bool) if vt.n else np.zeros(0, bool)
    call_key = (vt.contig.astype(np.int64) << 32) | vt.pos.astype(np.int64)
    frame = concordance.read_concordance(args.concordance_h5_input, args.dataset_key)
    chrom = np.asarray(frame["chrom"]).astype(str)
    index = {n: i for i, n in enumerate(names)}
    c = np.array([index.get(x, -1) for x in chrom], np.int64)
    key = (c << 32) | np.asarray(frame["pos"]).astype(np.int64)
    classify = np.asarray(frame[args.classify_column]).astype(str)
    indel = np.asarray(frame["indel"]).astype(bool) if "indel" in frame else np.zeros(key.size, bool)
    hmer = np.nan_to_num(np.asarray(frame["hmer_indel_length"], dtype=np.float64)).astype(np.int64) if "hmer_indel_length" in frame else np.zeros(key.size, np.int64)
    # SEC tag of every frame row: the call at the same (chrom, pos); rows without a call (fn) carry none
    order = np.argsort(call_key, kind="stable")
    j = np.searchsorted(call_key[order], key)
    j = np.minimum(j, max(call_key.size - 1, 0))
    found = (call_key[order][j] == key) if call_key.size else np.zeros(key.size, bool)
    sec = np.zeros(key.size, bool)
    if call_key.size:
        sec[found] = tagged_sorted[order][j[found]]
    rows = assess(classify, sec, indel, hmer)
    with open(args.output_prefix + ".sec_concordance.csv", "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    logger.info("%d frame rows, %d calls tagged SEC (%d true, %d false)", key.size, int(sec.sum()), rows[-1]["tp_tagged_sec"], rows[-1]["fp_tagged_sec"])
    return 0


if __name__ == "__main__":
    run(sys.argv)
