"""evaluate_concordance: precision / recall of a scored, labelled callset, from and to pandas HDF5.

Drop-in for /root/reference/ugvc/pipelines/evaluate_concordance.py (flags :30-66, flow :71-108): read the comparison
HDF5 (key "all" = every per-contig frame but the bookkeeping keys, :84-89), take `--score_key` as tree_score (:91-99),
classify by `classify` (with --ignore_genotype) or `classify_gt` (:100), write PREFIX.h5 with the keys
`optimal_recall_precision` and `recall_precision_curve` (:102-107), PREFIX.stats.csv (sep ";") and
PREFIX.thresholds.csv (:103,110).  The two table builders (`calc_accuracy_metrics`, `calc_recall_precision_curve`)
live in the absent submodule; here they are `evaluate.accuracy_rows` - pinned on the reference's own expected table,
test/resources/system/test_evaluate_concordance/expected.out.stats.csv - and the FN-aware
`evaluate.precision_recall_curve` (the in-tree ugvc/utils/stats_utils.py:141-210) per variant category.
Key `performance_curve` (BUILDER-DEFINED) holds, per category, the cumulative curve of the in-tree
`ReportUtils.__calc_performance` (ugvc/reports/report_utils.py:494-504); with `--device N` its sort + scan + finish run on
the GPU (`ugvc_pr_curve`), without it on the host - same bytes either way.  Host-side consumer of the hot path's two
output columns: no GPU needed."""
from __future__ import annotations

import argparse
import csv
import logging
import sys

import numpy as np

from .. import evaluate
from ..io import concordance, h5

logger = logging.getLogger("ugvc")


def parse_args(argv: list[str]):
    ap = argparse.ArgumentParser(prog="evaluate_concordance.py", description=run.__doc__)
    ap.add_argument("--input_file", help="Name of the input h5 file", type=str, required=True)
    ap.add_argument("--output_prefix", help="Prefix to output files", type=str, required=True)
    ap.add_argument("--dataset_key", help="h5 dataset name, such as chromosome name", default="all")
    ap.add_argument("--score_key", help="info key name for calculating the score", default="tree_score")
    ap.add_argument("--ignore_genotype", help="ignore genotype when comparing to ground-truth", action="store_true", default=False)
    ap.add_argument("--ignore_filters", help="comma separated list of filters to ignore", default="HPOL_RUN")
    ap.add_argument("--output_bed", help="output bed files of fp/fn/tp per variant-type", action="store_true", default=False)
    ap.add_argument("--use_for_group_testing", help="Column in the h5 to use for grouping (or generate default groupings)", type=str)
    ap.add_argument("--verbosity", help="Verbosity: ERROR, WARNING, INFO, DEBUG", required=False, default="INFO")
    # BUILDER-DEFINED (not a reference flag): the cumulative performance curves on an MI355X; same files either way
    ap.add_argument("--device", help="GPU index (MI355X) for the cumulative performance curves (default: host numpy)", type=int)
    return ap.parse_args(argv)


def passing(filter_col, ignored) -> np.ndarray:
    """FILTER text -> passes once the ignored filters are dropped ("PASS", "", "." and missing all pass)."""
    ignored = set(ignored) | {"PASS", "", "."}
    out = np.ones(len(filter_col), bool)
    for i, f in enumerate(filter_col):
        if isinstance(f, str):
            out[i] = all(t in ignored for t in f.split(";"))
    return out


def group_masks(df: h5.Frame, column=None) -> dict:
    """Row masks per group: the nine variant categories (evaluate.CATEGORIES) or the values of `column`."""
    n = df.n_rows
    if column:
        vals = np.asarray(df[column], dtype=object)
        return {str(v): vals == v for v in sorted({x for x in vals if x is not None and x == x}, key=str)}
    indel = np.asarray(df["indel"]).astype(bool) if "indel" in df else np.zeros(n, bool)
    hmer = np.nan_to_num(np.asarray(df["hmer_indel_length"], dtype=np.float64)) if "hmer_indel_length" in df else np.zeros(n)
    return evaluate.category_masks(indel, hmer)


def calc_accuracy_metrics(df: h5.Frame, classify_column: str, ignored_filters, group_column=None) -> list:
    cls = np.asarray(df[classify_column], dtype=object)
    ok = passing(df["filter"], ignored_filters) if "filter" in df else np.ones(df.n_rows, bool)
    tp, fp, fn = cls == "tp", cls == "fp", cls == "fn"
    masks = group_masks(df, group_column)
    counts = [(int((tp & m).sum()), int((fp & m).sum()), int((tp & m & ok).sum()), int((fp & m & ok).sum()), int((fn & m).sum()))
              for m in masks.values()]
    if group_column:
        saved = evaluate.CATEGORIES
        try:
            evaluate.CATEGORIES = tuple(masks)
            return evaluate.accuracy_rows(counts)
        finally:
            evaluate.CATEGORIES = saved
    return evaluate.accuracy_rows(counts)


def calc_recall_precision_curve(df: h5.Frame, classify_column: str, ignored_filters, group_column=None) -> list:
    """One row per group: the FN-aware curve of the in-tree stats_utils.precision_recall_curve (truth variants without
    a call only scale the recall) and the threshold of the best f1.  BUILDER-DEFINED frame layout (the builder is in
    the absent submodule): group, predictions (thresholds), precision, recall, f1 as arrays, threshold as a scalar -
    the two columns the reference exports are `group` and `threshold` (evaluate_concordance.py:110)."""
    cls = np.asarray(df[classify_column], dtype=object)
    score = np.nan_to_num(np.asarray(df["tree_score"], dtype=np.float64))
    rows = []
    for name, m in group_masks(df, group_column).items():
        sel = m & np.isin(cls, ["tp", "fp", "fn"])
        gtr = (cls[sel] != "fp").astype(int)
        if gtr.size:
            p, r, f1, thr = evaluate.precision_recall_curve(gtr, score[sel], cls[sel] == "fn", pos_label=1)
        else:
            p = r = f1 = thr = np.array([])
        best = float(thr[int(np.argmax(f1))]) if f1.size else float("nan")
        rows.append(dict(group=name, predictions=thr, precision=p, recall=r, f1=f1, threshold=best))
    return rows


def calc_performance_curves(df: h5.Frame, classify_column: str, ignored_filters, group_column=None, pr_curve=None) -> list:
    """One row per group: the cumulative curve of `ReportUtils.__calc_performance` (/root/reference/ugvc/reports/
    report_utils.py:415-505: calls sorted by normalised score, running tp / fp removed from the callset, recall /
    precision / f1 per position :494-504) - what the reports draw from this tool's input.  Truth variants without a call
    (classify == "fn") are the missing candidates (:443-446).  `pr_curve` = `Engine.pr_curve`: sort, scan and the f64
    finish run on the GPU (ugvc_pr_curve), bit-equal to the host statement.  Key `performance_curve`: BUILDER-DEFINED."""
    cls = np.asarray(df[classify_column], dtype=object)
    score = np.asarray(df["tree_score"], dtype=np.float64)
    ok = passing(df["filter"], ignored_filters) if "filter" in df else np.ones(df.n_rows, bool)
    rows = []
    for name, m in group_masks(df, group_column).items():
        tp, fp, fn = (cls == "tp") & m, (cls == "fp") & m, (cls == "fn") & m
        sel = tp | fp | fn
        res, curve = evaluate.calc_performance(score[sel], ok[sel], tp[sel], fp[sel], fn[sel], missing_candidate=fn[sel],
                                               pr_curve=pr_curve)
        empty = np.zeros(0)
        s, r, p, f = curve if curve is not None else (empty, empty, empty, empty)
        rows.append(dict(group=name, n_pos=res["# pos"], max_recall=res["max_recall"], score=s, recall=r, precision=p, f1=f))
    return rows


def _frame(rows) -> h5.Frame:
    fr = h5.Frame()
    for k in rows[0]:
        vals = [r[k] for r in rows]
        if isinstance(vals[0], (str, np.ndarray)):
            a = np.empty(len(vals), object)
            for i, v in enumerate(vals):
                a[i] = v
            fr[k] = a
        else:
            fr[k] = np.array(vals)
    return fr


def run(argv: list[str]):
    """Calculate precision and recall for compared HDF5"""
    args = parse_args(argv)
    logger.setLevel(getattr(logging, str(args.verbosity).upper(), logging.INFO))
    ignored = args.ignore_filters.split(",")
    skip = list(concordance.SKIP_KEYS_ALL) if args.dataset_key == "all" else []
    df = concordance.read_concordance(args.input_file, key=args.dataset_key, skip_keys=skip)
    score_column = args.score_key.lower()
    if score_column not in df or np.all(np.isnan(np.asarray(df[score_column], dtype=np.float64))):
        df[score_column] = np.ones(df.n_rows)
        logger.warning("No %s field in comparison hdf input, expect invalid recall/precision curves", score_column)
    df["tree_score"] = np.asarray(df[score_column], dtype=np.float64)
    classify_column = "classify" if args.ignore_genotype else "classify_gt"

    acc = calc_accuracy_metrics(df, classify_column, ignored, args.use_for_group_testing)
    curve = calc_recall_precision_curve(df, classify_column, ignored, args.use_for_group_testing)
    if args.device is not None:
        from ..engine import Engine            # fails loudly if the library or the GPU is missing
        with Engine(args.device) as eng:
            perf = calc_performance_curves(df, classify_column, ignored, args.use_for_group_testing, pr_curve=eng.pr_curve)
    else:
        perf = calc_performance_curves(df, classify_column, ignored, args.use_for_group_testing)
    h5.write_hdf(f"{args.output_prefix}.h5", {"optimal_recall_precision": _frame(acc), "recall_precision_curve": _frame(curve),
                                              "performance_curve": _frame(perf)})
    with open(f"{args.output_prefix}.stats.csv", "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(acc[0]), delimiter=";", lineterminator="\n")
        w.writeheader()
        w.writerows(acc)
    with open(f"{args.output_prefix}.thresholds.csv", "w", newline="") as fh:
        w = csv.writer(fh, lineterminator="\n")
        w.writerow(["group", "threshold"])
        for r in curve:
            w.writerow([r["group"], "" if r["threshold"] != r["threshold"] else r["threshold"]])
    if args.output_bed:
        cls = np.asarray(df[classify_column], dtype=object)
        for c in ("tp", "fp", "fn"):
            with open(f"{args.output_prefix}.{c}.bed", "w") as fh:
                for i in np.flatnonzero(cls == c):
                    fh.write(f"{df['chrom'][i]}\t{int(df['pos'][i]) - 1}\t{int(df['pos'][i])}\n")
    return 0


if __name__ == "__main__":
    run(sys.argv[1:])
