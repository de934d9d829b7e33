"""Score evaluation: accuracy metrics and precision/recall curves from (tree_score, filter, labels).

Consumer side of the hot path: `evaluate_concordance.run`
(ugvc/pipelines/evaluate_concordance.py:71-108) calls `calc_accuracy_metrics` /
`calc_recall_precision_curve` of the absent submodule; the in-tree equivalents this module follows are
`ReportUtils.__calc_performance` (ugvc/reports/report_utils.py:415-505: score direction :435-440, missing
candidates :443-446, post-filter tp/fp/fn :449-457, cumulative curve :494-504), the FN-aware
`precision_recall_curve` (ugvc/utils/stats_utils.py:141-210) and get_precision / get_recall / get_f1
(:76-138).  Host numpy (sort + prefix sums); used by `train_models_pipeline --evaluate_concordance`.
Product code: it never imports the oracle."""
from __future__ import annotations

import numpy as np

EPS = np.finfo(float).eps


def get_precision(fp, tp, if_zero=1.0):
    fp = np.asarray(fp, dtype=np.float64)
    tp = np.asarray(tp, dtype=np.float64)
    den = fp + tp
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(den == 0, if_zero, 1 - fp / np.where(den == 0, 1, den))


def get_recall(fn, tp, if_zero=1.0):
    return get_precision(fn, tp, if_zero)


def get_f1(precision, recall):
    p = np.asarray(precision, dtype=np.float64)
    r = np.asarray(recall, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        f = np.where(p + r == 0, 0.0, 2 * p * r / np.where(p + r == 0, 1, p + r))
    return np.where(np.isnan(p) | np.isnan(r), np.nan, f)


def _binary_pr_curve(y_true: np.ndarray, score: np.ndarray):
    """scikit-learn's precision_recall_curve(y, score) for boolean y: stable descending sort, one point per
    distinct score, precision = tps / (tps + fps), recall = tps / tps[-1], reversed, (1, 0) appended."""
    order = np.argsort(-score, kind="mergesort")
    s = score[order]
    y = y_true[order].astype(np.float64)
    distinct = np.flatnonzero(np.diff(s)) if s.size else np.zeros(0, np.int64)
    idx = np.concatenate([distinct, [s.size - 1]]).astype(np.int64)
    tps = np.cumsum(y)[idx]
    fps = 1 + idx - tps
    ps = tps + fps
    precision = np.zeros_like(tps)
    np.divide(tps, ps, out=precision, where=ps != 0)
    recall = np.ones_like(tps) if tps[-1] == 0 else tps / tps[-1]
    return (np.concatenate([precision[::-1], [1.0]]), np.concatenate([recall[::-1], [0.0]]), s[idx][::-1])


def precision_recall_curve(gtr, predictions, fn_mask, pos_label=1, min_class_counts_to_output=20):
    """stats_utils.py:141-210: calls with fn_mask set are false negatives that never had a score; they
    only scale the recall.  Returns (precisions, recalls, f1, thresholds)."""
    gtr = np.asarray(gtr)
    predictions = np.asarray(predictions, dtype=np.float64)
    fn_mask = np.asarray(fn_mask, dtype=bool)
    if gtr.size == 0:
        return np.array([]), np.array([]), np.array([]), np.array([])
    if np.unique(gtr).size > 2:
        raise ValueError("Only up to two classes of variant labels are possible")
    if fn_mask.size != predictions.size:
        raise ValueError("FN mask should be of the length of predictions")
    sel = gtr[~fn_mask] == pos_label
    pred = predictions[~fn_mask]
    n_fn = fn_mask.sum()
    if sel.size > 0:
        raw_p, raw_r, thr = _binary_pr_curve(sel, pred)
    else:
        raw_p, raw_r, thr = np.array([np.nan, 1.0]), np.array([1.0, 0.0]), np.array([0.0])
    corr = sel.sum() / (sel.sum() + n_fn)
    recalls = (raw_r * corr)[1:-1]
    precisions = raw_p[1:-1]
    thr = thr[1:]
    f1 = 2 * (recalls * precisions) / (recalls + precisions + EPS)
    ps = np.sort(pred)
    cutoff = ps[max(0, ps.size - min_class_counts_to_output)] if ps.size else 0
    keep = ~(thr > cutoff)
    return precisions[keep], recalls[keep], f1[keep], thr[keep]


def calc_performance(score, passed, tp, fp, fn, missing_candidate=None, curve: bool = True, pr_curve=None):
    """report_utils.py:415-505 on arrays.  score: tree_score (NaN allowed); passed: FILTER == PASS;
    tp/fp/fn: boolean classification of every row; missing_candidate: FN rows that had no call at all.
    `pr_curve`: `Engine.pr_curve` - the cumulative curve (:494-504) then runs on the GPU, else on the host.
    Returns (metrics dict, (sorted score, recall, precision, f1) or None)."""
    score = np.asarray(score, dtype=np.float64)
    passed = np.asarray(passed, dtype=bool)
    tp, fp, fn = (np.asarray(x, dtype=bool) for x in (tp, fp, fn))
    miss = np.zeros(score.size, bool) if missing_candidate is None else np.asarray(missing_candidate, dtype=bool)
    ok = ~np.isnan(score)
    sp = score[passed & ok][:20]
    sn = score[~passed & ok][:20]
    with np.errstate(invalid="ignore"):
        dir_switch = 1 if (sp.mean() if sp.size else np.nan) > (sn.mean() if sn.size else np.nan) else -1
    s = score * dir_switch
    s = s - np.nanmin(s) if s.size and ok.any() else s
    s = np.where(miss, -1.0, s)
    filtered_tp = int((tp & ~passed).sum())
    filtered_fp = int((fp & ~passed).sum())
    i_tp, i_fp, i_fn = int(tp.sum()), int(fp.sum()), int(fn.sum())
    r_fp, r_fn, r_tp = i_fp - filtered_fp, i_fn + filtered_tp, i_tp - filtered_tp
    n_miss = int(miss.sum())
    recall = float(get_recall(r_fn, r_tp, np.nan))
    precision = float(get_precision(r_fp, r_tp, np.nan))
    res = {"# pos": i_tp + i_fn, "recall": recall, "precision": precision, "f1": float(get_f1(precision, recall)),
           "max_recall": float(get_recall(n_miss, r_tp + r_fn - n_miss, np.nan)),
           "initial_tp": i_tp, "initial_fp": i_fp, "initial_fn": i_fn, "tp": r_tp, "fp": r_fp, "fn": r_fn,
           "miss_candidate": n_miss}
    if not curve or score.size < 10:
        return res, None
    if pr_curve is not None:
        # the sort + running counts + per-position formulas on the GPU (Engine.pr_curve -> ugvc_pr_curve: stable radix sort
        # of an order-preserving key, one scan, f64 finish): bit-equal to the host statement below (tests/test_gpu_eval.py)
        cls = np.where(tp, 1, np.where(fp, 2, 0)).astype(np.uint8)
        ss, rec, prec, f1 = pr_curve(s, cls, i_tp, i_fp, i_fn)[:4]
        return res, (ss, rec, prec, f1)
    # pandas sort_values' default quicksort leaves the order inside a run of equal scores unspecified; BUILDER-DEFINED:
    # stable (input order), which is also what the GPU curve (Engine.pr_curve: stable radix sort) produces
    order = np.argsort(s, kind="stable")
    ctp = np.cumsum(tp[order])
    cfp = np.cumsum(fp[order])
    c_fn = i_fn + ctp
    c_tp = i_tp - ctp
    c_fp = i_fp - cfp
    rec = get_recall(c_fn, c_tp, np.nan)
    prec = get_precision(c_fp, c_tp, np.nan)
    return res, (s[order], rec, prec, get_f1(prec, rec))


CATEGORIES = ("SNP", "Non-hmer INDEL", "HMER indel <= 4", "HMER indel (4,8)", "HMER indel [8,10]", "HMER indel 11,12",
              "HMER indel > 12", "INDELS", "H-INDELS")


def category_masks(indel, hmer_len):
    """Variant categories of the accuracy table (group names:
    test/resources/system/test_evaluate_concordance/expected.out.stats.csv:1-10; bins report_utils.py:508-538)."""
    indel = np.asarray(indel, dtype=bool)
    h = np.asarray(hmer_len)
    return {"SNP": ~indel, "Non-hmer INDEL": indel & (h == 0), "HMER indel <= 4": indel & (h > 0) & (h <= 4),
            "HMER indel (4,8)": indel & (h > 4) & (h < 8), "HMER indel [8,10]": indel & (h >= 8) & (h <= 10),
            "HMER indel 11,12": indel & (h >= 11) & (h <= 12), "HMER indel > 12": indel & (h > 12),
            "INDELS": indel, "H-INDELS": indel & (h > 0)}


def category_bits(indel, hmer_len) -> np.ndarray:
    """u16 per row: bit c set = the row belongs to CATEGORIES[c] (the layout Engine.eval_counts takes)."""
    bits = np.zeros(np.asarray(indel).shape[0], np.uint16)
    for c, m in enumerate(category_masks(indel, hmer_len).values()):
        bits |= (m.astype(np.uint16) << np.uint16(c))
    return bits


def accuracy_rows(counts) -> list:
    """Accuracy-table rows from integer counts[c] = (true, false, true & passing, false & passing[, missed]) per
    category - the counts come from the host (`accuracy_table`) or from the GPU (`Engine.eval_counts`).  `missed`
    (truth variants without a call, classify == "fn") adds to fn before and after filtering; a filtered true call
    becomes a false negative (report_utils.py:449-457).  Column set, order and the 5-digit rounding:
    test/resources/system/test_evaluate_concordance/expected.out.stats.csv."""
    rows = []
    for c, name in enumerate(CATEGORIES):
        tp0, fp0, tp1, fp1 = (int(x) for x in counts[c][:4])
        fn0 = int(counts[c][4]) if len(counts[c]) > 4 else 0
        fn1 = fn0 + tp0 - tp1
        p1, r1 = float(get_precision(fp1, tp1)), float(get_recall(fn1, tp1))
        p0, r0 = float(get_precision(fp0, tp0)), float(get_recall(fn0, tp0))
        rows.append(dict(group=name, tp=tp1, fp=fp1, fn=fn1, precision=round(p1, 5), recall=round(r1, 5),
                         f1=round(float(get_f1(p1, r1)), 5), initial_tp=tp0, initial_fp=fp0, initial_fn=fn0,
                         initial_precision=round(p0, 5), initial_recall=round(r0, 5),
                         initial_f1=round(float(get_f1(p0, r0)), 5)))
    return rows


def accuracy_table(score, passed, label_tp, indel, hmer_len, ignored_pass=None):
    """Per-category tp/fp/fn/precision/recall/f1 before ('initial_*') and after filtering for scored CALLS
    with a true/false label (no un-called truth variants: fn counts only filtered true calls).
    `ignored_pass`: rows whose only filter is an ignored one (HPOL_RUN,
    evaluate_concordance.py:44-48) count as passing."""
    passed = np.asarray(passed, bool) if ignored_pass is None else (np.asarray(passed, bool) | np.asarray(ignored_pass, bool))
    label_tp = np.asarray(label_tp, bool)
    counts = []
    for m in category_masks(indel, hmer_len).values():
        counts.append((int((label_tp & m).sum()), int((~label_tp & m).sum()),
                       int((label_tp & m & passed).sum()), int((~label_tp & m & passed).sum())))
    return accuracy_rows(counts)
