"""GPU score evaluation against the host numpy statement in variantcalling_amd/evaluate.py
(itself pinned to the reference's precision_recall_curve / get_precision / get_f1: tests/test_io_host.py):
category counts on the resident FILTER column, and the cumulative recall / precision / f1 curve - bit-exact f64."""
import os

import numpy as np
import pytest

from variantcalling_amd import evaluate, schema as S

pytestmark = pytest.mark.gpu
RF = "rf_model_ignore_gt_incl_hpol_runs"


def test_eval_counts_on_the_resident_filter_column(engine, small_callset, frozen_models):
    from variantcalling_amd.engine import configure
    cs = small_callset
    configure(engine, cs.ref, cs.runs, cs.tracks, cs.blacklist, frozen_models[RF], "TGCA", 10, 10, True)
    res = engine.filter_variants(cs.variants)
    X, group = engine.feature_matrix()
    engine.filter_resident()                                   # the feature-matrix pass does not score: FILTER again
    n = cs.variants.n
    rng = np.random.default_rng(4)
    label = rng.integers(-1, 2, n).astype(np.int8)              # -1 unlabelled
    names = S.feature_names(len(cs.tracks))
    indel = group != S.GROUP_SNP
    hmer = X[:, names.index("hmer_indel_length")]
    bits = evaluate.category_bits(indel, hmer)
    got = engine.eval_counts(label, bits)
    sel = label >= 0
    rows_host = evaluate.accuracy_table(res.tree_score[sel], res.filter[sel] == S.FILTER_PASS, label[sel] == 1,
                                        indel[sel], hmer[sel])
    assert evaluate.accuracy_rows(got) == rows_host
    assert got[9:].sum() == 0 and got[0].sum() > 0 and got[7, 0] == got[1:7, 0].sum()
    with pytest.raises(ValueError):
        engine.eval_counts(label[:-1], bits[:-1])


@pytest.mark.parametrize("n", [10, 1000, 200_003])
def test_pr_curve_is_the_host_curve_bit_for_bit(engine, n):
    rng = np.random.default_rng(n)
    tp = rng.random(n) < 0.55
    fn = ~tp & (rng.random(n) < 0.25)
    fp = ~tp & ~fn
    # f32-valued scores with many ties (RF means of 40 leaf fractions), some NaN, some missing candidates
    score = (rng.integers(0, 41, n) / 40.0 * (0.5 + 0.5 * tp)).astype(np.float32).astype(np.float64)
    score[rng.random(n) < 0.01] = np.nan
    passed = np.nan_to_num(score) > 0.4
    miss = fn & (rng.random(n) < 0.5)
    res, curve = evaluate.calc_performance(score, passed, tp, fp, fn, missing_candidate=miss)
    # the same pre-processing calc_performance applies before its sort (direction, shift, missing -> -1)
    ok = ~np.isnan(score)
    sp, sn = score[passed & ok][:20], score[~passed & ok][:20]
    d = 1 if (sp.mean() if sp.size else np.nan) > (sn.mean() if sn.size else np.nan) else -1
    s = score * d
    s = s - np.nanmin(s)
    s = np.where(miss, -1.0, s)
    cls = np.where(tp, 1, np.where(fp, 2, 0)).astype(np.uint8)
    gs, grec, gprec, gf1, order, ms = engine.pr_curve(s, cls, res["initial_tp"], res["initial_fp"], res["initial_fn"],
                                                      want_order=True)
    hs, hrec, hprec, hf1 = curve
    assert np.array_equal(order, np.argsort(s, kind="stable"))
    for g, h, nm in ((gs, hs, "score"), (grec, hrec, "recall"), (gprec, hprec, "precision"), (gf1, hf1, "f1")):
        assert np.array_equal(g.view(np.uint64), np.asarray(h, np.float64).view(np.uint64)) or \
            np.array_equal(g, h, equal_nan=True), nm
        assert np.array_equal(np.isnan(g), np.isnan(h)), nm
    assert ms >= 0.0


def test_pr_curve_edges(engine):
    z = np.zeros(0)
    out = engine.pr_curve(z, z.astype(np.uint8), 0, 0, 0)
    assert all(o.size == 0 for o in out[:4])
    # all equal scores, negative zero, infinities: stable order = input order
    s = np.array([0.0, -0.0, np.inf, -np.inf, 0.0, np.nan, -1.0, 0.0])
    cls = np.array([1, 2, 1, 2, 0, 1, 2, 1], np.uint8)
    gs, rec, prec, f1, order, _ = engine.pr_curve(s, cls, 4, 3, 1, want_order=True)
    assert order.tolist() == np.argsort(s, kind="stable").tolist()
    ctp, cfp = np.cumsum(cls[order] == 1), np.cumsum(cls[order] == 2)
    assert np.array_equal(rec, evaluate.get_recall(1 + ctp, 4 - ctp, np.nan), equal_nan=True)
    assert np.array_equal(prec, evaluate.get_precision(3 - cfp, 4 - ctp, np.nan), equal_nan=True)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 256, 257, 4095, 4096, 4097, 8191, 8193, 12289, 70_001])
def test_radix_sort_and_scan_at_tile_edges(engine, n):
    """The library's own sort / scan (csrc/kernels_prims.hip) around their tile sizes (256-key rounds, 4096-key blocks, scan
    tiles of 4096 words): full-range f64 keys - every digit position differs, no pass is skipped - with duplicates, both
    zeros, infinities and NaNs; the order must be numpy's stable order and the running counts exact."""
    rng = np.random.default_rng(n)
    s = rng.standard_normal(n) * 10.0 ** rng.integers(-300, 300, n)
    dup = rng.random(n) < 0.2
    if dup.any():
        s[dup] = rng.choice(s, int(dup.sum()))                  # runs of equal keys: the order inside them is the input order
    if n > 8:
        s[:8] = [0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, 1.0, -1.0]
    cls = rng.choice(np.array([0, 1, 2], np.uint8), n)
    i_tp, i_fp = int((cls == 1).sum()), int((cls == 2).sum())
    gs, grec, gprec, gf1, order, _ = engine.pr_curve(s, cls, i_tp, i_fp, 7, want_order=True)
    # numpy sorts -0.0 and 0.0 as equal and NaN last; stable
    want = np.argsort(s, kind="stable")
    assert np.array_equal(order, want.astype(np.int32))
    assert np.array_equal(gs, s[want], equal_nan=True)
    ctp, cfp = np.cumsum(cls[want] == 1), np.cumsum(cls[want] == 2)
    rec = evaluate.get_recall(7 + ctp, i_tp - ctp, np.nan)
    prec = evaluate.get_precision(i_fp - cfp, i_tp - ctp, np.nan)
    assert np.array_equal(grec, rec, equal_nan=True) and np.array_equal(gprec, prec, equal_nan=True)
    assert np.array_equal(gf1, evaluate.get_f1(prec, rec), equal_nan=True)
