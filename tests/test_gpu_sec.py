"""SEC database build + apply on the GPU against oracle/stats.py.

The statistic inside is the reference's (stats_utils.py:12-70; its known answers run through the kernel in
tests/test_gpu_parity.py::test_sec_statistic_kats and through the oracle in tests/test_oracle_golden.py); the database
layout and the decision rule are builder-defined (include/ugvc_mi355x.h), so parity here is kernel == oracle on the same
inputs: integer sums exact, likelihood ratios within 1e-9 relative (f64 lgamma / exp of the device library vs scipy),
flags equal outside that sliver around the threshold."""
import numpy as np
import pytest

from variantcalling_amd import schema as S

pytestmark = pytest.mark.gpu
RF = "rf_model_ignore_gt_incl_hpol_runs"


def _cohort(rng, loci, n_obs, k):
    keys = loci[rng.integers(0, loci.size, n_obs)]
    counts = rng.integers(0, 60, size=(n_obs, k)).astype(np.int32)
    counts[rng.random(n_obs) < 0.1] = 0
    return keys, counts


@pytest.mark.parametrize("k", [2, 3, 5])
def test_build_sums_cohort_observations(engine, k):
    from oracle import stats as st
    rng = np.random.default_rng(k)
    loci = np.unique(rng.integers(1, 1 << 40, 30_000).astype(np.uint64))
    keys, counts = _cohort(rng, loci, 250_000, k)
    uk, ue = engine.sec_db_build(keys, counts)
    ok, oe = st.sec_db_build(keys, counts)
    assert np.array_equal(uk, ok) and np.array_equal(ue, oe)
    one_k, one_e = engine.sec_db_build(keys[:1], counts[:1])
    assert one_k.tolist() == [int(keys[0])] and np.array_equal(one_e, counts[:1])
    none_k, none_e = engine.sec_db_build(keys[:0], counts[:0])
    assert none_k.size == 0 and none_e.shape == (0, k)
    with pytest.raises(RuntimeError, match="non-negative"):
        engine.sec_db_build(keys[:4], -np.ones((4, k), np.int32))
    big = np.full((3, k), 2 ** 30, np.int32)
    with pytest.raises(RuntimeError, match="exceeds int32"):
        engine.sec_db_build(np.array([7, 7, 7], np.uint64), big)


@pytest.mark.parametrize("k,scale", [(3, True), (3, False), (2, True)])
def test_apply_matches_the_oracle(engine, small_callset, frozen_models, k, scale):
    from oracle import stats as st
    from variantcalling_amd.engine import configure
    cs = small_callset
    vt = cs.variants
    configure(engine, cs.ref, cs.runs, cs.tracks, cs.blacklist, frozen_models[RF], "TGCA", 10, 10, True)
    engine.upload_variants(vt)
    rng = np.random.default_rng(11 + k)
    vkeys = vt.keys()
    # cohort: two thirds of the call loci (some seen in many samples), plus loci without calls
    on = np.unique(vkeys[rng.random(vt.n) < 0.66])
    off = np.setdiff1d(np.unique(rng.integers(1, int(vkeys.max()) + 1000, 20_000).astype(np.uint64)), vkeys)
    loci = np.concatenate([on, off])
    keys, counts = _cohort(rng, loci, 6 * loci.size, k)
    # make part of the cohort resemble the calls themselves, so ratios spread over (0, 1]
    like = rng.random(keys.size) < 0.5
    j = np.searchsorted(vkeys, keys[like])
    j[j >= vt.n] = 0
    same = vkeys[j] == keys[like]
    obs = np.stack([vt.ad_ref[j], vt.ad_alt[j], np.maximum(vt.dp[j] - vt.ad_ref[j] - vt.ad_alt[j], 0)], axis=1)[:, :min(k, 3)]
    rows = np.flatnonzero(like)[same]
    counts[rows, :obs.shape[1]] = np.maximum(obs[same] + rng.integers(-2, 3, size=obs[same].shape), 0)
    db_k, db_e = engine.sec_db_build(keys, counts)
    engine.set_sec_db(db_k, db_e)
    thr = 0.05
    ratio, hit = engine.sec_apply(thr, scale_expected=scale)
    o_ratio, o_hit = st.sec_apply(vkeys, vt.dp, vt.ad_ref, vt.ad_alt, db_k, db_e, thr, scale)
    on_db = ~np.isnan(o_ratio)
    assert np.array_equal(np.isnan(ratio), ~on_db) and 0.5 < on_db.mean() < 0.8
    assert np.allclose(ratio[on_db], o_ratio[on_db], rtol=1e-9, atol=1e-300)
    sliver = on_db & (np.abs(o_ratio - thr) <= 1e-8 * thr)
    assert np.array_equal(hit[~sliver], o_hit[~sliver]) and not hit[~on_db].any()
    assert 0.02 < hit.mean() < 0.9                                # both verdicts occur
    # mark: the SEC bit joins the resident flags of a scoring pass, nothing else moves
    base = engine.filter_variants(vt)
    _, hit2 = engine.sec_apply(thr, scale_expected=scale, mark=True)
    res = engine.download_results()
    assert np.array_equal(hit2, hit)
    assert np.array_equal(res.flags & ~np.uint8(S.FLAG_SEC), base.flags) and not (base.flags & S.FLAG_SEC).any()
    assert np.array_equal((res.flags & S.FLAG_SEC) > 0, hit)
    assert np.array_equal(res.filter, base.filter) and np.array_equal(res.tree_score, base.tree_score)


def test_apply_edges_and_errors(engine, small_callset, frozen_models):
    from variantcalling_amd.engine import Engine, configure
    cs = small_callset
    with Engine(0) as fresh:
        configure(fresh, cs.ref, cs.runs, cs.tracks, cs.blacklist, frozen_models[RF], "TGCA", 10, 10, True)
        fresh.upload_variants(cs.variants)
        with pytest.raises(RuntimeError, match="no SEC database"):
            fresh.sec_apply()
        fresh.set_sec_db(np.zeros(0, np.uint64), np.zeros((0, 3), np.int32))          # empty database: nothing is SEC
        ratio, hit = fresh.sec_apply()
        assert np.isnan(ratio).all() and not hit.any()
        with pytest.raises(RuntimeError, match="mark needs"):
            fresh.sec_apply(mark=True)                                               # no scoring pass yet
        with pytest.raises(RuntimeError, match="sorted and unique"):
            fresh.set_sec_db(np.array([5, 5], np.uint64), np.zeros((2, 3), np.int32))
        with pytest.raises(RuntimeError, match="k must be"):
            fresh.set_sec_db(np.array([5], np.uint64), np.zeros((1, 1), np.int32))
        with pytest.raises(RuntimeError, match="min_ratio"):
            fresh.sec_apply(min_ratio=float("nan"))
        # a database of exactly the first / last call locus and of 64- and 65-key sizes (block edges of the lookup)
        vk = np.unique(cs.variants.keys())
        for size in (1, 64, 65, 129):
            pick = np.unique(np.concatenate([vk[:1], vk[-1:], vk[np.linspace(0, vk.size - 1, size).astype(int)]]))
            fresh.set_sec_db(pick, np.ones((pick.size, 3), np.int32))
            ratio, _ = fresh.sec_apply()
            assert np.array_equal(~np.isnan(ratio), np.isin(cs.variants.keys(), pick))
