"""Randomised differential test: every production kernel path against the oracle on callsets whose shape,
side tables, parameters and model are drawn at random (seeded).  Complements the targeted cases of
tests/test_gpu_parity.py: the bar is the same - bit-exact FILTER / flags / RF tree_score, 1e-6 on the XGBoost sigmoid."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RF = "rf_model_ignore_gt_incl_hpol_runs"
XGB = "xgb_model_ignore_gt_incl_hpol_runs"


def _draw(seed):
    from variantcalling_amd import synth
    rng = np.random.default_rng(seed)
    n = int(rng.choice([257, 3_000, 20_000, 45_001]))
    n_contigs = int(rng.choice([1, 2, 7, 24]))
    genome = max(int(rng.choice([400_000, 3_000_000, 25_000_000])) * (1 + n_contigs // 8), 40 * n)
    cs = synth.make_callset(n, genome_len=genome, n_contigs=n_contigs, seed=int(rng.integers(1, 10_000)),
                            snv_only=bool(rng.random() < 0.25))
    vt = cs.variants
    # stretch the integer features past their code tables and through zero on a random subset
    k = max(1, vt.n // 50)
    idx = rng.choice(vt.n, size=k, replace=False)
    vt.dp = vt.dp.copy(); vt.ad_alt = vt.ad_alt.copy(); vt.ad_ref = vt.ad_ref.copy(); vt.gq = vt.gq.copy()
    vt.dp[idx] = rng.choice([0, 1, 200, 5000, 70_000, -7], size=k)
    vt.ad_alt[idx] = rng.choice([0, 3, 500, -1], size=k)
    vt.gq[rng.choice(vt.n, size=k, replace=False)] = rng.choice([0, 99, 255], size=k)
    vt.qual = vt.qual.copy()
    vt.qual[rng.choice(vt.n, size=k, replace=False)] = rng.choice([0.0, np.nan, 1e9, 2999.99, np.inf], size=k).astype(np.float32)
    vt.sor = vt.sor.copy()
    vt.sor[rng.choice(vt.n, size=k, replace=False)] = rng.choice([0.0, np.nan, np.inf, -1.0, 1e-30], size=k).astype(np.float32)
    n_tr = int(rng.integers(0, 4))
    tracks = [cs.tracks[j] for j in rng.permutation(3)[:n_tr]]
    runs = cs.runs if rng.random() < 0.8 else None
    bl = cs.blacklist if rng.random() < 0.7 else None
    if bl is not None and rng.random() < 0.5:
        bl = np.unique(np.concatenate([bl, vt.keys()[:: int(rng.integers(2, 9))]]))
    flow = str(rng.choice(["TGCA", "ACGT", "GTAC", "CATG", "TACG"]))
    hp_len, hp_dist = int(rng.choice([4, 10, 12])), int(rng.choice([0, 1, 10, 25]))
    model = str(rng.choice([RF, RF, XGB]))
    return cs, vt, runs, tracks, bl, flow, hp_len, hp_dist, model


@pytest.mark.parametrize("seed", list(range(40)))
def test_random_configurations(engine, frozen_models, seed):
    from oracle import oracle as O
    from variantcalling_amd.engine import configure
    cs, vt, runs, tracks, bl, flow, hp_len, hp_dist, model = _draw(1000 + seed)
    forests = frozen_models[model]
    if any(f is not None and f.n_features > 17 + len(tracks) for f in forests):
        tracks = list(cs.tracks)                              # the frozen models test all three track features
    configure(engine, cs.ref, runs, tracks, bl, forests, flow, hp_len, hp_dist, True)
    exp = O.filter_variants(vt, cs.ref, runs, tracks, bl, forests, hpol_len=hp_len, hpol_dist=hp_dist, flow_order=flow)
    for path in (0, 128, 512, 256):
        engine.set_kernel_variant(path)
        res = engine.filter_variants(vt)
        what = f"seed {seed} path {path}"
        assert np.array_equal(res.filter, exp.filter), what
        assert np.array_equal(res.flags, exp.flags), what
        if model == RF:
            assert np.array_equal(res.tree_score, exp.tree_score), what
        else:
            assert np.max(np.abs(res.tree_score - exp.tree_score)) <= 1e-6, what
    engine.set_kernel_variant(0)


@pytest.mark.parametrize("seed", list(range(6)))
def test_random_pileups(engine, seed):
    """Pileup tally on random depth profiles (empty loci, single reads, loci far deeper than a wave's LDS span),
    one-sided strands (zero cells of the SOR table), every allele code and the full base-quality range."""
    from oracle import oracle as O
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([1, 63, 64, 65, 1000, 30_000]))
    shape = rng.choice(["poisson", "geometric", "spiky"])
    if shape == "poisson":
        d = rng.poisson(float(rng.choice([0.3, 5, 30, 120])), n)
    elif shape == "geometric":
        d = rng.geometric(0.05, n) - 1
    else:
        d = np.where(rng.random(n) < 0.01, rng.integers(1000, 20_000, n), rng.integers(0, 3, n))
    d = d.astype(np.int64)
    off = np.concatenate([[0], np.cumsum(d)]).astype(np.int64)
    m = int(off[-1])
    allele = rng.choice([0, 1, 2, 3], size=m, p=[0.45, 0.45, 0.08, 0.02]).astype(np.uint16)
    strand = (rng.random(m) < float(rng.choice([0.0, 0.5, 1.0, 0.9]))).astype(np.uint16)
    bq = rng.integers(0, 8192, m).astype(np.uint16) if rng.random() < 0.3 else rng.integers(2, 46, m).astype(np.uint16)
    obs = (allele | (strand << 2) | (bq << 3)).astype(np.uint16)
    got = engine.pileup_tally(off, obs)
    exp = O.pileup_tally(off, obs)
    for k in ("ref_fwd", "ref_rev", "alt_fwd", "alt_rev", "other", "dp", "bq_ref", "bq_alt", "ad_ref", "ad_alt"):
        assert np.array_equal(got[k], exp[k]), (k, seed)
    assert np.array_equal(got["vaf"], exp["vaf"]), seed
    assert np.max(np.abs(got["sor"] - exp["sor"]), initial=0.0) <= 1e-5, seed
