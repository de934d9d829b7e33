"""Collected first (file name): is this GPU usable at all?  A copy round trip and the library's smallest kernel, checked on the
host, in front of every parity test - a dead box fails HERE with the step named, a kernel bug fails later with the kernel named
(conftest.py turns the launch breadcrumbs on for every GPU test)."""
import numpy as np
import pytest


@pytest.mark.gpu
def test_device_canary_copies_and_computes(engine):
    info = engine.device_info()
    print("canary device:", info, flush=True)
    assert "gfx950" in info["name"], f"not a gfx950 device (the library holds gfx950 code only): {info}"
    assert info["n_cus"] >= 1 and info["hbm_bytes"] > 0          # (a partitioned device - fewer CUs, less memory - is still a usable one)
    # (the large sizes: copies above 4 MB are split over host threads - 8 n + n % 7 bytes with n / 4 not a whole number of
    # 64-byte lines, the shape that lost a buffer's last two bytes in the first version of the pinned-slot copies)
    for n in (1, 63, 64, 65, 1024, 100_003, 524_289, 779_936, (1 << 20) + 3, (2 << 20) + 5):
        engine.selftest(n)


@pytest.mark.gpu
def test_second_context_canary():
    """a fresh context beside the session's one: creation, the canary and destruction leave the first one usable"""
    from variantcalling_amd.engine import Engine
    with Engine(0) as e2:
        e2.selftest(4096)


@pytest.mark.gpu
def test_smallest_scoring_pass_after_canary(engine, frozen_models):
    """the smallest callsets through the production pass right behind the canary, every launch named and waited for: if
    the first real kernel faults, the log says which"""
    import os
    from oracle import oracle as O
    from variantcalling_amd import synth
    from variantcalling_amd.engine import configure
    forests = frozen_models["rf_model_ignore_gt_incl_hpol_runs"]
    old = os.environ.get("UGVC_DEBUG_SYNC")
    os.environ["UGVC_DEBUG_SYNC"] = "1"
    try:
        for n, contigs in ((1, 1), (65, 3), (1025, 3)):
            cs = synth.make_callset(n, genome_len=2_000_000, n_contigs=contigs, seed=100 + n)
            configure(engine, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
            got = engine.filter_variants(cs.variants)
            exp = O.filter_variants(cs.variants, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
            assert np.array_equal(got.filter, exp.filter) and np.array_equal(got.flags, exp.flags)
            assert np.array_equal(got.tree_score, exp.tree_score)
    finally:
        if old is None:
            del os.environ["UGVC_DEBUG_SYNC"]
        else:
            os.environ["UGVC_DEBUG_SYNC"] = old


@pytest.mark.gpu
@pytest.mark.parametrize("n_contigs", [1, 3, 40])
def test_small_callset_matrix(engine, frozen_models, n_contigs):
    """The small-callset regime, where a launch is a handful of workgroups, most tiles are partial and (with 40 contigs) nearly
    every tile ends at a contig boundary: 1 ... 20 000 rows x 1 / 3 / 40 contigs through the production pass, the three-launch
    pass and the universal kernel, every row against the oracle.  (tools/gpu_suite.sh runs this file under the guard-page and
    poison modes too: VERDICT r4 item 1.)"""
    from oracle import oracle as O
    from variantcalling_amd import synth
    from variantcalling_amd.engine import configure
    forests = frozen_models["rf_model_ignore_gt_incl_hpol_runs"]
    try:
        for n in (1, 63, 64, 65, 1023, 1025, 20_000):
            cs = synth.make_callset(n, genome_len=max(400_000, 300 * n), n_contigs=n_contigs, seed=1000 * n_contigs + n)
            exp = O.filter_variants(cs.variants, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
            configure(engine, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
            for name, variant in (("v5", 0), ("v3", 65536), ("v1", 256)):
                engine.set_kernel_variant(variant)
                got = engine.filter_variants(cs.variants)
                what = f"{name}: {cs.variants.n} rows, {n_contigs} contigs"
                assert np.array_equal(got.filter, exp.filter), what
                assert np.array_equal(got.flags, exp.flags), what
                assert np.array_equal(got.tree_score, exp.tree_score), what
            X, group = engine.feature_matrix(cs.variants)
            ft = O.featurize(cs.variants, cs.ref, cs.runs, cs.tracks)
            assert np.array_equal(X, ft["X"]) and np.array_equal(group, ft["group"]), f"feature matrix: {cs.variants.n} rows, {n_contigs} contigs"
    finally:
        engine.set_kernel_variant(0)
