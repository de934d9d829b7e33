"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, bit-exact for
FILTER / flags / every integer feature / RF tree_score; stated tolerances for the two
floating-point outputs that go through device libm (GBT sigmoid: 1e-6 abs, SOR: 1e-5 abs)."""
import os

import numpy as np
import pytest

import edge_cases as E
from conftest import GOLDEN, real_chr1_reference

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RF = "rf_model_ignore_gt_incl_hpol_runs"
XGB = "xgb_model_ignore_gt_incl_hpol_runs"


def _oracle():
    from oracle import oracle as O
    return O


def _configure(engine, cs_ref, runs, tracks, bl, forests, hpol_len=10, hpol_dist=10, flow="TGCA"):
    from variantcalling_amd.engine import configure
    configure(engine, cs_ref, runs, tracks, bl, forests, flow, hpol_len, hpol_dist, True)


def _assert_same(res, exp, what=""):
    assert np.array_equal(res.filter, exp.filter), f"FILTER mismatch {what}: {np.flatnonzero(res.filter != exp.filter)[:10]}"
    assert np.array_equal(res.flags, exp.flags), f"flags mismatch {what}: {np.flatnonzero(res.flags != exp.flags)[:10]}"
    assert np.array_equal(res.tree_score, exp.tree_score), f"TREE_SCORE mismatch {what}"


def test_library_is_loaded_in_tree(engine):
    from variantcalling_amd import engine as eng_mod
    assert os.path.dirname(eng_mod.LIB_PATH).endswith("variantcalling_amd")
    info = engine.device_info()
    assert "gfx950" in info["name"], info
    with open("/proc/self/maps") as fh:
        assert "libugvc_mi355x.so" in fh.read()


# kernel paths (include/ugvc_mi355x.h, ugvc_set_kernel_variant): 0 = production (v5 when the models allow it),
# 65536 = v3 kernels, 256 = v1 universal kernel
PATHS = pytest.mark.parametrize("path", [0, 65536, 256], ids=["v5-fused", "v3-lockstep", "v1-universal"])


@pytest.fixture(autouse=True)
def _reset_variant(request):
    yield
    if "engine" in request.fixturenames:
        request.getfixturevalue("engine").set_kernel_variant(0)


@PATHS
def test_filter_synthetic_rf(engine, small_callset, frozen_models, path):
    cs = small_callset
    O = _oracle()
    _configure(engine, cs.ref, cs.runs, cs.tracks, cs.blacklist, frozen_models[RF])
    engine.set_kernel_variant(path)
    res = engine.filter_variants(cs.variants)
    exp = O.filter_variants(cs.variants, cs.ref, cs.runs, cs.tracks, cs.blacklist, frozen_models[RF])
    _assert_same(res, exp, "synthetic C3-shaped")
    assert 0 < (res.filter == 0).mean() < 1


def test_feature_matrix_bit_exact(engine, small_callset, frozen_models):
    cs = small_callset
    O = _oracle()
    _configure(engine, cs.ref, cs.runs, cs.tracks, cs.blacklist, frozen_models[RF])
    X, g = engine.feature_matrix(cs.variants)
    ft = O.featurize(cs.variants, cs.ref, cs.runs, cs.tracks)
    assert X.shape == ft["X"].shape
    for j, name in enumerate(__import__("variantcalling_amd.schema", fromlist=["x"]).feature_names(3)):
        assert np.array_equal(X[:, j], ft["X"][:, j]), name
    assert np.array_equal(g, ft["group"].astype(np.uint8))


def test_feature_matrix_with_a_model_too_large_for_the_fused_kernel(engine, small_callset, frozen_models):
    """a group-0 forest that fits the v3 kernels' LDS budget but not the fused kernel's (64 complete trees of depth
    8 with few distinct payloads) sends the SCORING pass to v3 - and must not make the feature matrix, which needs no model,
    fail: the feature-matrix launch leaves the forest out of its LDS budget."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_fuzz import _random_forest
    from variantcalling_amd import model_io, schema as S
    cs = small_callset
    O = _oracle()
    rng = np.random.default_rng(77)
    # (coarse fractions: a handful of distinct payloads; the dense tables are sized by trees x 2^depth whatever the trees' shape)
    big = _random_forest(rng, S.MODEL_RF, 17 + len(cs.tracks), 64, 8, normalised="coarse")
    big.max_depth = max(model_io._depth(big.left, big.right, big.feature, int(r)) for r in big.tree_root)
    assert big.max_depth == 8 and big.n_trees == 64
    forests = [big, frozen_models[RF][1], frozen_models[RF][2]]
    _configure(engine, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
    X, g = engine.feature_matrix(cs.variants)                      # used to return -1: "the SNP forest does not fit the fused kernel's LDS"
    ft = O.featurize(cs.variants, cs.ref, cs.runs, cs.tracks)
    assert np.array_equal(X, ft["X"]) and np.array_equal(g, ft["group"].astype(np.uint8))
    res = engine.filter_variants(cs.variants)                      # whatever path takes it (v3 when the fused kernel cannot hold the forest)
    _assert_same(res, O.filter_variants(cs.variants, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests), "large group-0 forest")


def _hip_device_count() -> int:
    import ctypes
    n = ctypes.c_int(0)
    try:
        ctypes.CDLL("libamdhip64.so").hipGetDeviceCount(ctypes.byref(n))
    except OSError:
        return 0
    return n.value


def test_v3_path_on_the_second_device(small_callset, frozen_models):
    """the v3 kernels' dynamic-LDS attribute is set per DEVICE (kernels_v3.hip) - a process that scores on device 1
    after device 0 used to launch with the default limit there.  Needs two GPUs."""
    if _hip_device_count() < 2:
        pytest.skip("needs two GPUs")
    from variantcalling_amd.engine import Engine
    cs = small_callset
    O = _oracle()
    exp = O.filter_variants(cs.variants, cs.ref, cs.runs, cs.tracks, cs.blacklist, frozen_models[RF])
    for dev in (0, 1):
        with Engine(dev) as eng:
            _configure(eng, cs.ref, cs.runs, cs.tracks, cs.blacklist, frozen_models[RF])
            for path in (65536, 0):
                eng.set_kernel_variant(path)
                _assert_same(eng.filter_variants(cs.variants), exp, f"device {dev} path {path}")


@PATHS
def test_filter_edge_cases_real_hg38(engine, frozen_models, path):
    """Contig ends, N runs, 50 kb 'N homopolymers', MNPs, 60 bp indels, dp = 0, empty track contig."""
    O = _oracle()
    ref = real_chr1_reference()
    vt = E.edge_table(ref)
    runs, tracks = E.simple_tracks(ref)
    bl = np.unique(np.concatenate([vt.keys()[::7], vt.keys()[::11] + np.uint64(1)]))
    for flow in ("TGCA", "ACGT", "GTAC"):
        _configure(engine, ref, runs, tracks, bl, frozen_models[RF], hpol_len=8, hpol_dist=12, flow=flow)
        engine.set_kernel_variant(path)
        res = engine.filter_variants(vt)
        exp = O.filter_variants(vt, ref, runs, tracks, bl, frozen_models[RF], flow_order=flow, hpol_len=8, hpol_dist=12)
        _assert_same(res, exp, f"edge flow={flow}")
        X, _ = engine.feature_matrix()
        assert np.array_equal(X, O.featurize(vt, ref, runs, tracks, flow, 8, 12)["X"])


@PATHS
def test_snv_only_c2_shape(engine, frozen_models, path):
    from variantcalling_amd import synth
    O = _oracle()
    cs = synth.make_callset(50_000, genome_len=20_000_000, n_contigs=4, seed=77, snv_only=True)
    _configure(engine, cs.ref, cs.runs, cs.tracks, cs.blacklist, frozen_models[RF])
    engine.set_kernel_variant(path)
    _assert_same(engine.filter_variants(cs.variants),
                 O.filter_variants(cs.variants, cs.ref, cs.runs, cs.tracks, cs.blacklist, frozen_models[RF]), "C2")


@PATHS
def test_gbt_model(engine, small_callset, frozen_models, path):
    """XGBoost-shaped ensemble: FILTER bit-exact (decided on the f32 margin), score within 1e-6."""
    cs = small_callset
    O = _oracle()
    _configure(engine, cs.ref, cs.runs, cs.tracks, cs.blacklist, frozen_models[XGB])
    engine.set_kernel_variant(path)
    res = engine.filter_variants(cs.variants)
    exp = O.filter_variants(cs.variants, cs.ref, cs.runs, cs.tracks, cs.blacklist, frozen_models[XGB])
    assert np.array_equal(res.filter, exp.filter)
    assert np.array_equal(res.flags, exp.flags)
    assert np.max(np.abs(res.tree_score - exp.tree_score)) <= 1e-6


@pytest.mark.parametrize("path", [0, 65536], ids=["v5-fused", "v3-lockstep"])
def test_v3_fallbacks_and_dense_tiles(engine, frozen_models, path):
    """v5 / v3 preconditions: (i) a tile whose side-table slices overflow the staged LDS slices takes the
    HBM search path; (ii) overlapping intervals with sorted ends stay on the fast paths; (iii) a track whose
    ends are not sorted is rejected, overlapping runs silently select the universal kernel - results equal the oracle."""
    from variantcalling_amd import schema as S, synth
    O = _oracle()
    cs = synth.make_callset(30_000, genome_len=3_000_000, n_contigs=2, seed=31)
    ref, vt = cs.ref, cs.variants
    rng = np.random.default_rng(8)

    def track(starts, ends, contig_of):
        order = np.lexsort((starts, contig_of))
        starts, ends, contig_of = starts[order], ends[order], contig_of[order]
        ptr = np.searchsorted(contig_of, np.arange(ref.n_contigs + 1)).astype(np.int32)
        return S.IntervalTrack(starts.astype(np.int32), ends.astype(np.int32), ptr, "t")

    L0 = ref.contig_len(0)
    # (i) 40k tiny intervals packed into 200 kb of contig 0: > 2048 staged entries per tile there
    st = np.sort(rng.choice(np.arange(100_000, 300_000), size=40_000, replace=False))
    dense = track(st, st + 1, np.zeros(st.size, np.int64))
    # (ii) overlapping but with sorted ends
    st2 = np.sort(rng.integers(0, L0 - 2000, size=4000))
    en2 = np.maximum.accumulate(st2 + rng.integers(1, 1500, size=4000))
    overl = track(st2, en2, np.zeros(4000, np.int64))
    bl = np.unique(np.concatenate([vt.keys()[::3], cs.blacklist]))
    for tracks in ([dense, cs.tracks[0], cs.tracks[2]], [overl, dense, cs.tracks[1]]):
        _configure(engine, ref, cs.runs, tracks, bl, frozen_models[RF])
        engine.set_kernel_variant(path)
        got = engine.filter_variants(vt)
        exp = O.filter_variants(vt, ref, cs.runs, tracks, bl, frozen_models[RF])
        _assert_same(got, exp, "dense/overlapping tracks on v3")
        assert (got.flags >> 3).any()
    # (iii) nested intervals (ends not sorted) are rejected at upload; overlapping runs select the universal kernel
    en3 = st2 + rng.integers(1, 3000, size=4000)
    nested = track(st2, en3, np.zeros(4000, np.int64))
    assert (np.diff(nested.ends) < 0).any()
    with pytest.raises(RuntimeError, match="not sorted"):
        engine.set_tracks([nested, cs.tracks[1], cs.tracks[2]])
    ov_runs = track(st2, st2 + 400, np.zeros(4000, np.int64))
    _configure(engine, ref, ov_runs, cs.tracks, None, frozen_models[RF], hpol_len=10, hpol_dist=25)
    _assert_same(engine.filter_variants(vt),
                 O.filter_variants(vt, ref, ov_runs, cs.tracks, None, frozen_models[RF], hpol_len=10, hpol_dist=25),
                 "overlapping runs")


@pytest.mark.parametrize("world", [2, 8])
def test_sliced_context_equals_full(engine, small_callset, frozen_models, world):
    """What a rank of a multi-GPU run uploads - its shard plus the slices of the genome / interval tables / blacklist the
    shard can touch, re-based (shard.slice_context) - scores exactly like the shard against the full tables."""
    from variantcalling_amd import shard
    cs = small_callset
    _configure(engine, cs.ref, cs.runs, cs.tracks, cs.blacklist, frozen_models[RF])
    whole = engine.filter_variants(cs.variants)
    b = shard.shard_bounds(cs.variants.n, world)
    for r in (0, world - 1):
        mine = shard.shard_of(cs.variants, r, world)
        ref_s, runs_s, tracks_s, bl_s, mine_s = shard.slice_context(cs.ref, cs.runs, cs.tracks, cs.blacklist, mine)
        _configure(engine, ref_s, runs_s, tracks_s, bl_s, frozen_models[RF])
        got = engine.filter_variants(mine_s)
        lo, hi = int(b[r]), int(b[r + 1])
        assert np.array_equal(got.flags, whole.flags[lo:hi]) and np.array_equal(got.filter, whole.filter[lo:hi])
        assert np.array_equal(got.tree_score, whole.tree_score[lo:hi])


def test_v5_many_small_contigs_and_row_counts(engine, frozen_models):
    """The v5 pass carries a wave's table ranks from tile to tile and searches afresh at contig changes: 400 contigs of a
    few dozen variants each (most tiles span contigs), and callset sizes around the workgroup / tile granularity."""
    from variantcalling_amd import synth
    O = _oracle()
    cs = synth.make_callset(24_000, genome_len=6_000_000, n_contigs=400, seed=77)
    _configure(engine, cs.ref, cs.runs, cs.tracks, cs.blacklist, frozen_models[RF])
    exp = O.filter_variants(cs.variants, cs.ref, cs.runs, cs.tracks, cs.blacklist, frozen_models[RF])
    _assert_same(engine.filter_variants(cs.variants), exp, "400 contigs")
    for n in (1, 63, 64, 65, 1023, 1024, 1025, 2049, 4097):
        sub = cs.variants.slice(5000, 5000 + n)
        got = engine.filter_variants(sub)
        assert np.array_equal(got.flags, exp.flags[5000:5000 + n]) and np.array_equal(got.filter, exp.filter[5000:5000 + n]), n
        assert np.array_equal(got.tree_score, exp.tree_score[5000:5000 + n]), n


@pytest.mark.parametrize("n, n_contigs, seed", [(70_000, 2, 5), (150_001, 3, 6), (40_000, 7, 7), (300_000, 2, 8)])
def test_v5_tile_grid_restarts_at_a_contig_boundary(engine, frozen_models, n, n_contigs, seed):
    """Round 4: a workgroup whose rows cross a contig boundary finds the second contig's first row in its prologue, restarts
    the tile grid of both lists there and hands the second contig's table state over in LDS (fused5_kernel: has_b, brk_publish).
    One boundary per workgroup at most with two or three contigs, several with seven; sub-ranges move the boundary through
    every alignment of the 64-entry grid and through every wave of the workgroup."""
    from variantcalling_amd import synth
    O = _oracle()
    cs = synth.make_callset(n, genome_len=40_000_000, n_contigs=n_contigs, seed=seed)
    _configure(engine, cs.ref, cs.runs, cs.tracks, cs.blacklist, frozen_models[RF])
    exp = O.filter_variants(cs.variants, cs.ref, cs.runs, cs.tracks, cs.blacklist, frozen_models[RF])
    _assert_same(engine.filter_variants(cs.variants), exp, f"{n_contigs} contigs")
    first = int(np.flatnonzero(cs.variants.contig != cs.variants.contig[0])[0])      # first row of the second contig
    for lo, hi in [(first - 1, first + 1), (first - 64, first + 64), (first - 1000, first + 37), (first - 37, first + 3000),
                   (first, first + 500), (max(first - 20_000, 0), min(first + 20_000, cs.variants.n)), (0, first), (0, first + 1)]:
        lo, hi = max(lo, 0), min(hi, cs.variants.n)
        got = engine.filter_variants(cs.variants.slice(lo, hi))
        assert np.array_equal(got.flags, exp.flags[lo:hi]) and np.array_equal(got.filter, exp.filter[lo:hi]), (lo, hi)
        assert np.array_equal(got.tree_score, exp.tree_score[lo:hi]), (lo, hi)


@pytest.mark.parametrize("n_tracks", [0, 1, 2, 4, 5])
@PATHS
def test_track_counts_other_than_three(engine, small_callset, path, n_tracks):
    """The v5 kernels are instantiated per number of annotation tracks (0..5): every count, a dense track among them,
    with a random forest per variant type over the 17 + n_tracks features of that configuration."""
    from test_gpu_fuzz import _random_forest
    from variantcalling_amd import model_io, schema as S, synth
    O = _oracle()
    cs = small_callset
    extra = [synth.make_interval_track(cs.ref, 4000, 500.0, 991, "extra.a"), synth.make_interval_track(cs.ref, 60_000, 150.0, 992, "extra.b")]
    tracks = ([cs.tracks[2], extra[1]] + list(cs.tracks[:2]) + [extra[0]])[:n_tracks]
    rng = np.random.default_rng(4200 + n_tracks)
    forests = [_random_forest(rng, S.MODEL_RF, 17 + n_tracks, 24, 7, normalised=True) for _ in range(3)]
    for f in forests:
        f.max_depth = max(model_io._depth(f.left, f.right, f.feature, int(r)) for r in f.tree_root)
    _configure(engine, cs.ref, cs.runs, tracks, cs.blacklist, forests)
    engine.set_kernel_variant(path)
    _assert_same(engine.filter_variants(cs.variants), O.filter_variants(cs.variants, cs.ref, cs.runs, tracks, cs.blacklist, forests),
                 f"{n_tracks} tracks")


@pytest.mark.parametrize("rounds", [False, True], ids=["one-round", "pair-rounds"])
@pytest.mark.parametrize("iwide", ["0", "15"], ids=["narrow-slices", "wide-slices"])
def test_v5_indel_slice_widths_agree(engine, small_callset, frozen_models, iwide, rounds, monkeypatch):
    """Indel tiles stage two rows per lane of a sparse table and six of a dense one (chosen from the table sizes); a
    slice that does not reach the tile's last variant falls back to a search in HBM.  Forcing every table narrow (the
    dense track then takes the fallback on most tiles) or wide, and staging the narrow slices all at once or in pairs,
    must not change a bit."""
    O = _oracle()
    cs = small_callset
    _configure(engine, cs.ref, cs.runs, cs.tracks, cs.blacklist, frozen_models[RF])
    exp = O.filter_variants(cs.variants, cs.ref, cs.runs, cs.tracks, cs.blacklist, frozen_models[RF])
    monkeypatch.setenv("UGVC_IWIDE", iwide)
    if rounds:
        monkeypatch.setenv("UGVC_INDEL_ROUNDS", "1")          # narrow slices staged two tables at a time (the smaller scratch)
    _assert_same(engine.filter_variants(cs.variants), exp, f"UGVC_IWIDE={iwide}")
    assert engine.filter_variants(cs.variants).filter.tobytes() == exp.filter.tobytes()      # second pass: the other counter set


def test_empty_no_tables_and_ragged(engine, small_callset, frozen_models):
    from variantcalling_amd.engine import Engine
    O = _oracle()
    cs = small_callset
    with Engine(0) as e2:      # fresh context: no runs, no tracks, no blacklist, only an SNP model
        e2.set_reference(cs.ref)
        e2.set_tracks([])
        e2.set_blacklist(None)
        forests17 = None
        res = e2.filter_variants(cs.variants.slice(0, 0))
        assert res.filter.size == 0
        sub = cs.variants.slice(1000, 1777)          # not a multiple of the block size
        res = e2.filter_variants(sub)
        exp = O.filter_variants(sub, cs.ref, None, [], None, [None, None, None], mark_hpol=False)
        _assert_same(res, exp, "no tables / no model")
        assert np.all(res.tree_score == 0) and np.all(res.filter == 0)
        X, g = e2.feature_matrix()
        assert np.array_equal(X, O.featurize(sub, cs.ref, None, [])["X"])


def test_error_paths(engine, small_callset, frozen_models):
    import copy
    cs = small_callset
    _configure(engine, cs.ref, cs.runs, cs.tracks, cs.blacklist, frozen_models[RF])     # (contigs known: also when run alone)
    vt = copy.copy(cs.variants.slice(0, 100))
    vt.pos = vt.pos[::-1].copy()
    with pytest.raises(RuntimeError, match="sorted"):
        engine.filter_variants(vt)
    vt = copy.copy(cs.variants.slice(0, 100))
    vt.contig = np.full(100, 200, np.uint16)
    with pytest.raises(RuntimeError, match="contig index"):
        engine.filter_variants(vt)
    with pytest.raises(RuntimeError, match="permutation"):
        engine.set_flow_order("AAGT")


def test_reserve_changes_nothing_but_the_first_calls_allocations(frozen_models):
    """ugvc_reserve (round 4): the boundary call's one-time allocations from a helper thread, beside the uploads of the
    reference / tables / model on the calling thread (how filter_variants_pipeline uses it) - smaller, exact and larger sizes
    than the callset that follows; small callsets (plain upload path) and a second reserve are no-ops; bad sizes are errors."""
    import threading
    from variantcalling_amd import synth
    from variantcalling_amd.engine import Engine
    O = _oracle()
    cs = synth.make_callset(300_011, genome_len=150_000_000, n_contigs=5, seed=99)
    forests = frozen_models[RF]
    exp = O.filter_variants(cs.variants, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
    for n_res in (270_000, cs.variants.n, 700_000, 1_000):
        with Engine(0) as eng:
            box = {}

            def work():
                try:
                    eng.reserve(n_res, int(cs.variants.alleles.size))
                    box["ok"] = True
                except Exception as e:                       # noqa: BLE001 - reported by the assert below
                    box["err"] = e

            t = threading.Thread(target=work)
            t.start()
            _configure(eng, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
            t.join()
            assert box == {"ok": True}, box
            eng.reserve(n_res, int(cs.variants.alleles.size))          # (again: nothing left to do)
            _assert_same(eng.filter_variants(cs.variants), exp, f"after reserve({n_res})")
            _assert_same(eng.filter_variants(cs.variants.slice(0, 5000)), O.filter_variants(
                cs.variants.slice(0, 5000), cs.ref, cs.runs, cs.tracks, cs.blacklist, forests), "small callset after reserve")
            with pytest.raises(RuntimeError, match="reserve"):
                eng.reserve(-1, 0)
            # a reservation that fits what is resident leaves it alone; one that must grow a resident column empties the
            # context instead of leaving the old row count beside fresh, uninitialised columns
            res = eng.filter_variants(cs.variants)
            assert eng.resident_count() == (cs.variants.n, True)
            eng.reserve(cs.variants.n, int(cs.variants.alleles.size))
            assert eng.resident_count() == (cs.variants.n, True)
            _assert_same(eng.download_results(), res, "resident results after a reservation that fits")
            eng.reserve(4 * cs.variants.n, 4 * int(cs.variants.alleles.size))
            assert eng.resident_count() == (0, False) and eng.download_results().filter.size == 0
            eng.set_sec_db(np.array([5], np.uint64), np.ones((1, 3), np.int32))
            ratio, hit = eng.sec_apply()                              # (no rows: nothing to apply the database to)
            assert ratio.size == 0 and hit.size == 0
            _assert_same(eng.filter_variants(cs.variants), exp, "after an emptying reservation")


def test_chunk_pipeline_boundary(engine, frozen_models):
    """ugvc_filter_variants on a callset large enough for the chunk pipeline (csrc/pipeline.hip: >= 262144 rows; passes read
    the staging blocks, results are written packed and placed into the resident columns behind the pass): (i) equals the
    oracle on every row of a 300 k callset whose last chunk is ragged; (ii) leaves the context as upload + resident pass
    would - a resident pass and a download give the same columns, and the feature matrix of the resident callset equals the
    oracle's; (iii) a row that breaks the sort order in a LATER chunk is reported by its row number; (iv) the next valid call is unaffected."""
    import copy
    from variantcalling_amd import schema as S, synth
    O = _oracle()
    cs = synth.make_callset(300_011, genome_len=150_000_000, n_contigs=5, seed=99)
    forests = frozen_models[RF]
    _configure(engine, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
    vt = cs.variants
    assert vt.n >= 262144
    got = engine.filter_variants(vt)
    exp = O.filter_variants(vt, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
    _assert_same(got, exp, "pipelined")
    _assert_same(engine.download_results(), exp, "resident result columns after the pipelined call")
    engine.filter_resident()
    _assert_same(engine.download_results(), exp, "resident pass over the placed columns")
    X, g = engine.feature_matrix()
    f = O.featurize(vt, cs.ref, cs.runs, cs.tracks)
    assert np.array_equal(X, f["X"]) and np.array_equal(g, f["group"])
    bad = copy.copy(vt)
    bad.pos = vt.pos.copy()
    row = 200_003
    same = np.flatnonzero(vt.contig == vt.contig[row])
    assert same[0] < row - 2
    bad.pos[row] = bad.pos[row - 2] - 1 if bad.pos[row - 2] > 1 else 0
    with pytest.raises(RuntimeError, match=f"row {row}$"):
        engine.filter_variants(bad)
    _assert_same(engine.filter_variants(vt), exp, "after the failed call")
    # the same callset with the allele pool laid out the OTHER way round (every row's ALT in front of its REF): the boundary
    # call ships ref_off / alt_off only when they do not follow from the lengths (csrc/pipeline.hip: canonical pools have
    # them rebuilt on the device) - both ways must give the same rows, and the resident offsets must be usable afterwards
    rl, al = vt.ref_len.astype(np.int64), vt.alt_len.astype(np.int64)
    start = np.cumsum(rl + al) - (rl + al)
    swapped = copy.copy(vt)
    swapped.alt_off = start.astype(np.uint32)
    swapped.ref_off = (start + al).astype(np.uint32)
    pool = np.zeros(int((rl + al).sum()), np.uint8)
    for k in range(int(max(rl.max(), al.max()))):
        m = np.flatnonzero(al > k)
        pool[swapped.alt_off[m].astype(np.int64) + k] = vt.alleles[vt.alt_off[m].astype(np.int64) + k]
        m = np.flatnonzero(rl > k)
        pool[swapped.ref_off[m].astype(np.int64) + k] = vt.alleles[vt.ref_off[m].astype(np.int64) + k]
    swapped.alleles = pool
    assert np.array_equal(vt.ref_off.astype(np.int64) + rl, vt.alt_off.astype(np.int64))        # the synthetic pool is canonical
    _assert_same(engine.filter_variants(swapped), exp, "non-canonical allele pool")
    X2, g2 = engine.feature_matrix()
    assert np.array_equal(X2, f["X"]) and np.array_equal(g2, f["group"])
    _assert_same(engine.filter_variants(vt), exp, "canonical again")
    X3, _ = engine.feature_matrix()
    assert np.array_equal(X3, f["X"])
    # the caller's result arrays, reused from call to call (no fresh 30 MB of numpy per 5 M variants)
    keep = S.FilterResult(np.full(vt.n, 7, np.float32), np.full(vt.n, 7, np.uint8), np.full(vt.n, 7, np.uint8))
    assert engine.filter_variants(vt, out=keep) is keep
    _assert_same(keep, exp, "out=")
    with pytest.raises(ValueError, match="out:"):
        engine.filter_variants(vt, out=S.FilterResult(np.zeros(vt.n, np.float64), keep.filter, keep.flags))
    with pytest.raises(ValueError, match="out:"):
        engine.filter_variants(vt, out=S.FilterResult(keep.tree_score[:-1], keep.filter[:-1], keep.flags[:-1]))


def test_golden_fixture_outputs(engine, frozen_models):
    """Committed golden outputs (tests/golden/filter_golden_v1.npz, made by make_filter_golden.py
    from the oracle) - guards oracle drift as well as the kernel."""
    z = np.load(os.path.join(GOLDEN, "filter_golden_v1.npz"))
    ref = real_chr1_reference()
    vt = E.edge_table(ref)
    runs, tracks = E.simple_tracks(ref)
    bl = np.unique(np.concatenate([vt.keys()[::7], vt.keys()[::11] + np.uint64(1)]))
    _configure(engine, ref, runs, tracks, bl, frozen_models[RF], hpol_len=8, hpol_dist=12)
    res = engine.filter_variants(vt)
    assert np.array_equal(res.filter, z["filter"]) and np.array_equal(res.flags, z["flags"])
    assert np.array_equal(res.tree_score, z["tree_score"])
    X, _ = engine.feature_matrix()
    assert np.array_equal(X, z["X"])


def test_pileup_tally(engine):
    from variantcalling_amd import synth
    O = _oracle()
    off, obs = synth.make_pileup(50_000, seed=5)
    # ragged: empty loci, a 5000-deep locus, a single-read locus
    d = np.diff(off).copy()
    d[[0, 17, 4999]] = 0
    d[123] = 5000
    d[-1] = 1
    off2 = np.concatenate([[0], np.cumsum(d)])
    rng = np.random.default_rng(1)
    obs2 = rng.choice(obs, size=int(off2[-1]))
    for o, b in ((off, obs), (off2, obs2)):
        got = engine.pileup_tally(o, b)
        exp = O.pileup_tally(o, b)
        for k in ("ref_fwd", "ref_rev", "alt_fwd", "alt_rev", "other", "dp", "bq_ref", "bq_alt", "ad_ref", "ad_alt"):
            assert np.array_equal(got[k], exp[k]), k
        assert np.array_equal(got["vaf"], exp["vaf"])
        assert np.max(np.abs(got["sor"] - exp["sor"])) <= 1e-5
    empty = engine.pileup_tally(np.zeros(1, np.int64), np.zeros(0, np.uint16))
    assert empty["dp"].size == 0


def test_sec_statistic_kats(engine):
    """Reference KATs (test/unit/utils/test_stats_utils.py:48-110) through the GPU kernel."""
    from oracle import stats as st
    cases = [([4, 4, 4], [4, 4, 4]), ([4, 4, 4], [40, 40, 40]), ([40, 40, 40], [40, 40, 40]), ([4, 4, 40], [4, 4, 4]),
             ([4, 4, 40], [40, 40, 40]), ([10, 10, 10], [1, 10, 40]), ([40, 10, 1], [1, 10, 40]),
             ([1, 10, 40], [1, 10, 40]), ([4, 4, 4], [4, 4, 0]), ([4, 4, 40], [0, 0, 0]), ([0, 0, 0], [3, 2, 1])]
    a = np.array([c[0] for c in cases], np.int32)
    e = np.array([c[1] for c in cases], np.int32)
    lik, ratio = engine.sec_likelihood_ratio(a, e)
    kat_lik = [0.0652, 0.0652, 0.0068, 3.3e-13, 3.3e-13, 2.1e-10, 2.7e-53, 0.039, 0.0043, 3.3e-13]
    kat_ratio = [1, 1, 1, 3.3e-13, 3.3e-13, 7.8e-9, 6.9e-52, 1, 0.0661, 9.1e-12]
    places = [3, 3, 3, 10, 10, 10, 40, 3, 3, 3]
    for i in range(10):
        assert round(abs(lik[i] - kat_lik[i]), places[i]) == 0, (i, lik[i])
    for i, pl in enumerate([3, 3, 3, 10, 10, 10, 40, 3, 3, 10]):
        assert round(abs(ratio[i] - kat_ratio[i]), pl) == 0, (i, ratio[i])
    ol, orr = st.sec_batch(a, e)
    assert np.allclose(lik, ol, rtol=1e-10, atol=0) and np.allclose(ratio, orr, rtol=1e-10, atol=0)
    rng = np.random.default_rng(3)
    A = rng.integers(0, 60, size=(20000, 5)).astype(np.int32)
    Ex = rng.integers(0, 400, size=(20000, 5)).astype(np.int32)
    lik, ratio = engine.sec_likelihood_ratio(A, Ex)
    ol, orr = st.sec_batch(A, Ex)
    assert np.allclose(lik, ol, rtol=1e-9, atol=0) and np.allclose(ratio, orr, rtol=1e-9, atol=0)


def test_bridging_snvs(engine):
    from oracle import bridging as B
    ref = real_chr1_reference()
    vt = E.edge_table(ref, seed=9, n_random=6000)
    rng = np.random.default_rng(4)
    n = vt.n
    is_pass = rng.random(n) < 0.3
    ad_alt = rng.integers(0, 40, n).astype(np.int32)
    bg_ad = rng.integers(0, 6, n).astype(np.int32)
    bg_dp = rng.integers(0, 40, n).astype(np.int32)
    engine.set_reference(ref)
    for h, edge in ((2, 0), (3, 0), (5, 1), (4, 0)):
        got = engine.bridging_snvs(vt, is_pass, ad_alt, bg_ad, bg_dp, min_query_hmer_size=h, min_distance_from_edge=edge)
        exp = B.calibrate(vt, ref, is_pass, ad_alt, bg_ad, bg_dp, min_query_hmer_size=h, min_distance_from_edge=edge)
        assert np.array_equal(got[0], exp[0]), h
        assert np.array_equal(got[1], exp[1]), h
    assert exp[0].sum() > 0


def test_full_size_properties(engine, frozen_models):
    """BASELINE C3 size (5 M variants, 3.1 Gb genome): properties that need no oracle run.
    (i) idempotence/determinism; (ii) shard invariance: scoring [0,n) equals scoring the two
    halves separately; (iii) a 20 k random slice equals the oracle."""
    from variantcalling_amd import synth
    O = _oracle()
    cs = synth.make_callset(5_000_000)
    _configure(engine, cs.ref, cs.runs, cs.tracks, cs.blacklist, frozen_models[RF])
    full = engine.filter_variants(cs.variants)
    again = engine.filter_variants(cs.variants)
    _assert_same(full, again, "determinism")
    n = cs.variants.n
    cut = n // 2 + 13
    a = engine.filter_variants(cs.variants.slice(0, cut))
    b = engine.filter_variants(cs.variants.slice(cut, n))
    assert np.array_equal(np.concatenate([a.filter, b.filter]), full.filter)
    assert np.array_equal(np.concatenate([a.flags, b.flags]), full.flags)
    assert np.array_equal(np.concatenate([a.tree_score, b.tree_score]), full.tree_score)
    lo = 3_111_111
    sub = cs.variants.slice(lo, lo + 20_000)
    exp = O.filter_variants(sub, cs.ref, cs.runs, cs.tracks, cs.blacklist, frozen_models[RF])
    assert np.array_equal(full.filter[lo:lo + 20_000], exp.filter)
    assert np.array_equal(full.flags[lo:lo + 20_000], exp.flags)
    assert np.array_equal(full.tree_score[lo:lo + 20_000], exp.tree_score)
    assert 0.05 < (full.filter == 0).mean() < 0.95


def test_feature_matrix_between_scoring_passes(engine, small_callset, frozen_models):
    """Regression (round 3): the feature matrix now comes from the fused kernel's featurize waves; such a launch writes no
    indel records and runs no forest kernel, so it must not touch the two record-counter sets - it used to mark the set the
    previous scoring pass had counted into as zeroed, and the next pass on ANOTHER callset then walked the stale records
    (caught by the determinism test: wrong TREE_SCORE in the first rows)."""
    from variantcalling_amd import synth
    O = _oracle()
    a = small_callset
    b = synth.make_callset(45_000, genome_len=25_000_000, n_contigs=3, seed=4321)
    forests = frozen_models[RF]
    for first, second in ((a, b), (b, a)):
        _configure(engine, first.ref, first.runs, first.tracks, first.blacklist, forests)
        engine.filter_variants(first.variants)
        engine.feature_matrix(first.variants)
        engine.feature_matrix(first.variants)
        _configure(engine, second.ref, second.runs, second.tracks, second.blacklist, forests)
        got = engine.filter_variants(second.variants)
        _assert_same(got, O.filter_variants(second.variants, second.ref, second.runs, second.tracks, second.blacklist, forests), "after feature matrices")
        again = engine.filter_variants(second.variants)
        _assert_same(again, got, "determinism")


@pytest.mark.parametrize("config", ["C3", "C2"])
def test_full_size_every_row_equals_the_oracle(engine, frozen_models, config):
    """BASELINE.json configs at FULL size, every row: C3 (5 M SNV + indel calls, the configuration the metric is quoted on)
    and C2 (1 M SNV-only) through the C ABI, then FILTER / flags / TREE_SCORE of ALL rows against `oracle.filter_variants`,
    in chunks the vectorised oracle digests in a few seconds (~1.8e5 variants/s on one core: ~30 s for C3, ~6 s for C2).
    Round 2 compared a 20 k slice."""
    from variantcalling_amd import synth
    O = _oracle()
    cs = synth.make_callset(5_000_000) if config == "C3" else synth.make_callset(1_000_000, snv_only=True)
    forests = frozen_models[RF]
    _configure(engine, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
    got = engine.filter_variants(cs.variants)
    # ... and the RESIDENT path - upload_variants + filter_resident + download_results: one launch pair over the whole callset, the
    # path bench.py times (the host-buffer call above scores the callset chunk by chunk) - every row as well
    engine.upload_variants(cs.variants)
    engine.filter_resident()
    got_res = engine.download_results()
    # a second resident pass over the same rows: what the bench's K timed steps repeat
    engine.filter_resident()
    got_res2 = engine.download_results()
    n = cs.variants.n
    assert n > (4_900_000 if config == "C3" else 990_000)
    bad = 0
    for a in range(0, n, 250_000):
        b = min(a + 250_000, n)
        exp = O.filter_variants(cs.variants.slice(a, b), cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
        for path, res in (("host-buffer", got), ("resident", got_res), ("resident, second pass", got_res2)):
            for what in ("filter", "flags", "tree_score"):
                g, e = getattr(res, what)[a:b], getattr(exp, what)
                if not np.array_equal(g, e):
                    rows = a + np.flatnonzero(g != e)
                    bad += rows.size
                    print(f"{config} {path} rows {a}..{b}: {rows.size} {what} differ, first {rows[:5]}")
    assert bad == 0


def test_c5_leaf_matrix_gemm(engine, small_callset, frozen_models):
    """Config C5: the XGBoost-shaped ensemble (T = 100, depth 6) evaluated on the resident feature matrix
    as a leaf-matrix GEMM on MFMA and as a row traversal: both bit-identical to the oracle's f32 margins."""
    from variantcalling_amd import schema as S
    O = _oracle()
    cs = small_callset
    forests = frozen_models[XGB]
    assert all(f.n_trees == 100 and f.max_depth <= 6 for f in forests)
    _configure(engine, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
    X, group = engine.feature_matrix(cs.variants)
    for g in range(S.N_GROUPS):
        rows = np.flatnonzero(group == g).astype(np.int32)
        exp_margin, exp_score = O.forest_predict(forests[g], X[rows])
        for mfma in (1, 2, 0):                                   # round-4 kernel (lane = row predicates), round-1 kernel, traversal
            got, ms = engine.forest_gemm(g, rows, use_mfma=mfma)
            assert np.array_equal(got, exp_margin), (g, mfma, np.flatnonzero(got != exp_margin)[:5])
            assert ms > 0
    # round 5: the three groups in ONE launch, margins by row of the resident matrix - every row against the oracle; then with an
    # empty group, a group of one row, and rows named in descending order
    rows_g = [np.flatnonzero(group == g).astype(np.int32) for g in range(S.N_GROUPS)]
    exp_all = np.zeros(X.shape[0], np.float32)
    for g in range(S.N_GROUPS):
        exp_all[rows_g[g]] = O.forest_predict(forests[g], X[rows_g[g]])[0]
    got3, ms3 = engine.forest_gemm3(rows_g)
    assert np.array_equal(got3, exp_all) and ms3 > 0
    odd = [rows_g[0][:1], None, rows_g[2][::-1][:130]]
    got3, _ = engine.forest_gemm3(odd)
    named = np.zeros(X.shape[0], bool)
    named[odd[0]] = True
    named[odd[2]] = True
    assert np.array_equal(got3[named], exp_all[named]) and not got3[~named].any()
    with pytest.raises(RuntimeError, match="out of range"):
        engine.forest_gemm3([np.array([X.shape[0]], np.int32), None, None])
    # all rows with one group's model, ragged tail (n not a multiple of 16), tiny inputs
    got, _ = engine.forest_gemm(0, None, use_mfma=True)
    assert np.array_equal(got, O.forest_predict(forests[0], X)[0])
    for k in (1, 15, 17, 63, 64, 65, 129):                       # (tiles of 64 rows in the round-4 kernel, of 16 in the round-1 one)
        for mfma in (1, 2):
            got, _ = engine.forest_gemm(2, np.arange(k, dtype=np.int32), use_mfma=mfma)
            assert np.array_equal(got, O.forest_predict(forests[2], X[:k])[0]), (k, mfma)
    # RF / deep trees are refused, not approximated
    _configure(engine, cs.ref, cs.runs, cs.tracks, cs.blacklist, frozen_models[RF])
    engine.feature_matrix(cs.variants)
    with pytest.raises(RuntimeError, match="additive|depth"):
        engine.forest_gemm(0, None)


def test_gemm3_falls_back_to_the_traversal_when_its_self_check_fails(tmp_path):
    """forest_gemm3_kernel reads its predicates' operands by register index and relies on where the compiler put the row's feature
    registers; its first use per process is checked against the scalar traversal.  Round 5 FAILED the call on a mismatch - a hipcc
    update could have turned config C5's default path into an error.  Round 6: the traversal serves the call (and every later one),
    bit-identical margins, one note on stderr.  The mismatch is forced (UGVC_GEMM3_FORCE_MISMATCH, read by the self-check alone)
    in a process of its own: the state is per process."""
    import subprocess
    import sys
    code = r"""
import os, sys
import numpy as np
sys.path.insert(0, %r)
from oracle import oracle as O
from variantcalling_amd import model_io, schema as S, synth
from variantcalling_amd.engine import Engine, configure
forests = model_io.load_models(os.path.join(%r, "tests", "golden", "synth_rf_v1.npz"))["xgb_model_ignore_gt_incl_hpol_runs"]
cs = synth.make_callset(20_000, genome_len=10_000_000, n_contigs=3, seed=11)
with Engine(0) as eng:
    configure(eng, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
    X, group = eng.feature_matrix(cs.variants)
    rows_g = [np.flatnonzero(group == g).astype(np.int32) for g in range(S.N_GROUPS)]
    exp = np.zeros(X.shape[0], np.float32)
    for g in range(S.N_GROUPS):
        exp[rows_g[g]] = O.forest_predict(forests[g], X[rows_g[g]])[0]
    for _ in range(2):                                   # the first call runs the self-check, the second finds the state
        got, ms = eng.forest_gemm3(rows_g, iters=2)
        assert np.array_equal(got, exp) and ms > 0
print("FALLBACK-OK")
""" % (ROOT, ROOT)
    env = dict(os.environ, UGVC_GEMM3_FORCE_MISMATCH="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "FALLBACK-OK" in r.stdout, r.stderr[-2000:]
    assert r.stderr.count("forest_gemm3_kernel disagrees with the scalar traversal") == 1     # said once


def _stump_forest(specs, n_features=20):
    """specs: [(feature, threshold, (p0, p1) left, (p0, p1) right)] -> one MODEL_RF FlatForest of stumps."""
    from variantcalling_amd import schema as S
    feat, thr, left, right, roots, leaves = [], [], [], [], [], []
    for k, (f, t, lv, rv) in enumerate(specs):
        roots.append(3 * k)
        feat += [f, -1, -1]; thr += [t, 0.0, 0.0]
        left += [3 * k + 1, 2 * k, 2 * k + 1]; right += [3 * k + 2, 0, 0]
        leaves += [lv, rv]
    return S.FlatForest(S.MODEL_RF, np.array(feat, np.int32), np.array(thr, np.float32), np.array(left, np.int32),
                        np.array(right, np.int32), np.array(roots, np.int32), np.array(leaves, np.float64),
                        n_features=n_features, max_depth=1)


@pytest.mark.parametrize("path", [0, 65536, 65536 | 1024, 256], ids=["v5", "v3-single-sum", "v3-pair-sums", "v1"])
def test_rf_vote_ties_and_unnormalised_payloads(engine, small_callset, path):
    """The single-sum forest kernel decides PASS on the class-1 sum alone and must fall back to both
    sums where scikit-learn's argmax is decided by them: exact ties (pure leaves, even T), near ties
    (0.3 + 0.7 payloads) - and must not be selected at all for payloads that do not sum to 1."""
    cs = small_callset
    O = _oracle()
    pure = [(2, 24.5, (1.0, 0.0), (0.0, 1.0)), (2, 29.5, (1.0, 0.0), (0.0, 1.0)),
            (2, 34.5, (0.0, 1.0), (1.0, 0.0)), (6, 50.5, (0.0, 1.0), (1.0, 0.0))]
    near = [(2, 27.5, (0.3, 0.7), (0.7, 0.3)), (0, 40.0, (0.7, 0.3), (0.3, 0.7)),
            (4, 9.5, (0.1, 0.9), (0.9, 0.1)), (3, 12.5, (0.9, 0.1), (0.1, 0.9)),
            (1, 1.0, (1.0 / 3.0, 2.0 / 3.0), (2.0 / 3.0, 1.0 / 3.0)), (5, 0.4, (2.0 / 3.0, 1.0 / 3.0), (1.0 / 3.0, 2.0 / 3.0))]
    # 0.9 + 0.15 + 0.15 + 0.8 and 0.1 + 0.85 + 0.85 + 0.2 differ by an ulp in f64: PASS is decided by it
    ulp = [(2, 1000.5, (0.1, 0.9), (0.9, 0.1)), (2, 1000.5, (0.85, 0.15), (0.15, 0.85)),
           (2, 30.5, (0.85, 0.15), (0.5, 0.5)), (2, 1000.5, (0.2, 0.8), (0.8, 0.2))]
    counts = [(2, 27.5, (3.0, 5.0), (6.0, 2.0)), (0, 40.0, (4.0, 4.0), (1.0, 7.0))]
    for name, specs in (("pure", pure), ("near", near), ("ulp", ulp), ("counts", counts)):
        forests = [_stump_forest(specs)] * 3
        _configure(engine, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
        engine.set_kernel_variant(path)
        res = engine.filter_variants(cs.variants)
        exp = O.filter_variants(cs.variants, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
        _assert_same(res, exp, name)
        if name != "counts":
            T = len(specs)
            tied = np.abs(exp.tree_score.astype(np.float64) - 0.5) < 1e-6
            assert tied.sum() > 100, "the case must exercise the tie band"
            if name == "ulp":
                assert (exp.filter[tied] == 0).sum() > 100, "ties broken towards PASS by one ulp"
        engine.set_kernel_variant(0)


@PATHS
def test_code_table_edges(engine, small_callset, path):
    """Integer features are quantised through code tables that reach one past the model's top threshold (larger
    values take the last entry) and are cut at 8192 entries (larger values search the thresholds); negative
    values always search.  Thresholds around both limits, values on both sides."""
    import copy
    cs = small_callset
    O = _oracle()
    vt = copy.deepcopy(cs.variants)
    rng = np.random.default_rng(3)
    n = vt.n
    vt.dp = vt.dp.copy(); vt.ad_alt = vt.ad_alt.copy(); vt.ad_ref = vt.ad_ref.copy()
    pick = rng.permutation(n)
    vt.dp[pick[:400]] = rng.choice([0, 1, 59, 60, 61, 62, 8190, 8191, 8192, 8193, 8999, 9000, 9001, 20000, 2**31 - 1], 400)
    vt.dp[pick[400:600]] = rng.choice([-1, -3, -(2**31)], 200)
    vt.ad_alt[pick[600:800]] = rng.choice([-2, 0, 41, 42, 43, 100000], 200)
    specs = [(2, 60.5, (0.9, 0.1), (0.2, 0.8)), (2, 9000.5, (0.3, 0.7), (0.6, 0.4)), (2, 8190.5, (0.7, 0.3), (0.4, 0.6)),
             (4, 41.5, (0.8, 0.2), (0.1, 0.9)), (4, -0.5, (0.25, 0.75), (0.5, 0.5)), (3, 12.5, (0.35, 0.65), (0.65, 0.35))]
    forests = [_stump_forest(specs)] * 3
    _configure(engine, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
    engine.set_kernel_variant(path)
    res = engine.filter_variants(vt)
    exp = O.filter_variants(vt, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
    _assert_same(res, exp, "code tables")
    assert len(np.unique(exp.tree_score)) > 4


def test_rccl_gather_path_single_rank(small_callset, frozen_models):
    """The N > 1 data path (RCCL all-gather of the three result columns on its own stream, overlapped
    with the next scoring pass) exercised with a one-rank communicator: librccl is dlopen'ed, the
    communicator initialised, the grouped in-place all-gathers issued and fenced on real hardware."""
    from variantcalling_amd.engine import Engine, configure
    O = _oracle()
    cs = small_callset
    with Engine(0) as e2:
        configure(e2, cs.ref, cs.runs, cs.tracks, cs.blacklist, frozen_models[RF])
        e2.upload_variants(cs.variants)
        e2.comm_init(e2.comm_unique_id(), 0, 1)
        assert e2.comm_info() == dict(nranks=1, rank=0, device=0)        # what RCCL itself reports (ncclCommCount / UserRank / CuDevice)
        cap = cs.variants.n + 37                      # padded shard, as ceil(N / world) is in general
        tot, ker = e2.timed_steps(4, cap, True)       # 4 x {scoring pass, overlapped gather}
        assert tot > 0 and ker > 0
        e2.filter_resident()
        e2.allgather_resident(cap)
        got = e2.gathered_download(cap, 1, [cs.variants.n])
        exp = O.filter_variants(cs.variants, cs.ref, cs.runs, cs.tracks, cs.blacklist, frozen_models[RF])
        _assert_same(got, exp, "gathered")
        _assert_same(e2.download_results(), exp, "resident")
