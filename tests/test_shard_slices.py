"""Per-rank context slices ("reference slice [min_pos - 64, max_pos + 64] per touched contig + the
overlapping part of each table"): scoring a shard against its sliced, re-based tables gives the rows of the whole-callset
result.  CPU: on the oracle; GPU: the same through the C ABI (tests/test_gpu_parity.py::test_sliced_context_equals_full)."""
import numpy as np
import pytest

import edge_cases as E
from conftest import real_chr1_reference
from oracle import oracle as O
from variantcalling_amd import schema as S
from variantcalling_amd import shard, synth

RF = "rf_model_ignore_gt_incl_hpol_runs"


def _same(got, full, lo, hi, what):
    assert np.array_equal(got.flags, full.flags[lo:hi]), what
    assert np.array_equal(got.filter, full.filter[lo:hi]), what
    assert np.array_equal(got.tree_score, full.tree_score[lo:hi]), what


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sliced_context_scores_like_the_full_one(frozen_models, world):
    cs = synth.make_callset(12_000, genome_len=6_000_000, n_contigs=4, seed=3)
    forests = frozen_models[RF]
    full = O.filter_variants(cs.variants, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
    b = shard.shard_bounds(cs.variants.n, world)
    kept = 0
    for r in range(world):
        mine = shard.shard_of(cs.variants, r, world)
        ref_s, runs_s, tracks_s, bl_s, mine_s = shard.slice_context(cs.ref, cs.runs, cs.tracks, cs.blacklist, mine)
        assert ref_s.n_contigs == cs.ref.n_contigs and mine_s.n == mine.n and int(mine_s.pos.min()) >= 1
        mine_s.validate()
        _same(O.filter_variants(mine_s, ref_s, runs_s, tracks_s, bl_s, forests), full, int(b[r]), int(b[r + 1]), f"world {world} rank {r}")
        kept += ref_s.codes.size
    assert kept < 1.2 * cs.ref.codes.size                       # the ranks' slices together are about one genome


def test_sliced_context_on_real_hg38_edges(frozen_models):
    """Contig ends, N runs, 50 kb homopolymers, long deletions: a cut never splits a run a variant can see."""
    ref = real_chr1_reference()
    vt = E.edge_table(ref)
    runs, tracks = E.simple_tracks(ref)
    bl = np.unique(np.concatenate([vt.keys()[::7], vt.keys()[::11] + np.uint64(1)]))
    forests = frozen_models[RF]
    full = O.filter_variants(vt, ref, runs, tracks, bl, forests, hpol_len=8, hpol_dist=12)
    for world in (2, 5):
        b = shard.shard_bounds(vt.n, world)
        for r in range(world):
            mine = shard.shard_of(vt, r, world)
            if mine.n == 0:
                continue
            ref_s, runs_s, tracks_s, bl_s, mine_s = shard.slice_context(ref, runs, tracks, bl, mine)
            _same(O.filter_variants(mine_s, ref_s, runs_s, tracks_s, bl_s, forests, hpol_len=8, hpol_dist=12), full,
                  int(b[r]), int(b[r + 1]), f"edges world {world} rank {r}")


def test_cuts_snap_to_a_contig_change_within_one_percent(frozen_models):
    """a cut moves to a contig's first row when that keeps every shard within 1 % of the equal share; cuts
    without such a neighbour stay equal-count; the snapped shards still reassemble to the whole callset."""
    # the plain cut is untouched without the column, with one contig, or when no change is near
    assert shard.shard_bounds(1000, 4).tolist() == [0, 250, 500, 750, 1000]
    assert shard.shard_bounds(1000, 4, np.zeros(1000, np.uint16)).tolist() == [0, 250, 500, 750, 1000]
    far = np.repeat(np.arange(2, dtype=np.uint16), [400, 600])            # change at row 400: 100 rows from the nearest cut
    assert shard.shard_bounds(1000, 4, far).tolist() == [0, 250, 500, 750, 1000]
    # changes 2 rows before the first cut and 1 row behind the last one: both snap; the middle cut has no neighbour
    near = np.repeat(np.arange(3, dtype=np.uint16), [248, 503, 249])       # changes at rows 248 and 751
    assert shard.shard_bounds(1000, 4, near).tolist() == [0, 248, 500, 751, 1000]
    assert shard.shard_cap(1000, 4, shard.shard_bounds(1000, 4, near)) == 256 and shard.shard_cap(1000, 4) == 256
    # one row beyond the 1 % slack (2.5 rows of a 250-row share): stays
    out = np.repeat(np.arange(2, dtype=np.uint16), [246, 754])
    assert shard.shard_bounds(1000, 4, out).tolist() == [0, 250, 500, 750, 1000]
    # properties on random columns: monotone, ends fixed, every shard within (1 + tol) of the share, a moved cut IS a change
    rng = np.random.default_rng(2)
    for _ in range(200):
        n, world = int(rng.integers(1, 5000)), int(rng.integers(1, 9))
        contig = np.sort(rng.integers(0, int(rng.integers(1, 30)), n)).astype(np.uint16)
        plain, b = shard.shard_bounds(n, world), shard.shard_bounds(n, world, contig)
        assert b[0] == 0 and b[-1] == n and np.all(np.diff(b) >= 0)
        assert np.max(np.diff(b)) <= n / world * 1.01 + 1
        for r in np.flatnonzero(b != plain):
            assert contig[b[r]] != contig[b[r] - 1]
        assert shard.shard_cap(n, world, b) >= np.max(np.diff(b)) and shard.shard_cap(n, world, b) % 256 == 0
    # end to end: a callset whose second contig starts 3 rows behind the middle; the two snapped shards score like the whole
    cs = synth.make_callset(6_000, genome_len=3_000_000, n_contigs=2, seed=9)
    vt = cs.variants
    change = int(np.flatnonzero(vt.contig[1:] != vt.contig[:-1])[0]) + 1
    k = min(change - 3, vt.n - change + 3)                                 # 2 k rows with the change at row k + 3
    vt = vt.slice(change - k - 3, change + k - 3)
    assert vt.n == 2 * k and k > 1000
    b = shard.shard_bounds(vt.n, 2, vt.contig)
    assert b[1] == vt.n // 2 + 3 and vt.contig[b[1]] != vt.contig[b[1] - 1]
    forests = frozen_models[RF]
    full = O.filter_variants(vt, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
    for r in range(2):
        mine = shard.shard_of(vt, r, 2, snap=True)
        assert mine.n == int(b[r + 1] - b[r]) and np.unique(mine.contig).size == 1      # neither rank carries the boundary
        ref_s, runs_s, tracks_s, bl_s, mine_s = shard.slice_context(cs.ref, cs.runs, cs.tracks, cs.blacklist, mine)
        _same(O.filter_variants(mine_s, ref_s, runs_s, tracks_s, bl_s, forests), full, int(b[r]), int(b[r + 1]), f"snapped rank {r}")


def test_empty_shard_and_missing_tables():
    cs = synth.make_callset(500, genome_len=400_000, n_contigs=2, seed=5)
    empty = cs.variants.slice(0, 0)
    ref_s, runs_s, tracks_s, bl_s, mine_s = shard.slice_context(cs.ref, None, [], None, empty)
    assert ref_s.codes.size == 0 and runs_s is None and tracks_s == [] and bl_s is None and mine_s.n == 0


def test_cut_inside_a_long_run_is_vectorised_and_pad_follows_hpol_dist():
    """the run a cut falls into is finished by block compares (a multi-megabase N run used to cost a Python
    iteration per base), and the table pad follows --hpol_filter_length_dist's distance."""
    import time
    n = 6_000_000
    codes = np.full(n, 0, np.uint8)                       # one long N run ...
    codes[:1000] = np.tile(np.array([1, 2, 3, 4], np.uint8), 250)
    codes[-1000:] = np.tile(np.array([4, 3, 2, 1], np.uint8), 250)
    assert shard._run_end(codes, 2000, n) == n - 1000
    ref = S.Reference(codes, np.array([0, n], np.int64), ["chrN"])
    vt = S.VariantTable(contig=np.zeros(2, np.uint16), pos=np.array([500, 900], np.int32), ref_len=np.ones(2, np.uint16),
                        alt_len=np.ones(2, np.uint16), ref_off=np.array([0, 2], np.uint32), alt_off=np.array([1, 3], np.uint32),
                        alleles=np.array([1, 2, 3, 4], np.uint8), qual=np.ones(2, np.float32), sor=np.ones(2, np.float32),
                        dp=np.ones(2, np.int32), ad_ref=np.ones(2, np.int32), ad_alt=np.ones(2, np.int32), gt=np.ones(2, np.uint8), gq=np.ones(2, np.uint8))
    t0 = time.perf_counter()
    ref_s, *_ = shard.slice_context(ref, None, [], None, vt, margin=200)          # the right cut (pos 901 + 200) lands in the N run
    assert time.perf_counter() - t0 < 2.0
    assert ref_s.codes.size == n - 1000 + 200 - (500 - 1 - 200)
    # a run that begins 300 bases past the shard's last call still marks it when the distance is 400
    runs = S.IntervalTrack(np.array([1200], np.int32), np.array([1230], np.int32), np.array([0, 1], np.int32), "runs")
    _, runs_near, *_ = shard.slice_context(ref, runs, [], None, vt, margin=64, pad=128)
    _, runs_far, *_ = shard.slice_context(ref, runs, [], None, vt, margin=64, pad=128, hpol_dist=400)
    assert runs_far.starts.size == 1
    assert runs_near.starts.size in (0, 1)                # (one row of halo may keep it; the pad is what guarantees it)
