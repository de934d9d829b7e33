"""Randomised differential test: every production kernel path against the oracle on callsets whose shape,
side tables, parameters and model are drawn at random (seeded).  Complements the targeted cases of
tests/test_gpu_parity.py: the bar is the same - bit-exact FILTER / flags / RF tree_score, 1e-6 on the XGBoost sigmoid."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RF = "rf_model_ignore_gt_incl_hpol_runs"
XGB = "xgb_model_ignore_gt_incl_hpol_runs"


# UGVC_FUZZ_OFFSET=k: the same tests over OTHER random draws (every seed below is shifted by 100 000 k) - tools/gpu_suite.sh
# `fuzzmore` runs a few offsets under the guard-page / poison allocation modes; the default suite is offset 0
_OFF = 100_000 * int(os.environ.get("UGVC_FUZZ_OFFSET", "0"))


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
    cs, vt, runs, tracks, bl, flow, hp_len, hp_dist, model = _draw(_OFF + 1000 + seed)
    forests = frozen_models[model]
    if any(f is not None and f.n_features > 17 + len(tracks) for f in forests):
        tracks = list(cs.tracks)                              # the frozen models test all three track features
    configure(engine, cs.ref, runs, tracks, bl, forests, flow, hp_len, hp_dist, True)
    exp = O.filter_variants(vt, cs.ref, runs, tracks, bl, forests, hpol_len=hp_len, hpol_dist=hp_dist, flow_order=flow)
    for path in (0, 65536, 256):
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
    rng = np.random.default_rng(_OFF + 7000 + seed)
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


def _random_forest(rng, kind, n_features, n_trees, max_depth, normalised=True):        # normalised: True | "coarse" | False
    """A random ensemble in the pointer layout: unbalanced trees, repeated / extreme thresholds, any feature."""
    from variantcalling_amd import schema as S
    feat, thr, left, right, roots, leaves = [], [], [], [], [], []

    def threshold(f):
        if f == 0: return float(rng.choice([rng.exponential(60.0), 0.0, 2999.5, 1e6, -1.0]))
        if f == 1: return float(rng.choice([rng.lognormal(0, 0.7), 0.0, 7.5]))
        if f in (5, 13): return float(rng.choice([rng.random(), 0.0, 0.5, 1.0, 1.5, -0.25]))
        if f in (11, 12): return float(rng.integers(0, 3125)) + float(rng.choice([0.0, 0.5]))
        if f >= 15: return float(rng.choice([0.5, 0.5, 0.5, 0.25, 1.5]))
        if f in (7, 10, 14): return float(rng.integers(0, 5)) + 0.5
        return float(rng.integers(-1, 80)) + float(rng.choice([0.0, 0.5]))

    def grow(depth):
        me = len(feat)
        feat.append(0); thr.append(0.0); left.append(0); right.append(0)
        if depth >= max_depth or (depth > 0 and rng.random() < 0.25):
            feat[me] = -1
            left[me] = len(leaves)
            if kind == S.MODEL_RF:
                if normalised:
                    top = 3 if normalised == "coarse" else 50          # coarse fractions: exact vote ties are common
                    n0, n1 = float(rng.integers(0, top)), float(rng.integers(0, top))
                    if n0 + n1 == 0: n1 = 1.0
                    leaves.append((n0 / (n0 + n1), n1 / (n0 + n1)))
                else:
                    leaves.append((float(rng.integers(0, 9)), float(rng.integers(1, 9))))
            else:
                leaves.append((float(np.float32(rng.normal(0, 0.3))), 0.0))
            return me
        f = int(rng.integers(0, n_features))
        feat[me] = f
        thr[me] = threshold(f)
        left[me] = grow(depth + 1)
        right[me] = grow(depth + 1)
        return me

    for _ in range(n_trees):
        roots.append(grow(0))
    return S.FlatForest(kind, np.array(feat, np.int32), np.array(thr, np.float32), np.array(left, np.int32),
                        np.array(right, np.int32), np.array(roots, np.int32), np.array(leaves, np.float64),
                        n_features=n_features, base_score=float(np.float32(rng.normal(0, 0.2))) if kind == S.MODEL_GBT else 0.0,
                        max_depth=0)


@pytest.mark.parametrize("seed", list(range(64)))
def test_random_models(engine, small_callset, seed):
    """Random ensembles (depth 1..10, 1..48 trees, any feature mix, odd thresholds, normalised and raw-count
    payloads, missing groups) through every kernel path: exercises the rank coding, the code tables, the LDS
    layouts and their fallbacks."""
    from oracle import oracle as O
    from variantcalling_amd import schema as S
    from variantcalling_amd.engine import configure
    cs = small_callset
    rng = np.random.default_rng(_OFF + 9000 + seed)
    kind = S.MODEL_RF if rng.random() < 0.7 else S.MODEL_GBT
    nf = 17 + len(cs.tracks)
    forests = []
    for g in range(3):
        if rng.random() < 0.15 and g > 0:
            forests.append(None)                              # no model for this variant type: score 0, PASS
            continue
        forests.append(_random_forest(rng, kind, nf, int(rng.integers(1, 49)), int(rng.integers(1, 11)),
                                      normalised=[True, True, "coarse", "coarse", False][int(rng.integers(0, 5))]))
    from variantcalling_amd import model_io
    for f in forests:
        if f is not None:
            f.max_depth = max(model_io._depth(f.left, f.right, f.feature, int(r)) for r in f.tree_root)
    configure(engine, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests, "TGCA", 10, 10, True)
    exp = O.filter_variants(cs.variants, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
    for path in (0, 65536, 65536 | 1024, 256):
        engine.set_kernel_variant(path)
        res = engine.filter_variants(cs.variants)
        what = f"seed {seed} path {path} kind {kind}"
        assert np.array_equal(res.filter, exp.filter), what
        assert np.array_equal(res.flags, exp.flags), what
        if kind == S.MODEL_RF:
            assert np.array_equal(res.tree_score, exp.tree_score), what
        else:
            assert np.max(np.abs(res.tree_score - exp.tree_score)) <= 1e-6, what
    engine.set_kernel_variant(0)


@pytest.mark.parametrize("seed", list(range(40)))
def test_random_side_tables_and_clustered_variants(engine, small_callset, frozen_models, seed):
    """Variants piled into a few clusters (dense tiles, repeated positions, contig starts and ends) against random
    interval tables: empty, sparse, or an interval every few bases (the staged slices overflow the LDS pool), runs
    of random lengths, blacklist keys on and next to variants and on tile boundaries."""
    import copy
    from oracle import oracle as O
    from variantcalling_amd import schema as S
    from variantcalling_amd.engine import configure
    from variantcalling_amd.io import bed
    cs = small_callset
    ref = cs.ref
    rng = np.random.default_rng(_OFF + 11_000 + seed)
    nc = ref.n_contigs
    clen = np.diff(ref.contig_off).astype(np.int64)
    n = int(rng.choice([300, 5_000, 30_000]))
    vt = copy.deepcopy(cs.variants.slice(0, n))
    contig = np.sort(rng.integers(0, nc, n)).astype(np.uint16) if rng.random() < 0.8 else np.full(n, int(rng.integers(0, nc)), np.uint16)
    pos = np.zeros(n, np.int64)
    for c in range(nc):
        m = contig == c
        k = int(m.sum())
        if not k:
            continue
        centers = rng.choice(np.array([1, 2, clen[c] // 2, clen[c] - 1, clen[c]] + list(rng.integers(1, clen[c], 3))),
                             size=k)
        spread = int(rng.choice([1, 50, 5000, 2_000_000]))
        pos[m] = np.sort(np.clip(centers + rng.integers(-spread, spread + 1, k), 1, clen[c]))
    vt.contig, vt.pos = contig, pos.astype(np.int32)
    # longest allele must still fit the contig for deletions: keep the table valid
    vt.validate()

    def random_track(name, density, merge=True):
        cc, ss, ee = [], [], []
        for c in range(nc):
            k = int(rng.poisson(density * clen[c])) if density > 0 else 0
            k = min(k, 400_000)
            st = rng.integers(0, clen[c], k)
            ln = rng.geometric(float(rng.choice([0.5, 0.05, 0.002])), k)
            if k and rng.random() < 0.5:
                st[0], ln[0] = 0, 7                                  # abuts the contig start
                st[-1], ln[-1] = clen[c] - 3, 3                      # and its end
            cc.append(np.full(k, c)); ss.append(st); ee.append(np.minimum(st + ln, clen[c]))
        return bed.track_from_arrays(np.concatenate(cc).astype(np.int64), np.concatenate(ss).astype(np.int64),
                                     np.concatenate(ee).astype(np.int64), nc, name, merge)

    dens = [float(rng.choice([0.0, 1e-6, 1e-4, 3e-3])) for _ in range(3)]
    if rng.random() < 0.4:
        # an interval every ~4 bases around the variant clusters of contig 0: dense tiles
        lo = max(0, int(pos[contig == contig[0]].min()) - 2000)
        st = np.arange(lo, min(lo + 400_000, clen[contig[0]] - 2), 4)
        dense = bed.track_from_arrays(np.full(st.size, int(contig[0]), np.int64), st, st + 2, nc, "dense", True)
    else:
        dense = None
    tracks = [random_track(f"t{j}", dens[j]) for j in range(3)]
    if dense is not None:
        tracks[int(rng.integers(0, 3))] = dense
    runs = random_track("runs", float(rng.choice([0.0, 2e-4, 2e-3])), merge=True)

    def overlapping(tr):                                             # overlapping intervals whose ends still ascend per contig
        ends = tr.ends.astype(np.int64).copy()
        for c in range(nc):
            a, b = int(tr.contig_ptr[c]), int(tr.contig_ptr[c + 1])
            if b > a:
                ends[a:b] = np.maximum.accumulate(ends[a:b])
        return S.IntervalTrack(tr.starts, ends.astype(np.int32), tr.contig_ptr, tr.name)
    if rng.random() < 0.3:
        j = int(rng.integers(0, 3))
        if tracks[j] is not dense:
            tracks[j] = overlapping(random_track("ov", 3e-4, merge=False))
    if rng.random() < 0.15:
        runs = overlapping(random_track("ovruns", 2e-4, merge=False))   # overlapping runs: the universal kernel takes over
    keys = vt.keys()
    bl_parts = [keys[:: int(rng.integers(1, 7))], keys[255::256], keys[::256] + np.uint64(1), keys[::97] - np.uint64(1),
                (rng.integers(0, nc, 2000).astype(np.uint64) << np.uint64(32)) | rng.integers(1, int(clen.min()), 2000).astype(np.uint64)]
    bl = np.unique(np.concatenate(bl_parts)) if rng.random() < 0.85 else None
    hp_len, hp_dist = int(rng.choice([1, 10])), int(rng.choice([0, 3, 10, 1000]))
    forests = frozen_models[RF]
    configure(engine, ref, runs, tracks, bl, forests, "TGCA", hp_len, hp_dist, True)
    exp = O.filter_variants(vt, ref, runs, tracks, bl, forests, hpol_len=hp_len, hpol_dist=hp_dist)
    for path in (0, 65536, 256):
        engine.set_kernel_variant(path)
        res = engine.filter_variants(vt)
        what = f"seed {seed} path {path}"
        assert np.array_equal(res.flags, exp.flags), what
        assert np.array_equal(res.filter, exp.filter), what
        assert np.array_equal(res.tree_score, exp.tree_score), what
    engine.set_kernel_variant(0)


@pytest.mark.parametrize("seed", list(range(30)))
def test_random_alleles_on_a_homopolymer_rich_reference(engine, frozen_models, seed):
    """A small synthetic reference full of long homopolymers, N blocks and tiny contigs; variants with hand-made
    alleles: hmer insertions / deletions of every length (runs longer than the 12-base look-ahead and the 48-byte
    window), long deletions, multi-base substitutions, N alleles, alleles that run over the contig end."""
    from oracle import oracle as O
    from variantcalling_amd import schema as S
    from variantcalling_amd.engine import configure
    rng = np.random.default_rng(_OFF + 13_000 + seed)
    nc = int(rng.choice([1, 3, 6]))
    parts = []
    for c in range(nc):
        L = int(rng.choice([7, 60, 5_000, 60_000]))
        seq = rng.integers(1, 5, L).astype(np.uint8)
        for _ in range(int(rng.integers(0, 1 + L // 200))):          # homopolymers, some far longer than the window
            a = int(rng.integers(0, L)); ln = int(rng.choice([2, 5, 11, 12, 13, 30, 47, 48, 49, 200]))
            seq[a: a + ln] = int(rng.integers(1, 5))
        for _ in range(int(rng.integers(0, 3))):                     # N blocks
            a = int(rng.integers(0, L)); seq[a: a + int(rng.choice([1, 20, 300]))] = 0
        parts.append(seq)
    off = np.concatenate([[0], np.cumsum([p.size for p in parts])]).astype(np.int64)
    ref = S.Reference(np.concatenate(parts), off, [f"c{c}" for c in range(nc)])
    rows = []
    n = int(rng.choice([50, 700, 4000]))
    for _ in range(n):
        c = int(rng.integers(0, nc)); L = parts[c].size
        pos = int(rng.integers(1, L + 1))
        kind = rng.choice(["snv", "ins_h", "del_h", "ins", "del", "mnp", "nall", "longdel"])
        r0 = int(parts[c][pos - 1]) or 1
        nxt = int(parts[c][pos]) if pos < L else int(rng.integers(1, 5))
        if kind == "snv":
            refa, alta = [r0], [int(rng.integers(0, 5))]
        elif kind == "ins_h":
            refa, alta = [r0], [r0] + [nxt or 1] * int(rng.choice([1, 2, 8, 9, 10, 30]))
        elif kind == "del_h":
            k = int(rng.choice([1, 2, 8, 9, 12, 40]))
            refa, alta = [r0] + [int(x) for x in parts[c][pos: pos + k]], [r0]
            if len(refa) == 1:
                refa = [r0, nxt or 1]
        elif kind == "ins":
            refa, alta = [r0], [r0] + [int(x) for x in rng.integers(0, 5, int(rng.integers(1, 12)))]
        elif kind == "del":
            refa, alta = [r0] + [int(x) for x in rng.integers(1, 5, int(rng.integers(1, 12)))], [r0]
        elif kind == "mnp":
            k = int(rng.integers(2, 6))
            refa, alta = [int(x) for x in rng.integers(1, 5, k)], [int(x) for x in rng.integers(0, 5, k)]
        elif kind == "nall":
            refa, alta = [0], [0, 0]
        else:
            k = int(rng.choice([60, 300, 2000]))
            refa, alta = [r0] + [int(x) for x in rng.integers(1, 5, k)], [r0]
        if refa == alta:
            alta = alta + [1]
        rows.append((c, pos, refa, alta))
    rows.sort(key=lambda r: (r[0], r[1]))
    pool, ro, ao = [], [], []
    for _, _, refa, alta in rows:
        ro.append(len(pool)); pool += refa
        ao.append(len(pool)); pool += alta
    m = len(rows)
    vt = S.VariantTable(
        contig=np.array([r[0] for r in rows], np.uint16), pos=np.array([r[1] for r in rows], np.int32),
        ref_len=np.array([len(r[2]) for r in rows], np.uint16), alt_len=np.array([len(r[3]) for r in rows], np.uint16),
        ref_off=np.array(ro, np.uint32), alt_off=np.array(ao, np.uint32), alleles=np.array(pool, np.uint8),
        qual=rng.exponential(60, m).astype(np.float32), sor=rng.lognormal(0, 0.7, m).astype(np.float32),
        dp=rng.poisson(30, m).astype(np.int32), ad_ref=rng.poisson(15, m).astype(np.int32),
        ad_alt=rng.poisson(15, m).astype(np.int32), gq=rng.integers(0, 100, m).astype(np.uint8), gt=np.ones(m, np.uint8))
    vt.validate()
    flow = str(rng.choice(["TGCA", "ACGT", "GATC"]))
    forests = frozen_models[RF]
    # the frozen model tests three track features: give it three (empty) tracks, no runs, no blacklist
    empty = [S.IntervalTrack(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(nc + 1, np.int32), f"t{j}") for j in range(3)]
    configure(engine, ref, None, empty, None, forests, flow, 10, 10, True)
    exp = O.filter_variants(vt, ref, None, empty, None, forests, flow_order=flow)
    X_exp = None
    for path in (0, 65536, 256):
        engine.set_kernel_variant(path)
        res = engine.filter_variants(vt)
        what = f"seed {seed} path {path}"
        if not np.array_equal(res.tree_score, exp.tree_score):                       # say where: it localises a kernel bug
            bad = np.flatnonzero(res.tree_score != exp.tree_score)
            indel = vt.ref_len != vt.alt_len
            what += (f": {bad.size} of {m} scores differ ({int(indel[bad].sum())} indels), first rows {bad[:6].tolist()}, "
                     f"contigs {vt.contig[bad[:6]].tolist()}, pos {vt.pos[bad[:6]].tolist()}, got {res.tree_score[bad[:3]].tolist()} "
                     f"want {exp.tree_score[bad[:3]].tolist()}; contig lengths {np.diff(ref.contig_off).tolist()}")
        assert np.array_equal(res.flags, exp.flags), what
        assert np.array_equal(res.filter, exp.filter), what
        assert np.array_equal(res.tree_score, exp.tree_score), what
    engine.set_kernel_variant(0)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_shards_scored_alone_reassemble_to_the_whole(engine, small_callset, frozen_models, world):
    """What every rank of a multi-GPU run does (score its equal-count slice on its own) gives, slice
    by slice, the columns of the single-GPU run: no result depends on a neighbour outside the slice."""
    from variantcalling_amd import shard, schema as S
    from variantcalling_amd.engine import configure
    cs = small_callset
    configure(engine, cs.ref, cs.runs, cs.tracks, cs.blacklist, frozen_models[RF], "TGCA", 10, 10, True)
    whole = engine.filter_variants(cs.variants)
    parts, counts = [], []
    cap = shard.shard_cap(cs.variants.n, world)
    for r in range(world):
        mine = shard.shard_of(cs.variants, r, world)
        parts.append(shard.pad_result(engine.filter_variants(mine), cap))
        counts.append(mine.n)
    got = shard.reassemble(parts, counts)
    assert np.array_equal(got.filter, whole.filter) and np.array_equal(got.flags, whole.flags)
    assert np.array_equal(got.tree_score, whole.tree_score)
