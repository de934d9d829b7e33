"""Seeded synthetic WGS-shaped inputs (SURVEY.md §8(d) "Synthetic inputs").

Data generation only - no featurize/score logic lives here.  The reference genome is
synthesised by tiling 1 kb blocks of the REAL hg38 chr1:1-5,000,000 slice recovered from
the reference repo's RTG SDF fixture (tests/golden/hg38_chr1_head.npz, minted by
tests/golden/make_hg38_fixture.py), so homopolymer-run statistics match a real genome
(SURVEY.md App. D).
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np

from .schema import IntervalTrack, Reference, VariantTable

_HERE = os.path.dirname(os.path.abspath(__file__))
HG38_FIXTURE = os.path.join(_HERE, "..", "tests", "golden", "hg38_chr1_head.npz")

# hg38 primary contig lengths chr1..22, X, Y (public assembly report); used as proportions
HG38_LENGTHS = np.array([
    248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
    138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
    83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415],
    dtype=np.int64)
HG38_NAMES = [f"chr{i}" for i in list(range(1, 23)) + ["X", "Y"]]
BLOCK = 1000


def load_hg38_slice(name: str = "chr1", path: str = HG38_FIXTURE) -> np.ndarray:
    """u8 codes (N,A,C,G,T = 0..4) of the real hg38 slice stored in the golden fixture."""
    z = np.load(path)
    n = int(z[f"{name}_len"])
    packed = z[f"{name}_packed"]
    two = np.empty((packed.size, 4), dtype=np.uint8)
    for k in range(4):
        two[:, k] = (packed >> (2 * k)) & 3
    codes = (two.reshape(-1)[:n] + 1).astype(np.uint8)
    for s, e in z[f"{name}_nruns"]:
        codes[s:e] = 0
    return codes


@dataclass
class SynthGenome:
    ref: Reference
    block_ids: np.ndarray        # which real 1 kb block each genome tile came from
    blocks: np.ndarray           # [n_blocks, BLOCK] u8, N-free real blocks


def make_genome(total_len: int, n_contigs: int = 24, seed: int = 20260116,
                path: str = HG38_FIXTURE) -> SynthGenome:
    """Tile/shuffle N-free 1 kb blocks of real chr1 into `n_contigs` hg38-proportioned contigs."""
    rng = np.random.default_rng(seed)
    real = load_hg38_slice("chr1", path)
    nb = real.size // BLOCK
    blocks = real[: nb * BLOCK].reshape(nb, BLOCK)
    blocks = np.ascontiguousarray(blocks[(blocks != 0).all(axis=1)])
    props = HG38_LENGTHS[:n_contigs] / HG38_LENGTHS[:n_contigs].sum()
    lens = np.maximum(BLOCK, (props * total_len / BLOCK).astype(np.int64) * BLOCK)
    n_tiles = int(lens.sum() // BLOCK)
    ids = rng.integers(0, blocks.shape[0], size=n_tiles, dtype=np.int32)
    codes = blocks[ids].reshape(-1)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    names = HG38_NAMES[:n_contigs] if n_contigs <= 24 else [f"ctg{i}" for i in range(n_contigs)]
    return SynthGenome(Reference(codes, off, names), ids, blocks)


def _global_to_contig(ref: Reference, g: np.ndarray):
    c = np.searchsorted(ref.contig_off, g, side="right") - 1
    return c.astype(np.uint16), (g - ref.contig_off[c] + 1).astype(np.int32)


def make_variants(ref: Reference, n: int, seed: int = 20260116, snv_only: bool = False,
                  indel_frac: float = 0.18, hmer_frac: float = 0.60,
                  mnp_frac: float = 0.0) -> VariantTable:
    """N sorted, unique-position variants: 82 % SNV / 18 % indel (60 % of indels hmer changes)."""
    rng = np.random.default_rng(seed + 1)
    total = int(ref.contig_off[-1])
    codes = ref.codes
    margin = 80
    # draws needed for n DISTINCT positions: total * -ln(1 - n / total) (+2 %); the 5 % surplus of the sparse callsets
    # (5 M, 50 M of 3.1 G positions: unchanged, the goldens depend on it) stops sufficing at 500 M
    span = total - 2 * margin
    need = int(-span * np.log1p(-min(n / span, 0.999)) * 1.02) if n < span else n
    g = np.unique(rng.integers(margin, total - margin, size=max(int(n * 1.05) + 64, need), dtype=np.int64))
    if g.size < n:
        raise ValueError("genome too small for the requested number of variants")
    g = np.sort(rng.choice(g, size=n, replace=False))
    # keep clear of contig ends (edge behaviour is tested with hand-made cases instead)
    c = np.searchsorted(ref.contig_off, g, side="right") - 1
    g = np.clip(g, ref.contig_off[c] + margin, ref.contig_off[c + 1] - margin - 1)

    kind = np.zeros(n, dtype=np.int8)          # 0 snv, 1 hmer indel, 2 non-hmer indel, 3 mnp
    if not snv_only:
        u = rng.random(n)
        kind[u < indel_frac * hmer_frac] = 1
        kind[(u >= indel_frac * hmer_frac) & (u < indel_frac)] = 2
        kind[(u >= indel_frac) & (u < indel_frac + mnp_frac)] = 3
    is_ins = rng.random(n) < 0.5
    ilen = np.minimum(rng.geometric(0.45, size=n), 50).astype(np.int64)

    # hmer indels: move the anchor to the base just before the next homopolymer run
    hm = np.where(kind == 1)[0]
    if hm.size:
        W = 64
        win = codes[g[hm, None] + np.arange(W)[None, :]]
        change = win[:, 1:] != win[:, :-1]
        k = np.argmax(change, axis=1) + 1                 # first index whose base differs
        k[~change.any(axis=1)] = 1
        g[hm] = g[hm] + k - 1                             # anchor; run starts at anchor + 1
        win = codes[g[hm, None] + 1 + np.arange(W)[None, :]]
        same = win == win[:, :1]
        run = np.where(same.all(axis=1), W, np.argmin(same, axis=1)).astype(np.int64)
        # deletion needs the run to outlive the deleted bases; otherwise make it an insertion
        dl = ~is_ins[hm]
        ilen[hm] = np.where(dl, np.minimum(ilen[hm], np.maximum(run - 1, 1)), ilen[hm])
        is_ins[hm] = is_ins[hm] | (run < 2)

    # unique positions again after the anchor moves
    order = np.argsort(g, kind="stable")
    g, kind, is_ins, ilen = g[order], kind[order], is_ins[order], ilen[order]
    keep = np.concatenate([[True], g[1:] != g[:-1]])
    g, kind, is_ins, ilen = g[keep], kind[keep], is_ins[keep], ilen[keep]
    n = g.size

    ref_base = codes[g]
    ref_len = np.ones(n, dtype=np.int64)
    alt_len = np.ones(n, dtype=np.int64)
    ins = (kind > 0) & (kind < 3) & is_ins
    dele = (kind > 0) & (kind < 3) & ~is_ins
    mnp = kind == 3
    alt_len[ins] = 1 + ilen[ins]
    ref_len[dele] = 1 + ilen[dele]
    ref_len[mnp] = alt_len[mnp] = 2
    tot = ref_len + alt_len
    off = np.concatenate([[0], np.cumsum(tot)])
    pool = np.zeros(int(off[-1]), dtype=np.uint8)
    ref_off = off[:-1]
    alt_off = ref_off + ref_len
    # reference alleles straight from the genome
    idx_rows = np.repeat(np.arange(n), ref_len)
    within = np.arange(int(ref_len.sum())) - np.repeat(np.cumsum(ref_len) - ref_len, ref_len)
    pool[np.repeat(ref_off, ref_len) + within] = codes[g[idx_rows] + within]
    # alt alleles
    pool[alt_off] = ref_base                                   # anchor base (indels)
    snv = kind == 0
    shift = rng.integers(1, 4, size=n)
    pool[alt_off[snv]] = ((ref_base[snv].astype(np.int64) - 1 + shift[snv]) % 4 + 1).astype(np.uint8)
    if mnp.any():
        m = np.where(mnp)[0]
        for j in range(2):
            rb = pool[ref_off[m] + j].astype(np.int64)
            pool[alt_off[m] + j] = ((rb - 1 + rng.integers(1, 4, size=m.size)) % 4 + 1).astype(np.uint8)
    if ins.any():
        m = np.where(ins)[0]
        rows = np.repeat(m, ilen[m])
        w = np.arange(int(ilen[m].sum())) - np.repeat(np.cumsum(ilen[m]) - ilen[m], ilen[m])
        rnd = rng.integers(1, 5, size=rows.size).astype(np.uint8)
        nxt = codes[g[rows] + 1]                               # base that opens the run
        pool[alt_off[rows] + 1 + w] = np.where(kind[rows] == 1, nxt, rnd)

    contig, pos = _global_to_contig(ref, g)
    dp = rng.poisson(30, size=n).astype(np.int32)
    pm = rng.random(n)
    p = np.where(pm < 0.6, 0.5, np.where(pm < 0.95, 1.0, rng.random(n) * 0.2))
    ad_alt = rng.binomial(dp, p).astype(np.int32)
    ad_ref = (dp - ad_alt).astype(np.int32)
    qual = np.minimum(np.round(rng.exponential(60.0, size=n) * (0.2 + 1.6 * p), 2), 3000.0)
    sor = np.round(rng.lognormal(0.0, 0.7, size=n), 3)
    gq = np.minimum(99, (qual * 0.8 + rng.integers(0, 20, size=n))).astype(np.uint8)
    gt = np.where(p == 1.0, 2, 1).astype(np.uint8)
    vt = VariantTable(
        contig=contig, pos=pos, ref_len=ref_len.astype(np.uint16), alt_len=alt_len.astype(np.uint16),
        ref_off=ref_off.astype(np.uint32), alt_off=alt_off.astype(np.uint32), alleles=pool,
        qual=qual.astype(np.float32), sor=sor.astype(np.float32), dp=dp, ad_ref=ad_ref,
        ad_alt=ad_alt, gq=gq, gt=gt)
    vt.validate()
    return vt


def runs_track_from_genome(sg: SynthGenome, min_len: int = 10) -> IntervalTrack:
    """Homopolymer runs >= min_len taken from the genome itself (per tile, so O(#blocks))."""
    blocks = sg.blocks
    nb = blocks.shape[0]
    brk = np.ones((nb, BLOCK + 1), dtype=bool)
    brk[:, 1:BLOCK] = blocks[:, 1:] != blocks[:, :-1]
    per_block = []
    for b in range(nb):
        e = np.flatnonzero(brk[b])
        ln = np.diff(e)
        sel = ln >= min_len
        per_block.append(np.stack([e[:-1][sel], e[1:][sel]], axis=1))
    counts = np.array([p.shape[0] for p in per_block], dtype=np.int64)
    flat = np.concatenate(per_block, axis=0) if counts.sum() else np.zeros((0, 2), np.int64)
    first = np.concatenate([[0], np.cumsum(counts)])
    tiles_cnt = counts[sg.block_ids]
    tile_idx = np.repeat(np.arange(sg.block_ids.size, dtype=np.int64), tiles_cnt)
    within = np.arange(int(tiles_cnt.sum())) - np.repeat(np.cumsum(tiles_cnt) - tiles_cnt, tiles_cnt)
    src = first[sg.block_ids[tile_idx]] + within
    gs = tile_idx * BLOCK + flat[src, 0]
    ge = tile_idx * BLOCK + flat[src, 1]
    return _track_from_global(sg.ref, gs, ge, "runs")


def _track_from_global(ref: Reference, gs: np.ndarray, ge: np.ndarray, name: str) -> IntervalTrack:
    c = np.searchsorted(ref.contig_off, gs, side="right") - 1
    ce = np.searchsorted(ref.contig_off, ge - 1, side="right") - 1
    ok = c == ce
    gs, ge, c = gs[ok], ge[ok], c[ok]
    starts = (gs - ref.contig_off[c]).astype(np.int32)      # BED: 0-based start
    ends = (ge - ref.contig_off[c]).astype(np.int32)        # BED: exclusive end
    ptr = np.searchsorted(c, np.arange(ref.n_contigs + 1)).astype(np.int32)
    return IntervalTrack(starts, ends, ptr, name)


def make_interval_track(ref: Reference, n_intervals: int, mean_len: float, seed: int,
                        name: str) -> IntervalTrack:
    """Sorted, non-overlapping intervals with ~exponential lengths."""
    rng = np.random.default_rng(seed)
    total = int(ref.contig_off[-1])
    gs = np.unique(rng.integers(0, total - 2, size=n_intervals, dtype=np.int64))
    ln = np.maximum(1, rng.exponential(mean_len, size=gs.size)).astype(np.int64)
    nxt = np.concatenate([gs[1:], [total]])
    ge = np.minimum(gs + ln, nxt - 1)
    ok = ge > gs
    return _track_from_global(ref, gs[ok], ge[ok], name)


def make_blacklist(ref: Reference, vt: VariantTable, n_keys: int, hit_frac: float = 0.02,
                   seed: int = 5) -> np.ndarray:
    """Sorted unique u64 keys (contig << 32 | pos); `hit_frac` of them coincide with variants."""
    rng = np.random.default_rng(seed)
    n_hit = min(int(n_keys * hit_frac), vt.n)
    hits = rng.choice(vt.keys(), size=n_hit, replace=False) if n_hit else np.zeros(0, np.uint64)
    total = int(ref.contig_off[-1])
    g = rng.integers(0, total, size=n_keys - n_hit, dtype=np.int64)
    c, p = _global_to_contig(ref, g)
    rnd = (c.astype(np.uint64) << np.uint64(32)) | p.astype(np.uint64)
    return np.unique(np.concatenate([hits, rnd]))


@dataclass
class SynthCallset:
    genome: SynthGenome
    variants: VariantTable
    runs: IntervalTrack
    tracks: list
    blacklist: np.ndarray

    @property
    def ref(self) -> Reference:
        return self.genome.ref


def make_callset(n_variants: int, genome_len: int | None = None, seed: int = 20260116,
                 snv_only: bool = False, n_contigs: int = 24) -> SynthCallset:
    """C2 (snv_only) / C3 callset; side tables scale with N/5M so per-variant bytes are fixed.
    UGVC_SYNTH_CACHE=<dir> (measurement scripts only): the generated callset is kept there as a pickle and re-read by later
    processes - a profiling session runs bench.py a dozen times on the same 5 M-variant callset, ~50 s of generation each."""
    cache = os.environ.get("UGVC_SYNTH_CACHE")
    if cache:
        import pickle
        path = os.path.join(cache, f"callset_{n_variants}_{genome_len}_{seed}_{int(snv_only)}_{n_contigs}.pkl")
        if os.path.exists(path):
            with open(path, "rb") as fh:
                return pickle.load(fh)
        os.environ.pop("UGVC_SYNTH_CACHE")
        try:
            cs = make_callset(n_variants, genome_len, seed, snv_only, n_contigs)
        finally:
            os.environ["UGVC_SYNTH_CACHE"] = cache
        try:
            os.makedirs(cache, exist_ok=True)
            tmp = path + f".{os.getpid()}.tmp"
            with open(tmp, "wb") as fh:
                pickle.dump(cs, fh, protocol=5)
            os.replace(tmp, path)
        except OSError:
            pass
        return cs
    if genome_len is None:
        genome_len = int(HG38_LENGTHS.sum() * min(1.0, n_variants / 5_000_000))
        genome_len = max(genome_len, 2_000_000)
    scale = n_variants / 5_000_000
    sg = make_genome(genome_len, n_contigs=n_contigs, seed=seed)
    vt = make_variants(sg.ref, n_variants, seed=seed, snv_only=snv_only)
    runs = runs_track_from_genome(sg, min_len=10)
    tracks = [
        make_interval_track(sg.ref, max(1, int(500_000 * scale)), 300.0, seed + 11, "LCR-hs38"),
        make_interval_track(sg.ref, max(1, int(700_000 * scale)), 200.0, seed + 12, "exome.twist"),
        make_interval_track(sg.ref, max(1, int(3_000_000 * scale)), 1000.0, seed + 13, "mappability.0"),
    ]
    bl = make_blacklist(sg.ref, vt, max(1, int(1_000_000 * scale)), 0.02, seed + 21)
    return SynthCallset(sg, vt, runs, tracks, bl)


def make_pileup(n_loci: int, seed: int = 99, mean_depth: float = 30.0):
    """CSR locus -> read observations for the pileup tally (builder-defined).

    obs u16 = allele (bits 0-1: 0 ref, 1 alt, 2 other) | strand << 2 | base quality << 3."""
    rng = np.random.default_rng(seed)
    d = rng.poisson(mean_depth, size=n_loci).astype(np.int64)
    off = np.concatenate([[0], np.cumsum(d)]).astype(np.int64)
    m = int(off[-1])
    pm = rng.random(n_loci)
    p = np.where(pm < 0.6, 0.5, np.where(pm < 0.95, 1.0, rng.random(n_loci) * 0.2))
    pr = np.repeat(p, d)
    u = rng.random(m)
    allele = np.where(u < pr * 0.98, 1, np.where(u < pr * 0.98 + 0.01, 2, 0)).astype(np.uint16)
    strand = (rng.random(m) < 0.5).astype(np.uint16)
    bq = np.clip(np.round(rng.normal(30, 6, size=m)), 2, 45).astype(np.uint16)
    obs = (allele | (strand << 2) | (bq << 3)).astype(np.uint16)
    return off, obs
