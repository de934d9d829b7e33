"""Hand-made edge-case variant tables shared by CPU and GPU tests (not a test module)."""
import numpy as np

from variantcalling_amd import schema as S


def table_from_records(recs):
    """recs: list of (contig, pos, ref_str, alt_str, qual, sor, dp, ad_ref, ad_alt, gq)."""
    recs = sorted(recs, key=lambda r: (r[0], r[1]))
    n = len(recs)
    pool, ro, ao = [], [], []
    for r in recs:
        ro.append(len(pool)); pool.extend(S.encode_bases(r[2]).tolist())
        ao.append(len(pool)); pool.extend(S.encode_bases(r[3]).tolist())
    col = lambda k, dt: np.array([r[k] for r in recs], dtype=dt)
    vt = S.VariantTable(
        contig=col(0, np.uint16), pos=col(1, np.int32),
        ref_len=np.array([len(r[2]) for r in recs], np.uint16), alt_len=np.array([len(r[3]) for r in recs], np.uint16),
        ref_off=np.array(ro, np.uint32), alt_off=np.array(ao, np.uint32), alleles=np.array(pool, np.uint8),
        qual=col(4, np.float32), sor=col(5, np.float32), dp=col(6, np.int32), ad_ref=col(7, np.int32),
        ad_alt=col(8, np.int32), gq=col(9, np.uint8), gt=np.ones(n, np.uint8))
    vt.validate()
    return vt


def edge_table(ref: S.Reference, seed=3, n_random=4000):
    """Variants at contig starts/ends, inside/next to N runs, long hmers, MNPs, long indels, dp=0."""
    rng = np.random.default_rng(seed)
    recs = []
    chars = S.CODE_TO_CHAR

    def base(c, p):   # 1-based
        return chars[int(ref.codes[ref.contig_off[c] + p - 1])]

    def add(c, p, ref_s, alt_s, dp=30, ada=15):
        recs.append((c, p, ref_s, alt_s, float(rng.integers(1, 400)), float(rng.random() * 4), dp, max(dp - ada, 0),
                     ada, int(rng.integers(0, 99))))

    for c in range(ref.n_contigs):
        L = ref.contig_len(c)
        for p in list(range(1, 13)) + list(range(L - 12, L + 1)):
            b = base(c, p)
            alt = "ACGT"[(("ACGT".find(b) if b in "ACGT" else 0) + 1) % 4]
            add(c, p, b, alt)
            add(c, p, b, b + "A" * int(rng.integers(1, 4)))                 # insertion at the edge
            if p + 3 <= L:
                add(c, p, "".join(base(c, p + k) for k in range(3)), b)     # deletion at the edge
    # around the N runs of real chr1 (10000, 207666-257666, ...)
    codes = ref.codes[: ref.contig_off[1]]
    isn = (codes == 0).astype(np.int8)
    edges = np.flatnonzero(np.diff(isn)) + 1
    for e in edges[:12]:
        for p in range(max(1, e - 8), min(codes.size, e + 9)):
            b = base(0, p)
            add(0, p, b, "ACGT"[int(rng.integers(0, 4))] if b == "N" else "ACGT"[("ACGT".find(b) + 2) % 4])
            add(0, p, b, b + "GG")
    # long homopolymers: find runs >= 15 in chr1 and put hmer ins/del + SNVs next to them
    brk = np.flatnonzero(np.concatenate([[True], codes[1:] != codes[:-1], [True]]))
    ln = np.diff(brk)
    for s in brk[:-1][(ln >= 15) & (codes[brk[:-1]] != 0)][:40]:
        anchor = int(s)            # 1-based pos of the base before the run == 0-based run start
        if anchor < 2:
            continue
        runb = base(0, anchor + 1)
        add(0, anchor, base(0, anchor), base(0, anchor) + runb * 2)
        add(0, anchor, base(0, anchor) + runb * 3, base(0, anchor))
        add(0, anchor + 5, base(0, anchor + 5), "ACGT"[("ACGT".find(runb) + 1) % 4])
        add(0, max(1, anchor - 6), base(0, max(1, anchor - 6)), "ACGT"[("ACGT".find(base(0, max(1, anchor - 6))) + 1) % 4])
    # random interior: MNPs, complex and long indels, dp = 0, huge qual
    for _ in range(n_random):
        c = int(rng.integers(0, ref.n_contigs))
        L = ref.contig_len(c)
        p = int(rng.integers(60, L - 120))
        k = rng.random()
        b = base(c, p)
        if k < 0.15:
            m = int(rng.integers(2, 5))
            r = "".join(base(c, p + j) for j in range(m))
            a = "".join("ACGT"[int(rng.integers(0, 4))] for _ in range(m))
            if a != r:
                add(c, p, r, a)
        elif k < 0.4:
            m = int(rng.integers(1, 60))
            add(c, p, "".join(base(c, p + j) for j in range(m + 1)), b, dp=int(rng.integers(0, 3)), ada=0)
        elif k < 0.65:
            m = int(rng.integers(1, 60))
            ins = "".join("ACGT"[int(rng.integers(0, 4))] for _ in range(m)) if rng.random() < 0.5 else base(c, p + 1) * m
            add(c, p, b, b + ins)
        elif k < 0.75:   # complex: both alleles longer than 1
            add(c, p, "".join(base(c, p + j) for j in range(3)), b + "TT" + "A" * int(rng.integers(1, 5)))
        else:
            add(c, p, b, "ACGT"[int(rng.integers(0, 4))] if b == "N" else "ACGT"[("ACGT".find(b) + 1) % 4],
                dp=int(rng.integers(0, 200)), ada=int(rng.integers(0, 100)))
    # de-duplicate positions (first record per locus wins)
    seen, out = set(), []
    for r in sorted(recs, key=lambda r: (r[0], r[1])):
        if (r[0], r[1]) not in seen and r[2] != r[3]:
            seen.add((r[0], r[1]))
            out.append(r)
    return table_from_records(out)


def simple_tracks(ref: S.Reference, seed=5):
    """Runs taken from the real sequence (>= 8 bp) and three random annotation tracks."""
    rng = np.random.default_rng(seed)

    def from_pairs(pairs_by_contig, name):
        s, e, ptr = [], [], [0]
        for c in range(ref.n_contigs):
            pc = sorted(pairs_by_contig.get(c, []))
            s += [a for a, _ in pc]; e += [b for _, b in pc]
            ptr.append(len(s))
        return S.IntervalTrack(np.array(s, np.int32), np.array(e, np.int32), np.array(ptr, np.int32), name)

    runs = {}
    for c in range(ref.n_contigs):
        codes = ref.codes[ref.contig_off[c]: ref.contig_off[c + 1]]
        brk = np.flatnonzero(np.concatenate([[True], codes[1:] != codes[:-1], [True]]))
        ln = np.diff(brk)
        sel = (ln >= 8) & (codes[brk[:-1]] != 0)
        runs[c] = list(zip(brk[:-1][sel].tolist(), brk[1:][sel].tolist()))
    tracks = []
    for t, (cnt, mean) in enumerate([(3000, 300), (5000, 150), (20000, 120)]):
        by = {}
        for c in range(ref.n_contigs):
            L = ref.contig_len(c)
            k = max(1, int(cnt * L / ref.codes.size))
            st = np.unique(rng.integers(0, L - 2, size=k))
            ln = np.maximum(1, rng.exponential(mean, size=st.size)).astype(np.int64)
            en = np.minimum(st + ln, np.concatenate([st[1:], [L]]) - 1)
            ok = en > st
            by[c] = list(zip(st[ok].tolist(), en[ok].tolist()))
        tracks.append(from_pairs(by, f"t{t}"))
    # one contig with no intervals at all in track 0
    tracks[0] = from_pairs({0: list(zip(tracks[0].starts[: tracks[0].contig_ptr[1]].tolist(),
                                        tracks[0].ends[: tracks[0].contig_ptr[1]].tolist()))}, "t0")
    return from_pairs(runs, "runs"), tracks
