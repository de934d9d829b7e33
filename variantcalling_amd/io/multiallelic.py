"""Multi-allelic records and spanning deletions around the scoring pass (host logic).

The scoring pass works on one ALT allele per row.  The reference's current filtering code splits multi-allelic records
and treats the spanning-deletion allele `*` specially - its fixtures are
test/resources/unit/filtering/test_multiallelics/cleanup_multiallelics_{input,expected}.h5 and
test/resources/unit/filtering/test_spandel/{spanning_deletions,expected_result_split_multiallelic}.pkl - but code and
fixtures are absent here (un-vendored submodule, un-pulled LFS objects), so the rule below is BUILDER-DEFINED and kept
to what the VCF specification fixes:

  * a record with k ALT alleles becomes one row per ALT allele that is an actual sequence; row j is featurised with ALT j
    and scored with AD[0] as the reference depth and AD[j] as the allele depth (QUAL, SOR, DP, GQ are per record);
  * `*` ("the allele is missing due to an overlapping deletion", VCF 4.2 section 1.6.1 / 5.3) and symbolic / breakend alleles
    (`<DEL>`, `]chr1:5]N`) are not sequences of this locus: they get no row.  A record whose every ALT is of that kind
    keeps one placeholder row (its first ALT read as N) so that it still receives a TREE_SCORE;
  * the record's verdict: PASS when ANY of its rows passes (the record carries at least one believable allele), else
    LOW_SCORE; TREE_SCORE = the best row's score; the position-based flags (HPOL_RUN, COHORT_FP, tracks) are the same
    for every row and are OR-ed.

`expand` reads only the records the readers flagged (`n_alt > 1`, or a first ALT that is not a plain sequence); on a
WGS callset that is a few percent of the rows."""
from __future__ import annotations

import numpy as np

from .. import schema as S

_PLAIN = frozenset(b"ACGTNacgtn")


def _is_sequence(a: bytes) -> bool:
    return len(a) > 0 and all(ch in _PLAIN for ch in a)


def _needs_expansion(vcf) -> np.ndarray:
    t = vcf.table
    n_alt = np.asarray(getattr(vcf, "n_alt", np.ones(t.n, np.uint8)))
    odd = np.zeros(t.n, bool)
    if t.n:
        # a first ALT that is not a plain sequence ('*', '<DEL>', breakends) decodes to code 0 somewhere: candidates only
        first_alt0 = t.alleles[t.alt_off.astype(np.int64)] if t.alleles.size else np.zeros(t.n, np.uint8)
        odd = first_alt0 == 0
    return np.flatnonzero((n_alt > 1) | odd)


def expand(vcf, sample: int = 0):
    """(table with one row per scorable ALT allele, base_row [rows] - the vcf.table row each came from).
    Rows stay sorted by (contig, pos); the rows of one record are adjacent, in ALT order."""
    t = vcf.table
    cand = _needs_expansion(vcf)
    if cand.size == 0:
        return t, np.arange(t.n, dtype=np.int64)
    extra = {c: [] for c in ("contig", "pos", "qual", "sor", "dp", "ad_ref", "ad_alt", "gq", "gt")}
    extra_base, extra_ref, extra_alt = [], [], []
    replace = {}                                    # base row -> (alt bytes, ad_alt) for its FIRST scorable allele
    for k in cand:
        f = vcf.record_line(int(k)).split(b"\t")
        alts = f[4].split(b",")
        ad = []
        if len(f) > 9 + sample:
            keys = f[8].split(b":")
            vals = f[9 + sample].split(b":")
            if b"AD" in keys and keys.index(b"AD") < len(vals):
                for x in vals[keys.index(b"AD")].split(b","):
                    try:
                        ad.append(int(float(x)))
                    except ValueError:
                        ad.append(0)
        good = [j for j, a in enumerate(alts) if _is_sequence(a)]
        if not good:
            continue                                # placeholder row stays as the reader made it
        j0 = good[0]
        if j0 != 0:
            replace[int(k)] = (alts[j0], ad[j0 + 1] if j0 + 1 < len(ad) else 0)
        for j in good[1:]:
            for c in ("contig", "pos", "qual", "sor", "dp", "ad_ref", "gq", "gt"):
                extra[c].append(getattr(t, c)[k])
            extra["ad_alt"].append(ad[j + 1] if j + 1 < len(ad) else 0)
            extra_base.append(int(k))
            extra_ref.append(f[3])
            extra_alt.append(alts[j])
    if not extra_base and not replace:
        return t, np.arange(t.n, dtype=np.int64)
    # new allele pool: the old pool, then the alleles of the replaced / added rows
    pool = [t.alleles]
    cur = int(t.alleles.size)
    ref_off, alt_off = t.ref_off.astype(np.int64).copy(), t.alt_off.astype(np.int64).copy()
    ref_len, alt_len = t.ref_len.copy(), t.alt_len.copy()
    ad_alt = t.ad_alt.copy()
    for k, (a, d) in replace.items():
        if len(a) > 65535:
            raise ValueError("allele longer than 65535 bases")
        pool.append(S.encode_bases(a.decode()))
        alt_off[k], alt_len[k], ad_alt[k] = cur, len(a), d
        cur += len(a)
    e_ro, e_ao = [], []
    for r, a in zip(extra_ref, extra_alt):
        if len(a) > 65535 or len(r) > 65535:
            raise ValueError("allele longer than 65535 bases")
        pool.append(S.encode_bases(r.decode())); e_ro.append(cur); cur += len(r)
        pool.append(S.encode_bases(a.decode())); e_ao.append(cur); cur += len(a)
    if cur >= 1 << 32:
        raise ValueError("allele pool exceeds 4 GiB")
    m = len(extra_base)
    base = np.concatenate([np.arange(t.n, dtype=np.int64), np.asarray(extra_base, np.int64)])
    sub = np.concatenate([np.zeros(t.n, np.int64), np.arange(1, m + 1, dtype=np.int64)])   # ALT order inside a record
    order = np.lexsort((sub, base))                 # base rows are already sorted by (contig, pos): keep each record's rows together
    cols = dict(
        contig=np.concatenate([t.contig, np.asarray(extra["contig"], np.uint16)]),
        pos=np.concatenate([t.pos, np.asarray(extra["pos"], np.int32)]),
        ref_len=np.concatenate([ref_len, np.asarray([len(r) for r in extra_ref], np.uint16)]),
        alt_len=np.concatenate([alt_len, np.asarray([len(a) for a in extra_alt], np.uint16)]),
        ref_off=np.concatenate([ref_off, np.asarray(e_ro, np.int64)]).astype(np.uint32),
        alt_off=np.concatenate([alt_off, np.asarray(e_ao, np.int64)]).astype(np.uint32),
        qual=np.concatenate([t.qual, np.asarray(extra["qual"], np.float32)]),
        sor=np.concatenate([t.sor, np.asarray(extra["sor"], np.float32)]),
        dp=np.concatenate([t.dp, np.asarray(extra["dp"], np.int32)]),
        ad_ref=np.concatenate([t.ad_ref, np.asarray(extra["ad_ref"], np.int32)]),
        ad_alt=np.concatenate([ad_alt, np.asarray(extra["ad_alt"], np.int32)]),
        gq=np.concatenate([t.gq, np.asarray(extra["gq"], np.uint8)]),
        gt=np.concatenate([t.gt, np.asarray(extra["gt"], np.uint8)]))
    out = S.VariantTable(alleles=np.concatenate(pool).astype(np.uint8), **{c: np.ascontiguousarray(v[order]) for c, v in cols.items()})
    out.validate()
    return out, base[order]


def collapse(res: S.FilterResult, base_row: np.ndarray, n: int) -> S.FilterResult:
    """Per-record verdict from the per-allele rows: PASS if any row passes, best score, OR of the flags."""
    if base_row.size == n and np.array_equal(base_row, np.arange(n)):
        return res
    score = np.full(n, -np.inf, np.float32)
    np.maximum.at(score, base_row, res.tree_score)
    filt = np.full(n, S.FILTER_LOW_SCORE, np.uint8)
    np.minimum.at(filt, base_row, res.filter)       # PASS = 0 < LOW_SCORE = 1
    flags = np.zeros(n, np.uint8)
    np.bitwise_or.at(flags, base_row, res.flags)
    score[~np.isfinite(score)] = 0.0
    return S.FilterResult(score, filt, flags)
