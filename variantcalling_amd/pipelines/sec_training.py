"""sec_training: build the cohort's systematic-error (SEC) database on an MI355X.

Stands in for `ugbio_filtering.sec.sec_training.run(argv)` (registered at /root/reference/ugvc/__main__.py:19,56; script
setup.py:45; "SEC ... still undocumented", README.md:12).  The tool's body and flags live in the absent submodule, so
the FLAGS BELOW ARE BUILDER-DEFINED; what is the reference's own is the statistic the database feeds
(`multinomial_likelihood_ratio` over add-one corrected counts, /root/reference/ugvc/utils/stats_utils.py:12-70).

Every cohort sample contributes one observation per called locus: key = contig << 32 | pos, counts = (ad_ref, ad_alt,
other reads = max(dp - ad_ref - ad_alt, 0)) from FORMAT/AD and DP of its first sample (field dictionary:
test/resources/unit/vcfbed/test_vcftools/header.txt:3391-3392).  ONE call of `ugvc_sec_db_build` sorts the pooled
observations by key (the library's LSD radix sort) and sums the counts per locus; the result - sorted unique keys +
k = 3 expected counts - is written as a .npz beside the contig names it was built against."""
from __future__ import annotations

import argparse
import logging
import sys

import numpy as np

logger = logging.getLogger("ugvc")

K_COUNTS = 3      # ref, alt, other


def get_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="sec_training.py", description=run.__doc__)
    ap.add_argument("--inputs", help="VCF of one cohort sample (repeatable)", type=str, action="append", default=[])
    ap.add_argument("--input_list", help="text file with one cohort VCF path per line", type=str)
    ap.add_argument("--reference_file", help="Indexed reference FASTA file (contig names and order)", type=str, required=True)
    ap.add_argument("--output_file", help="SEC database (.npz)", type=str, required=True)
    ap.add_argument("--min_samples", help="keep loci observed in at least this many samples", type=int, default=1)
    ap.add_argument("--device", help="GPU index (MI355X)", type=int, default=0)
    return ap


def observations(vt) -> tuple:
    """(u64 keys, i32 [n, 3] counts) of one sample's calls."""
    keys = (vt.contig.astype(np.uint64) << np.uint64(32)) | vt.pos.astype(np.uint64)
    other = np.maximum(vt.dp.astype(np.int64) - vt.ad_ref - vt.ad_alt, 0)
    counts = np.stack([vt.ad_ref.astype(np.int64), vt.ad_alt.astype(np.int64), other], axis=1).astype(np.int32)
    return keys, np.ascontiguousarray(counts)


def run(argv: list[str]):
    """Build the SEC (systematic error correction) database of a cohort: per locus the summed allele counts"""
    args = get_parser().parse_args(argv[1:])
    from ..engine import Engine            # fails loudly if the library or the GPU is missing
    from ..io import vcf_native
    paths = list(args.inputs)
    if args.input_list:
        with open(args.input_list) as fh:
            paths += [ln.strip() for ln in fh if ln.strip()]
    if not paths:
        raise ValueError("sec_training: no cohort VCFs given (--inputs / --input_list)")
    names = vcf_native.read_fasta_names(args.reference_file)
    keys, counts, seen = [], [], []
    for p in paths:
        vt = vcf_native.read_vcf(p, names).table
        k, c = observations(vt)
        keys.append(k)
        counts.append(c)
        seen.append(np.unique(k))                        # a locus counts once per sample, however many records it has
        logger.info("%s: %d observations", p, k.size)
    with Engine(args.device) as eng:
        db_keys, expected = eng.sec_db_build(np.concatenate(keys), np.concatenate(counts, axis=0))
        if args.min_samples > 1:
            # how many SAMPLES saw each locus: the same kernel on (key, 1) rows
            sk, sn = eng.sec_db_build(np.concatenate(seen), np.ones((sum(s.size for s in seen), 2), np.int32))
            assert np.array_equal(sk, db_keys)
            keep = sn[:, 0] >= args.min_samples
            db_keys, expected = db_keys[keep], expected[keep]
    np.savez_compressed(args.output_file, keys=db_keys, expected=expected, contigs=np.array(names, dtype=object).astype(str),
                        n_samples=np.int64(len(paths)), k=np.int64(K_COUNTS))
    logger.info("%d loci from %d samples -> %s", db_keys.size, len(paths), args.output_file)
    return 0


def load_db(path: str, contig_names: list) -> tuple:
    """(keys, expected) of a database file, its contig indices re-mapped onto `contig_names`."""
    z = np.load(path if path.endswith(".npz") else path + ".npz", allow_pickle=False)
    keys, expected = z["keys"].astype(np.uint64), z["expected"].astype(np.int32)
    theirs = [str(x) for x in z["contigs"]]
    if theirs != list(contig_names):
        index = {n: i for i, n in enumerate(contig_names)}
        remap = np.array([index.get(n, -1) for n in theirs], np.int64)
        c = remap[(keys >> np.uint64(32)).astype(np.int64)]
        ok = c >= 0
        keys = ((c[ok].astype(np.uint64) << np.uint64(32)) | (keys[ok] & np.uint64(0xFFFFFFFF))).astype(np.uint64)
        expected = expected[ok]
        order = np.argsort(keys, kind="stable")
        keys, expected = keys[order], np.ascontiguousarray(expected[order])
    return keys, expected


if __name__ == "__main__":
    run(sys.argv)
