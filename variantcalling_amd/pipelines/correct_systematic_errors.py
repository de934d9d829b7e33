"""correct_systematic_errors: mark the calls of a VCF that the cohort's systematic noise explains (SEC), on an MI355X.

Stands in for `ugbio_filtering.sec.correct_systematic_errors.run(argv)` (registered at /root/reference/ugvc/__main__.py:
19,56; script setup.py:44; "SEC ... still undocumented", README.md:12).  Body and flags are in the absent submodule: the
FLAGS BELOW ARE BUILDER-DEFINED.  The reference's own parts: the statistic - observed allele counts of a call against
the cohort's expected counts at that locus by `multinomial_likelihood_ratio` after `scale_contingency_table`
(/root/reference/ugvc/utils/stats_utils.py:12-70) - and the consequence: such a call carries "SEC" and the reports read
that as filter = "SEC" (/root/reference/ugvc/reports/report_utils.py:71-75,408-413).

VCF -> columns (native codec) -> ONE `ugvc_sec_apply` launch over the resident callset (the database is sorted by the
callset's own key, so the join is carried from tile to tile) -> the input records, in their order, with `SEC` added to
the FILTER of every hit (a PASS / "." is replaced, other filters are kept) and INFO/SEC_LR = the likelihood ratio."""
from __future__ import annotations

import argparse
import logging
import sys

import numpy as np

from ..io import vcf as pv

logger = logging.getLogger("ugvc")

SEC_HEADER = ['##FILTER=<ID=SEC,Description="Systematic error: the cohort\'s allele counts explain the call">',
              '##INFO=<ID=SEC_LR,Number=1,Type=Float,Description="Likelihood ratio observed vs cohort-expected allele counts">']


def get_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="correct_systematic_errors.py", description=run.__doc__)
    ap.add_argument("--input_file", help="Name of the input VCF file", type=str, required=True)
    ap.add_argument("--sec_db", help="SEC database written by sec_training (.npz)", type=str, required=True)
    ap.add_argument("--reference_file", help="Indexed reference FASTA file (contig names and order)", type=str, required=True)
    ap.add_argument("--output_file", help="Output VCF file", type=str, required=True)
    ap.add_argument("--min_ratio", help="a call is a systematic error when the likelihood ratio reaches this", type=float, default=0.05)
    ap.add_argument("--no_scaling", help="compare with the cohort's raw counts instead of scaling them to the call's depth",
                    action="store_true")
    ap.add_argument("--device", help="GPU index (MI355X)", type=int, default=0)
    return ap


def run(argv: list[str]):
    """Correct systematic errors: tag the calls the cohort's noise model explains"""
    args = get_parser().parse_args(argv[1:])
    from ..engine import Engine            # fails loudly if the library or the GPU is missing
    from ..io import vcf_native
    from .sec_training import load_db
    names = vcf_native.read_fasta_names(args.reference_file)
    db_keys, expected = load_db(args.sec_db, names)
    vcf = pv.read_vcf(args.input_file, names)
    vt = vcf.table
    with Engine(args.device) as eng:
        eng.set_contigs(names)                         # the join is on (contig, pos): no bases needed
        eng.set_sec_db(db_keys, expected)
        eng.upload_variants(vt)
        ratio, hit = eng.sec_apply(args.min_ratio, not args.no_scaling, mark=False)
    sec = np.zeros(vt.n, bool)
    lr = np.full(vt.n, np.nan)
    sec[vcf.order] = hit                               # back to file order
    lr[vcf.order] = ratio
    gz = args.output_file.endswith(".gz")
    out = pv._BgzfWriter(args.output_file) if gz else open(args.output_file, "wb")
    hdr = [h for h in vcf.header if not h.startswith("#CHROM")]
    for h in SEC_HEADER:
        if not any(x.startswith(h.split(",")[0]) for x in hdr):
            hdr.append(h)
    hdr += [h for h in vcf.header if h.startswith("#CHROM")]
    out.write(("\n".join(hdr) + "\n").encode())
    for j, line in enumerate(vcf.records):
        if lr[j] == lr[j]:                             # the locus is in the database: report the ratio
            f = line.split(b"\t")
            if sec[j]:
                old = [t for t in f[6].split(b";") if t not in (b"PASS", b".", b"", b"SEC")]
                f[6] = b";".join(old + [b"SEC"])
            info = [] if f[7] in (b".", b"") else [x for x in f[7].split(b";") if not x.startswith(b"SEC_LR=")]
            info.append(b"SEC_LR=" + np.format_float_positional(np.float32(lr[j]), unique=True, trim="0").encode())
            f[7] = b";".join(info)
            line = b"\t".join(f)
        out.write(line + b"\n")
    out.close()
    if gz:
        pv.tabix_index(args.output_file)
    logger.info("%d of %d calls tagged SEC (%d on database loci)", int(sec.sum()), vt.n, int((lr == lr).sum()))
    return 0


if __name__ == "__main__":
    run(sys.argv)
