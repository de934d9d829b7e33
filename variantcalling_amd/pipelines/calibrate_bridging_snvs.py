"""calibrate_bridging_snvs: un-filter borderline SNVs that bridge a long homopolymer; the per-record test on an MI355X.

Drop-in for /root/reference/ugvc/pipelines/vcfbed/calibrate_bridging_snvs.py (the one tool of the
reference whose per-variant function is fully in-tree): flags :69-88, flow :91-130 - for every record
`is_homopolymer_snp` (:9-66: bi-allelic SNV, not PASS, QUAL >= --min_initial_qual, the ALT base continuing a reference
run of >= --min_query_hmer_size that is no tandem repeat) and the tumor / normal VAF gate from AD, DP, BG_AD, BG_DP of
the first sample (:114-122); records that pass get FILTER = PASS and QUAL = --set_qual (:125-126); the output VCF is
tabix-indexed (:130).  The record test runs as one launch (`ugvc_bridging_snvs`; its checker oracle/bridging.py is
pinned on the reference function executed in the build container); host I/O is Python, per record, like the
reference's pysam loop - the native codec only carries the columns of the filtering path."""
from __future__ import annotations

import argparse
import logging
import os
import shutil
import sys

import numpy as np

from ..io import vcf as pv

logger = logging.getLogger("ugvc")


def init_parser():
    parser = argparse.ArgumentParser(prog="calibrate_bridging_snvs", description=run.__doc__)
    parser.add_argument("--vcf", required=True, help="Path to the VCF file")
    parser.add_argument("--reference", required=True, help="Path to the reference genome")
    parser.add_argument("--output", required=True, help="name of output vcf file")
    parser.add_argument("--min_query_hmer_size", default=5, type=int,
                        help="min size of the homopolymer in the query genome (with SNV alt allele) to be considered")
    parser.add_argument("--min_initial_qual", default=5, type=int, help="min quality of the initial SNV call")
    parser.add_argument("--min_tumor_vaf", default=0.2, type=float, help="min variant allele frequency in the tumor")
    parser.add_argument("--max_normal_vaf", default=0.1, type=float, help="max variant allele frequency in the normal")
    parser.add_argument("--min_normal_depth", default=10, type=int, help="min depth in the normal")
    parser.add_argument("--min_distance_from_edge", default=0, type=int, help="min distance from the edge of the homopolymer")
    parser.add_argument("--set_qual", default=20, type=int, help="set the quality of the SNV to this value")
    parser.add_argument("--device", help="GPU index (MI355X)", type=int, default=0)
    return parser


def _ints(val: bytes) -> list:
    return [0 if x in (b".", b"") else int(float(x)) for x in val.split(b",")]


def sample_fields(records: list) -> dict:
    """Per record, in file order: n_alts, PASS among the filters, sum(AD[1:]), sum(BG_AD[1:]), BG_DP of the first
    sample (calibrate_bridging_snvs.py:14-20,114-116).  A record the VAF gate would need but that lacks BG_AD / BG_DP
    raises, as the reference's `record.samples[0]["BG_DP"]` does."""
    n = len(records)
    out = dict(n_alts=np.zeros(n, np.int32), is_pass=np.zeros(n, bool), ad_alt_sum=np.zeros(n, np.int32),
               bg_ad_alt_sum=np.zeros(n, np.int32), bg_dp=np.zeros(n, np.int32), has_bg=np.zeros(n, bool))
    for j, line in enumerate(records):
        f = line.split(b"\t")
        out["n_alts"][j] = 0 if f[4] in (b".", b"") else f[4].count(b",") + 1
        out["is_pass"][j] = b"PASS" in f[6].split(b";")
        if len(f) > 9:
            got = dict(zip(f[8].split(b":"), f[9].split(b":")))
            if b"AD" in got:
                out["ad_alt_sum"][j] = sum(_ints(got[b"AD"])[1:])
            if b"BG_AD" in got and b"BG_DP" in got:
                out["bg_ad_alt_sum"][j] = sum(_ints(got[b"BG_AD"])[1:])
                out["bg_dp"][j] = _ints(got[b"BG_DP"])[0]
                out["has_bg"][j] = True
    return out


def run(argv):
    """
    Un-filter SNVs which generate a long homopolymer, have borderline quality
    and have a high VAF in the tumor and low VAF in the normal
    * DV often filters such true SNVs due to low confidence of the allele (SNV / deletion)
    """
    args = init_parser().parse_args(argv[1:])
    from ..engine import Engine            # fails loudly if the library or the GPU is missing
    from ..io import vcf_native
    ref = vcf_native.read_fasta(args.reference)
    vcf = pv.read_vcf(args.vcf, ref.names)
    vt = vcf.table
    fld = sample_fields(vcf.records)
    o = vcf.order                          # table row k <- record o[k]
    with Engine(args.device) as eng:
        eng.set_reference(ref)
        is_hm, ok = eng.bridging_snvs(vt, fld["is_pass"][o], fld["ad_alt_sum"][o], fld["bg_ad_alt_sum"][o], fld["bg_dp"][o],
                                      args.min_query_hmer_size, args.min_initial_qual, args.min_tumor_vaf,
                                      args.max_normal_vaf, args.min_normal_depth, args.min_distance_from_edge)
    bi = fld["n_alts"][o] == 1             # :16-17 one ALT only (the table carries the first ALT of every record)
    is_hm &= bi
    ok &= bi
    lacking = is_hm & ~fld["has_bg"][o]
    if lacking.any():
        k = int(np.flatnonzero(lacking)[0])
        raise KeyError(f"{args.vcf}: record {int(o[k]) + 1} ({ref.names[vt.contig[k]]}:{int(vt.pos[k])}) has no BG_AD / BG_DP "
                       "in its first sample")
    unfilter = np.zeros(vt.n, bool)
    unfilter[o] = ok                       # back to file order
    gz = args.output.endswith(".gz")
    out = pv._BgzfWriter(args.output) if gz else open(args.output, "wb")
    out.write(("\n".join(vcf.header) + "\n").encode())
    qual = str(args.set_qual).encode()
    for j, line in enumerate(vcf.records):
        if unfilter[j]:
            f = line.split(b"\t")
            f[5], f[6] = qual, b"PASS"     # :125-126 (adding PASS replaces whatever filters the record had)
            logger.info(line.decode(errors="replace"))
            line = b"\t".join(f)
        out.write(line + b"\n")
    out.close()
    # :130 pysam.tabix_index(output, preset="vcf"): a plain file is BGZF-compressed to output.gz (and removed) first
    target = args.output
    if not gz:
        target = args.output + ".gz"
        w = pv._BgzfWriter(target)
        with open(args.output, "rb") as src:
            shutil.copyfileobj(src, w)
        w.close()
        os.remove(args.output)
    if not pv.tabix_index(target):
        raise ValueError(f"{target}: records are not sorted by position within contiguous contigs; cannot index")
    logger.info("%d of %d records un-filtered (%d homopolymer-bridging SNVs)", int(ok.sum()), vt.n, int(is_hm.sum()))
    return 0


if __name__ == "__main__":
    run(sys.argv)
