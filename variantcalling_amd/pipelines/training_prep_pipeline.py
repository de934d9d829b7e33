"""training_prep_pipeline: label a raw callset for train_models_pipeline (host logic, no GPU).

The reference registers `ugbio_filtering.training_prep_pipeline.run` (ugvc/__main__.py:18,50; script setup.py:40); its
body and documentation are in the absent submodule.  What the tree still shows: the two ways a training set is
labelled (docs/train_models_pipeline.md:5-10 - exact labels from a comparison against a truth set, or approximate:
dbSNP => true positive, blacklist => false positive) and the fixture names of its tests
(test/resources/unit/filtering/test_training_prep/: `input.vcf.gz`, `vcfeval_output.vcf.gz`,
`blacklist_chr1_1_5000000.h5`, `expected_labels.h5`, `expected_result_calculate_labeled_vcf.h5`): a call VCF, the output
of `rtg vcfeval` on it, a blacklist, and label tables.  The flags below are therefore BUILDER-DEFINED around those facts:

  --call_vcf            the raw callset
  --vcfeval_output      optional: `rtg vcfeval --output-mode=combine|annotate` output; a record's INFO/CALL (TP / FP; TP
                        variants: CALL=TP, others FP / FP_CA) is its label
  --blacklist           optional: loci that are false positives (any format `--blacklist` of the other tools reads)
  --hcr                 optional BED: only calls inside it are labelled (the truth set's high-confidence region)
  --reference           FASTA (contig order)
  --output_prefix       writes PREFIX.h5: one concordance-shaped frame per contig (chrom, pos, ref, alleles, ..., classify,
                        classify_gt) that `train_models_pipeline --input_file PREFIX.h5` reads, plus the key `labels`
                        (chrom, pos, label)

Label precedence: vcfeval CALL where present, else blacklist => fp, else dbSNP id => tp, else unlabelled."""
from __future__ import annotations

import argparse
import logging
import sys

import numpy as np

from .. import schema as S
from ..io import bed, concordance, h5
from ..io import vcf_native as vcfio
from .train_models_pipeline import _inside_intervals

logger = logging.getLogger("ugvc")


def get_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="training_prep_pipeline.py", description="Label a callset for model training")
    ap.add_argument("--call_vcf", help="Raw callset VCF", type=str, required=True)
    ap.add_argument("--vcfeval_output", help="rtg vcfeval output VCF of the same calls (INFO/CALL = TP | FP)", type=str)
    ap.add_argument("--blacklist", help="blacklist file by which we decide variants as FP", type=str)
    ap.add_argument("--hcr", help="BED of the region in which calls are labelled", type=str)
    ap.add_argument("--reference", help="Reference genome", type=str, required=True)
    ap.add_argument("--output_prefix", help="Output prefix: PREFIX.h5", type=str, required=True)
    ap.add_argument("--verbosity", help="Verbosity: ERROR, WARNING, INFO, DEBUG", default="INFO")
    return ap


def vcfeval_labels(path: str, contig_names: list) -> dict:
    """{(contig index, pos, ref, first alt): 1 tp / 0 fp} from the INFO/CALL tags of a vcfeval output VCF."""
    from ..io import vcf as pyvcf
    idx = {n: i for i, n in enumerate(contig_names)}
    out = {}
    with pyvcf._open(path) as fh:
        for line in fh:
            if line.startswith(b"#") or not line.strip():
                continue
            f = line.rstrip(b"\r\n").split(b"\t")
            if len(f) < 8 or f[0].decode() not in idx:
                continue
            call = None
            for kv in f[7].split(b";"):
                if kv.startswith(b"CALL="):
                    call = kv[5:].decode()
            if call is None:
                continue                                  # a baseline-only record (BASE=FN): no call to label
            out[(idx[f[0].decode()], int(f[1]), f[3].decode().upper(), f[4].split(b",")[0].decode().upper())] = 1 if call == "TP" else 0
    return out


def label_calls(vcf, contig_names, eval_labels: dict | None, blacklist: np.ndarray | None, region: S.IntervalTrack | None) -> np.ndarray:
    vt = vcf.table
    label = np.full(vt.n, -1, np.int8)
    label[np.asarray(vcf.ids, bool)] = 1                   # dbSNP => TP
    if blacklist is not None and blacklist.size:
        k = vt.keys()
        i = np.minimum(np.searchsorted(blacklist, k), blacklist.size - 1)
        label[blacklist[i] == k] = 0                      # blacklist => FP
    if eval_labels:
        for r in range(vt.n):
            key = (int(vt.contig[r]), int(vt.pos[r]), S.decode_bases(vt.alleles[vt.ref_off[r]: vt.ref_off[r] + vt.ref_len[r]]),
                   S.decode_bases(vt.alleles[vt.alt_off[r]: vt.alt_off[r] + vt.alt_len[r]]))
            if key in eval_labels:
                label[r] = eval_labels[key]
    if region is not None:
        label[~_inside_intervals(region, vt.contig, vt.pos)] = -1
    return label


def run(argv: list[str]):
    """Label a callset for model training"""
    args = get_parser().parse_args(argv[1:])
    logger.setLevel(getattr(logging, str(args.verbosity).upper(), logging.INFO))
    ref_names = vcfio.read_fasta_names(args.reference)
    vcf = vcfio.read_vcf(args.call_vcf, ref_names)
    ev = vcfeval_labels(args.vcfeval_output, ref_names) if args.vcfeval_output else None
    bl = bed.read_blacklist(args.blacklist, ref_names) if args.blacklist else None
    region = vcfio.read_intervals(args.hcr, ref_names, merge=True) if args.hcr else None
    label = label_calls(vcf, ref_names, ev, bl, region)
    vt = vcf.table
    logger.info("%d calls: %d tp, %d fp, %d unlabelled", vt.n, int((label == 1).sum()), int((label == 0).sum()), int((label < 0).sum()))
    frames = {}
    for c in np.unique(vt.contig):
        rows = np.flatnonzero(vt.contig == c)
        frames[ref_names[int(c)]] = concordance.table_to_frame(vt.slice(int(rows[0]), int(rows[-1]) + 1), ref_names, label[rows])
    chrom = np.array(ref_names, dtype=object)[vt.contig]
    frames["labels"] = h5.Frame([("chrom", chrom), ("pos", vt.pos.astype(np.int64)), ("label", label.astype(np.int64))])
    h5.write_hdf(args.output_prefix + ".h5", frames)
    return 0


if __name__ == "__main__":
    run(sys.argv)
