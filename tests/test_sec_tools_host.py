"""The two SEC consumer tools' host logic (round 4; flags BUILDER-DEFINED, /root/reference/ugvc/__main__.py:19,44,56):
`assess_sec_concordance` end to end on the host (it needs no GPU: a join on (chrom, pos) and the reference's accuracy formulas),
`sec_validation`'s report rows.  The GPU end of sec_validation is tests/test_gpu_pipelines.py."""
import csv
import os

import numpy as np

from variantcalling_amd import evaluate as E, schema as S, synth
from variantcalling_amd.io import fasta, h5, vcf as pv
from variantcalling_amd.pipelines import assess_sec_concordance as asc, sec_validation as sv


def test_assess_counts_and_accuracy():
    cls = np.array(["tp"] * 6 + ["fp"] * 4 + ["fn"] * 2)
    sec = np.array([True, False, False, False, False, False, True, True, False, False, False, False])
    indel = np.array([False] * 5 + [True] + [False, True, False, False] + [False, True])
    hmer = np.array([0] * 5 + [3] + [0, 6, 0, 0] + [0, 0])
    rows = {r["group"]: r for r in asc.assess(cls, sec, indel, hmer)}
    a = rows["ALL"]
    assert (a["tp"], a["fp"], a["fn"], a["tp_tagged_sec"], a["fp_tagged_sec"]) == (6, 4, 2, 1, 2)
    assert a["precision"] == float(E.get_precision(4, 6)) and a["recall"] == float(E.get_recall(2, 6))
    # a tagged true call becomes a false negative, a tagged false call disappears
    assert a["precision_after_sec"] == float(E.get_precision(2, 5)) and a["recall_after_sec"] == float(E.get_recall(3, 5))
    assert a["f1_after_sec"] == float(E.get_f1(a["precision_after_sec"], a["recall_after_sec"]))
    assert rows["SNP"]["tp"] == 5 and rows["Indel"]["tp"] == 1 and rows["hmer Indel >4"]["fp_tagged_sec"] == 1
    assert rows["hmer Indel <=4"]["tp"] == 1 and rows["non-hmer Indel"]["fn"] == 1


def test_assess_tool_joins_the_frame_with_the_tagged_vcf(tmp_path):
    cs = synth.make_callset(3_000, genome_len=3_000_000, n_contigs=2, seed=3)
    vt = cs.variants
    fa = str(tmp_path / "ref.fa"); fasta.write_fasta(fa, cs.ref)
    rng = np.random.default_rng(1)
    truth = rng.random(vt.n) < 0.8                            # a call is true ...
    tagged = rng.random(vt.n) < np.where(truth, 0.02, 0.5)    # ... and the SEC tag hits mostly the false ones
    # the VCF as correct_systematic_errors leaves it: SEC in FILTER of the tagged calls
    res = S.FilterResult(np.zeros(vt.n, np.float32), np.zeros(vt.n, np.uint8), np.where(tagged, S.FLAG_SEC, 0).astype(np.uint8))
    raw = str(tmp_path / "raw.vcf"); pv.write_vcf_from_table(raw, vt, cs.ref.names)
    out = str(tmp_path / "sec.vcf.gz")
    pv.write_filtered_vcf(out, pv.read_vcf(raw, cs.ref.names), res)
    # the comparison frame of the same calls + 50 missed truth variants
    n_fn = 50
    chrom = np.array([cs.ref.names[c] for c in vt.contig] + [cs.ref.names[0]] * n_fn, dtype=object)
    pos = np.concatenate([vt.pos.astype(np.int64), np.arange(n_fn, dtype=np.int64) * 7 + 2_900_000])
    classify = np.array(["tp" if t else "fp" for t in truth] + ["fn"] * n_fn, dtype=object)
    indel = np.concatenate([vt.ref_len != vt.alt_len, np.zeros(n_fn, bool)])
    fr = h5.Frame([("chrom", chrom), ("pos", pos), ("classify", classify), ("indel", indel),
                   ("hmer_indel_length", np.zeros(pos.size, np.float64))])
    frame_path = str(tmp_path / "comp.h5")
    h5.write_hdf(frame_path, {"concordance": fr})
    prefix = str(tmp_path / "rep")
    assert asc.run(["assess_sec_concordance", "--input_file", out, "--concordance_h5_input", frame_path, "--reference_file", fa,
                    "--output_prefix", prefix, "--dataset_key", "concordance"]) == 0
    rows = {r["group"]: r for r in csv.DictReader(open(prefix + ".sec_concordance.csv"))}
    a = rows["ALL"]
    assert int(a["tp"]) == int(truth.sum()) and int(a["fp"]) == int((~truth).sum()) and int(a["fn"]) == n_fn
    assert int(a["tp_tagged_sec"]) == int((tagged & truth).sum()) and int(a["fp_tagged_sec"]) == int((tagged & ~truth).sum())
    assert float(a["precision_after_sec"]) > float(a["precision"]) and float(a["recall_after_sec"]) < float(a["recall"])
    assert int(rows["SNP"]["tp"]) + int(rows["Indel"]["tp"]) == int(a["tp"])


def test_sec_validation_rows():
    ratio = np.array([np.nan, 0.5, 0.01, 0.2, np.nan, 1.0])
    hit = np.array([False, True, False, True, False, True])
    r = sv.summarise("s", ratio, hit)
    assert (r["n_calls"], r["n_on_database"], r["n_sec"]) == (6, 4, 3)
    assert r["frac_sec_of_database_calls"] == 0.75 and r["ratio_q50"] == float(np.quantile([0.5, 0.01, 0.2, 1.0], 0.5))
    e = sv.summarise("empty", np.zeros(0), np.zeros(0, bool))
    assert e["n_calls"] == 0 and np.isnan(e["ratio_q50"])
