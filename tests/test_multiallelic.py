"""Multi-allelic records, spanning deletions and a production-sized contig list through the host layer.
The per-ALT expansion rule is BUILDER-DEFINED (variantcalling_amd/io/multiallelic.py states it and why); these tests pin
the rule itself, on both VCF readers.  GPU: the same records through the CLI (tests/test_gpu_pipelines.py)."""
import gzip
import os

import numpy as np
import pytest

from conftest import ROOT
from variantcalling_amd import schema as S
from variantcalling_amd.io import multiallelic, vcf as pyvcf, vcf_native

HDR = ["##fileformat=VCFv4.2", "##contig=<ID=c1>", "##contig=<ID=c2>",
       "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ts1"]
RECS = [
    "c1\t100\t.\tA\tC\t50\t.\tSOR=1.0\tGT:AD:DP:GQ\t0/1:10,12:22:60",
    "c1\t200\t.\tA\tC,G,AT\t60\t.\tSOR=2.0\tGT:AD:DP:GQ\t1/2:3,9,11,4:27:70",          # three real alleles
    "c1\t300\t.\tAT\tA,*\t70\t.\tSOR=0.5\tGT:AD:DP:GQ\t0/1:8,9,5:22:80",                # deletion + spanning deletion
    "c1\t301\t.\tT\t*,G\t40\t.\tSOR=0.7\tGT:AD:DP:GQ\t1/2:6,7,9:22:50",                 # '*' first: the real allele takes the row
    "c1\t400\t.\tC\t*\t30\t.\tSOR=0.9\tGT:AD:DP:GQ\t0/1:5,6:11:40",                     # only a spanning deletion: placeholder row
    "c2\t50\t.\tG\t<DEL>,GA\t35\t.\tSOR=1.1\tGT:AD:DP:GQ\t0/2:9,0,7:16:45",             # symbolic allele skipped
    "c2\t60\t.\tG\tT\t20\t.\tSOR=1.2\tGT:AD:DP:GQ\t0/1:4,5:9:30",
]


def _write(tmp_path):
    p = str(tmp_path / "m.vcf")
    open(p, "w").write("\n".join(HDR + RECS) + "\n")
    return p


@pytest.mark.parametrize("reader", ["python", "native"])
def test_expand_makes_one_row_per_real_alt_and_collapse_folds_them(tmp_path, reader):
    p = _write(tmp_path)
    v = (pyvcf.read_vcf if reader == "python" else vcf_native.read_vcf)(p, ["c1", "c2"])
    assert v.n_alt.tolist() == [1, 3, 2, 2, 1, 2, 1]
    t, base = multiallelic.expand(v)
    assert base.tolist() == [0, 1, 1, 1, 2, 3, 4, 5, 6]
    alt = ["".join("NACGT"[c] for c in t.alleles[int(o): int(o) + int(n)]) for o, n in zip(t.alt_off, t.alt_len)]
    ref = ["".join("NACGT"[c] for c in t.alleles[int(o): int(o) + int(n)]) for o, n in zip(t.ref_off, t.ref_len)]
    assert alt == ["C", "C", "G", "AT", "A", "G", "N", "GA", "T"]            # '*' alone stays a placeholder (N)
    assert ref == ["A", "A", "A", "A", "AT", "T", "C", "G", "G"]
    assert t.ad_ref.tolist() == [10, 3, 3, 3, 8, 6, 5, 9, 4]
    assert t.ad_alt.tolist() == [12, 9, 11, 4, 9, 9, 6, 7, 5]                 # AD[j] of the row's own allele
    assert t.pos.tolist() == [100, 200, 200, 200, 300, 301, 400, 50, 60] and t.dp.tolist() == [22, 27, 27, 27, 22, 22, 11, 16, 9]
    # verdicts: any row PASS -> PASS; best score; flags OR-ed
    res = S.FilterResult(np.array([.9, .2, .8, .1, .3, .6, .5, .4, .7], np.float32),
                         np.array([0, 1, 0, 1, 1, 0, 1, 1, 0], np.uint8), np.array([0, 1, 0, 2, 0, 0, 8, 0, 0], np.uint8))
    got = multiallelic.collapse(res, base, v.table.n)
    assert got.filter.tolist() == [0, 0, 1, 0, 1, 1, 0]
    assert np.allclose(got.tree_score, [.9, .8, .3, .6, .5, .4, .7]) and got.flags.tolist() == [0, 3, 0, 0, 8, 0, 0]


def test_expand_is_the_identity_on_biallelic_callsets(tmp_path):
    p = str(tmp_path / "b.vcf")
    open(p, "w").write("\n".join(HDR + [RECS[0], RECS[6]]) + "\n")
    v = vcf_native.read_vcf(p, ["c1", "c2"])
    t, base = multiallelic.expand(v)
    assert t is v.table and base.tolist() == [0, 1]
    res = S.FilterResult(np.zeros(2, np.float32), np.zeros(2, np.uint8), np.zeros(2, np.uint8))
    assert multiallelic.collapse(res, base, 2) is res


def test_production_contig_list_is_indexed_whole(tmp_path):
    """3 366 contigs (the reference's production VCF header): calls on chrUn / decoy / HLA contigs beyond index 255 are
    read into the u16 contig column by both readers - no contig is dropped, none raises."""
    names = [ln.split("\t")[0] for ln in gzip.open(os.path.join(ROOT, "tests", "golden", "hg38_contigs.tsv.gz"), "rt") if not ln.startswith("#")]
    assert len(names) == 3366 and names[0] == "chr1" and names[-1].startswith("HLA-")
    picks = [0, 24, 255, 256, 300, 2000, 3365]
    lines = ["##fileformat=VCFv4.2"] + [f"##contig=<ID={n}>" for n in names] + ["#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ts1"]
    lines += [f"{names[c]}\t{10 + c}\t.\tA\tG\t30\t.\tSOR=1\tGT:AD:DP:GQ\t0/1:5,5:10:30" for c in picks]
    p = str(tmp_path / "wide.vcf")
    open(p, "w").write("\n".join(lines) + "\n")
    for reader in (pyvcf.read_vcf, vcf_native.read_vcf):
        v = reader(p, names)
        assert v.table.contig.dtype == np.uint16 and v.table.contig.tolist() == picks
        assert v.table.keys().tolist() == [(c << 32) | (10 + c) for c in picks]
