"""pipelines/calibrate_bridging_snvs.py (SURVEY.md 8(a) a12 "+ caller loop"): host pieces on CPU, the tool end to end on
the GPU against oracle/bridging.py (itself pinned on the reference's is_homopolymer_snp executed in the build container,
tests/test_oracle_golden.py)."""
import gzip
import os

import numpy as np
import pytest

from variantcalling_amd import schema as S, synth
from variantcalling_amd.io import vcf as pv
from variantcalling_amd.pipelines import calibrate_bridging_snvs as cb


def _tumor_normal_vcf(path, cs, rng, contig_order=None):
    """A tumor VCF with a background (normal) sample folded into FORMAT as DeepVariant-somatic does: AD, DP, BG_AD, BG_DP."""
    vt = cs.variants
    lines = ["##fileformat=VCFv4.2", '##FILTER=<ID=RefCall,Description="x">', '##FORMAT=<ID=GT,Number=1,Type=String,Description="g">',
             '##FORMAT=<ID=AD,Number=R,Type=Integer,Description="a">', '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="d">',
             '##FORMAT=<ID=BG_AD,Number=R,Type=Integer,Description="b">', '##FORMAT=<ID=BG_DP,Number=1,Type=Integer,Description="b">']
    lines += [f"##contig=<ID={n}>" for n in cs.ref.names]
    lines.append("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ttumor")
    recs = []
    for i in range(vt.n):
        ref = S.decode_bases(vt.alleles[vt.ref_off[i]:vt.ref_off[i] + vt.ref_len[i]])
        alt = S.decode_bases(vt.alleles[vt.alt_off[i]:vt.alt_off[i] + vt.alt_len[i]])
        multi = rng.random() < 0.05
        alts = alt + (",T" if multi and alt != "T" else "")
        dp = int(rng.integers(0, 60)) if rng.random() > 0.02 else 0
        ad1 = int(rng.integers(0, dp + 1)) if dp else 0
        ad = f"{max(dp - ad1, 0)},{ad1}" + (",1" if "," in alts else "")
        bg_dp = int(rng.integers(0, 40))
        bg1 = int(rng.integers(0, 3)) if rng.random() < 0.8 else int(rng.integers(0, bg_dp + 1))
        flt = ["RefCall", "PASS", ".", "RefCall;LowQual"][int(rng.integers(0, 4))]
        qual = [f"{rng.random() * 40:.2f}", "3", "."][int(rng.choice(3, p=[0.8, 0.1, 0.1]))]
        recs.append(f"{cs.ref.names[vt.contig[i]]}\t{vt.pos[i]}\t.\t{ref}\t{alts}\t{qual}\t{flt}\tDP={dp}\tGT:AD:DP:BG_AD:BG_DP\t"
                    f"0/1:{ad}:{dp}:{max(bg_dp - bg1, 0)},{bg1}:{bg_dp}")
    text = "\n".join(lines + recs) + "\n"
    if path.endswith(".gz"):
        w = pv._BgzfWriter(path)
        w.write(text.encode())
        w.close()
    else:
        open(path, "w").write(text)
    return recs


def test_sample_fields_parse_like_the_reference_reads_them():
    recs = [b"chr1\t5\t.\tA\tG\t30\tRefCall\t.\tGT:AD:DP:BG_AD:BG_DP\t0/1:10,7:17:20,1:21",
            b"chr1\t6\t.\tA\tG,T\t30\tPASS\t.\tGT:AD:DP:BG_AD:BG_DP\t1/2:1,7,3:11:20,1,2:23",
            b"chr1\t7\t.\tA\tG\t.\tq1;PASS\t.\tGT:AD:DP\t0/1:.,.:0",
            b"chr1\t8\t.\tA\t.\t1\t.\t."]
    f = cb.sample_fields(recs)
    assert f["n_alts"].tolist() == [1, 2, 1, 0] and f["is_pass"].tolist() == [False, True, True, False]
    assert f["ad_alt_sum"].tolist() == [7, 10, 0, 0] and f["bg_ad_alt_sum"].tolist() == [1, 3, 0, 0]
    assert f["bg_dp"].tolist() == [21, 23, 0, 0] and f["has_bg"].tolist() == [True, True, False, False]


def test_python_tabix_index_equals_the_native_one(tmp_path):
    """io.vcf.tabix_index (what the tool calls after writing) is the pure-Python statement of the native codec's index
    writer: byte-equal .tbi files on the same BGZF VCF; unsorted / ungrouped files are refused; plain files rejected."""
    from variantcalling_amd.io import vcf_native as nv
    for n, seed in ((0, 1), (700, 2), (40_000, 3)):
        cs = synth.make_callset(max(n, 10), genome_len=5_000_000, n_contigs=3, seed=seed)
        vt = cs.variants if n else cs.variants.slice(0, 0)
        src, dst = str(tmp_path / "i.vcf.gz"), str(tmp_path / "o.vcf.gz")
        pv.write_vcf_from_table(src, vt, cs.ref.names)
        v = nv.read_vcf(src, cs.ref.names)
        res = S.FilterResult(np.zeros(v.table.n, np.float32), np.zeros(v.table.n, np.uint8), np.zeros(v.table.n, np.uint8))
        assert nv.write_filtered_vcf(dst, v, res) is True
        native = open(dst + ".tbi", "rb").read()
        os.remove(dst + ".tbi")
        assert pv.tabix_index(dst) is True
        assert open(dst + ".tbi", "rb").read() == native
    lines = gzip.open(dst, "rb").read().split(b"\n")
    hdr = [x for x in lines if x.startswith(b"#")]
    body = [x for x in lines if x and not x.startswith(b"#")]
    for bad in (body[::-1], body[:10] + body[-10:] + body[10:20]):       # unsorted; a contig appearing twice
        w = pv._BgzfWriter(str(tmp_path / "bad.vcf.gz"))
        w.write(b"\n".join(hdr + bad) + b"\n")
        w.close()
        assert pv.tabix_index(str(tmp_path / "bad.vcf.gz")) is False and not os.path.exists(str(tmp_path / "bad.vcf.gz.tbi"))
    open(tmp_path / "plain.vcf", "wb").write(b"\n".join(hdr + body) + b"\n")
    with pytest.raises(ValueError, match="not a BGZF file"):
        pv.tabix_index(str(tmp_path / "plain.vcf"))


@pytest.mark.gpu
@pytest.mark.parametrize("hmer,gz", [(2, True), (5, False)])
def test_tool_matches_the_oracle(tmp_path, hmer, gz):
    import edge_cases as E
    from conftest import real_chr1_reference
    from oracle import bridging as B
    ref = real_chr1_reference()
    vt = E.edge_table(ref, seed=20 + hmer, n_random=3000)
    cs = type("CS", (), {})()
    cs.variants, cs.ref = vt, ref
    rng = np.random.default_rng(hmer)
    src = str(tmp_path / "in.vcf.gz")
    recs = _tumor_normal_vcf(src, cs, rng)
    fa = str(tmp_path / "ref.fa")
    with open(fa, "w") as fh:
        for c, name in enumerate(ref.names):
            seq = S.decode_bases(ref.codes[ref.contig_off[c]:ref.contig_off[c + 1]])
            fh.write(f">{name}\n")
            for k in range(0, len(seq), 80):
                fh.write(seq[k:k + 80] + "\n")
    out = str(tmp_path / ("out.vcf.gz" if gz else "out.vcf"))
    rc = cb.run(["calibrate_bridging_snvs", "--vcf", src, "--reference", fa, "--output", out, "--min_query_hmer_size", str(hmer),
                 "--min_tumor_vaf", "0.15", "--min_normal_depth", "8"])
    assert rc == 0
    final = out if gz else out + ".gz"
    assert os.path.exists(final) and os.path.exists(final + ".tbi") and (gz or not os.path.exists(out))
    got = [x for x in gzip.open(final, "rb").read().decode().split("\n") if x and not x.startswith("#")]
    assert len(got) == len(recs)
    # expected verdicts: the oracle on the same table and sample columns
    v = pv.read_vcf(src, ref.names)
    fld = cb.sample_fields(v.records)
    o = v.order
    hm, ok = B.calibrate(v.table, ref, fld["is_pass"][o], fld["ad_alt_sum"][o], fld["bg_ad_alt_sum"][o], fld["bg_dp"][o],
                         min_query_hmer_size=hmer, min_tumor_vaf=0.15, min_normal_depth=8)
    ok &= fld["n_alts"][o] == 1
    exp = np.zeros(len(recs), bool)
    exp[o] = ok
    assert 1 <= exp.sum() < len(recs) // 2
    for j, (a, b) in enumerate(zip(recs, got)):
        fa_, fb = a.split("\t"), b.split("\t")
        if exp[j]:
            assert fb[5] == "20" and fb[6] == "PASS" and fa_[:5] == fb[:5] and fa_[7:] == fb[7:], j
        else:
            assert a == b, j
