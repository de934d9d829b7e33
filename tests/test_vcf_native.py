"""Native VCF <-> SoA codec (include/ugvc_vcf.h) against the pure-Python host reference `io.vcf`
(SURVEY.md 8(f) rank 1): identical table columns, identical output bytes (compressed stream included,
both sides drive zlib level 6 over the same 65280-byte blocks), identical errors.  CPU only."""
import gzip
import os
import re
import subprocess
import zlib

import numpy as np
import pytest

from variantcalling_amd import schema as S, synth
from variantcalling_amd.io import vcf as pv
from variantcalling_amd.io import vcf_native as nv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COLS = ("contig", "pos", "ref_len", "alt_len", "ref_off", "alt_off", "alleles", "qual", "sor", "dp", "ad_ref",
        "ad_alt", "gq", "gt")


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not os.path.exists(nv.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "variantcalling_amd", "csrc_host")], check=True)


@pytest.fixture(autouse=True)
def _zlib_bytes():
    """The byte-for-byte comparisons of this file are against io/vcf.py, which drives zlib: the native codec is switched to
    zlib for them (its default since round 4 is libdeflate when the host has it: same text, same blocks, other bytes - the
    `deflate`-parametrised tests below cover that path by inflate-equality and by index queries)."""
    nv.set_deflate("zlib")
    yield
    nv.set_deflate("auto")


def _has_libdeflate():
    try:
        return nv.set_deflate("libdeflate") == "libdeflate"
    except RuntimeError:
        return False
    finally:
        nv.set_deflate("zlib")


def _same_file(a, b, mutect=False):
    for c in COLS:
        x, y = getattr(a.table, c), getattr(b.table, c)
        assert x.dtype == y.dtype and np.array_equal(x, y, equal_nan=x.dtype.kind == "f"), c
    assert np.array_equal(a.order, b.order) and np.array_equal(a.ids, b.ids)
    assert a.header == b.header and list(a.orig_filter) == list(b.orig_filter)
    if mutect:
        assert np.array_equal(a.tlod, b.tlod, equal_nan=True)
    else:
        assert a.tlod is None and b.tlod is None


def test_header_symbols_are_exported():
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "ugvc_vcf.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(ugvc_(?:vcf|fasta|intervals|bgzf|blob)_[a-z0-9_]+)\s*\(", text)))
    assert len(names) == 20                                      # (round 6: + ugvc_intervals_track, ugvc_bgzf_read, ugvc_blob_free)
    lib = nv.load_library()
    for n in names:
        assert hasattr(lib, n), n
    out = subprocess.run(["nm", "-D", "--defined-only", nv.LIB_PATH], capture_output=True, text=True).stdout
    assert set(names) <= set(re.findall(r" T (ugvc_\w+)", out))
    assert lib.ugvc_vcf_abi_version() == 2
    # host-only library: no HIP runtime dependency
    deps = subprocess.run(["ldd", nv.LIB_PATH], capture_output=True, text=True).stdout
    assert "amdhip" not in deps and "libz" in deps


@pytest.mark.parametrize("deflate", ["zlib", "libdeflate"])
@pytest.mark.parametrize("threads", [1, 3, 0])
def test_synthetic_callset_read_and_write_back(tmp_path, threads, deflate):
    if deflate == "libdeflate" and not _has_libdeflate():
        pytest.skip("libdeflate.so.0 is not on this host")
    nv.set_deflate(deflate)
    cs = synth.make_callset(30_000, genome_len=20_000_000, n_contigs=3, seed=5)
    vt = cs.variants
    ids = np.arange(vt.n) % 3 == 0
    rng = np.random.default_rng(2)
    res = S.FilterResult(rng.random(vt.n).astype(np.float32), rng.integers(0, 2, vt.n).astype(np.uint8),
                         rng.integers(0, 8, vt.n).astype(np.uint8))
    res.tree_score[:6] = [0.0, 1.0, 0.5, 1e-7, 0.333333343, 0.1]
    cg = rng.random(vt.n) < 0.01
    for name in ("in.vcf", "in.vcf.gz"):
        p = str(tmp_path / name)
        pv.write_vcf_from_table(p, vt, cs.ref.names, ids=ids)
        a = pv.read_vcf(p, cs.ref.names)
        b = nv.read_vcf(p, cs.ref.names, n_threads=threads)
        _same_file(a, b)
        for c in S.VariantTable.COLS:
            assert np.array_equal(getattr(b.table, c), getattr(vt, c)), c
        for out_name in ("o.vcf", "o.vcf.gz"):
            oa, ob = str(tmp_path / ("py_" + out_name)), str(tmp_path / ("nv_" + out_name))
            pv.write_filtered_vcf(oa, a, res, cg)
            nv.write_filtered_vcf(ob, b, res, cg, n_threads=threads)
            A, B = open(oa, "rb").read(), open(ob, "rb").read()
            if out_name.endswith(".gz"):
                assert gzip.decompress(A) == gzip.decompress(B)
                assert B.endswith(pv._BGZF_EOF) and B[12:14] == b"BC"
            if deflate == "zlib" or not out_name.endswith(".gz"):
                assert A == B, "output streams differ"
            else:
                # libdeflate: other compressed bytes, the same 65280-byte blocks; the file reads back (both readers, libdeflate
                # and zlib inflate) to the table and to the verdicts that were written
                off, sizes = 0, []
                while off < len(B):
                    bsize = int.from_bytes(B[off + 16: off + 18], "little") + 1
                    sizes.append(len(zlib.decompress(B[off + 18: off + bsize - 8], -15)))
                    off += bsize
                assert all(x == 65280 for x in sizes[:-2]) and sizes[-1] == 0
                for reader_deflate in ("libdeflate", "zlib"):
                    nv.set_deflate(reader_deflate)
                    c = nv.read_vcf(ob, cs.ref.names, n_threads=threads)
                    d = pv.read_vcf(oa, cs.ref.names)
                    _same_file(d, c)
                    c.close()
                nv.set_deflate(deflate)
                assert os.path.exists(ob + ".tbi")
        b.close()


EDGE = ("##fileformat=VCFv4.2\n##FILTER=<ID=LOW_SCORE,Description=\"old\">\n"
        "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ts1\ts2\n"
        "chr2\t50\t.\tA\tG,T\t.\t.\tDP=3;SOR=1.5;TLOD=4.5,7.25\tGT:AD:DP:GQ\t1|1:0,7,1:8:99\t0/0:1,0,0:1:3\n"
        "chr1\t10\trs5\tAT\tA\t33.5\tLowQual\t.\tGT:DP\t./.:.\t0/1:2\n"
        "\n"
        "chr1\t10\t.\tn\t<DEL>\t7\tPASS\tSOR=.\n"
        "chr1\t7\t.\tACGTN\tacgtn\t1e2\tq10;s50\tTREE_SCORE=0.1;HPOL_RUN;AC=2;;X\tGQ:GT:AD\t300:1/0:5\t-4:0|1|1:2,9\r\n"
        "##late header line\n"
        "chr2\t5\t.\tC\tT\tnan\t.\tSOR=0x10;TLOD=-3,-9\tGT:AD:DP:GQ\t1/1/1:3.9,4.2:11.7:-5\n"
        "chr1\t 12 \t.\tG\tA\t 4.5 \t.\tSOR= 2 \tDP:DP\t3:5\n")


@pytest.mark.parametrize("container", ["plain", "gzip", "bgzf"])
@pytest.mark.parametrize("sample", [0, 1])
def test_edge_records_match_the_python_reference(tmp_path, container, sample):
    p = str(tmp_path / ("e.vcf" if container == "plain" else "e.vcf.gz"))
    data = EDGE.encode()
    if container == "plain":
        open(p, "wb").write(data)
    elif container == "gzip":                       # ordinary multi-member gzip, not BGZF: serial inflate path
        open(p, "wb").write(gzip.compress(data[:150]) + gzip.compress(data[150:]))
    else:
        w = pv._BgzfWriter(p)
        w.write(data)
        w.close()
    names = ["chr1", "chr2"]
    for mutect in (False, True):
        a = pv.read_vcf(p, names, is_mutect=mutect, sample=sample)
        b = nv.read_vcf(p, names, is_mutect=mutect, sample=sample)
        _same_file(a, b, mutect)
    assert b.table.n == 6 and len(b.header) == 4
    res = S.FilterResult(np.array([.5, .25, 1, 0, .75, .125], np.float32), np.array([1, 0, 0, 1, 0, 1], np.uint8),
                         np.array([0, 1, 2, 3, 7, 0], np.uint8))
    cg = np.array([0, 0, 0, 0, 1, 1], bool)
    for out_name in ("o.vcf", "o.vcf.gz"):
        oa, ob = str(tmp_path / ("py_" + out_name)), str(tmp_path / ("nv_" + out_name))
        pv.write_filtered_vcf(oa, a, res, cg)
        nv.write_filtered_vcf(ob, b, res, cg)
        assert open(oa, "rb").read() == open(ob, "rb").read()
    txt = open(str(tmp_path / "nv_o.vcf")).read().splitlines()
    assert sum(x.startswith("##FILTER=<ID=LOW_SCORE") for x in txt) == 1          # existing header line kept, not duplicated
    rec = [x for x in txt if x.startswith("chr1\t7")][0].split("\t")
    assert rec[7].startswith("AC=2;;X;TREE_SCORE=") and "TREE_SCORE=0.1" not in rec[7]


def test_errors_match_the_python_reference(tmp_path):
    p = str(tmp_path / "bad.vcf")
    open(p, "w").write("#CHROM\nchr1\t5\t.\tA\tC\t1\t.\t.\nchr9\t5\t.\tA\tC\t1\t.\t.\nchr1\t5\t.\tA\n")
    for mod in (pv, nv):
        with pytest.raises(ValueError, match="contig 'chr9' is not in the reference"):
            mod.read_vcf(p, ["chr1"])
        with pytest.raises(ValueError, match=r"record 3 has 4 columns"):
            mod.read_vcf(p, ["chr1", "chr9"])
    with pytest.raises(ValueError, match="cannot open"):
        nv.read_vcf(str(tmp_path / "missing.vcf"), ["chr1"])
    g = str(tmp_path / "trunc.vcf.gz")
    w = pv._BgzfWriter(g)
    w.write(b"#CHROM\n" + b"chr1\t5\t.\tA\tC\t1\t.\t.\n" * 5000)
    w.close()
    raw = bytearray(open(g, "rb").read())
    raw[40] ^= 0xFF                                   # corrupt the first block's deflate payload
    open(g, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="corrupt"):
        nv.read_vcf(g, ["chr1"])
    open(str(tmp_path / "empty.vcf"), "w").write("##fileformat=VCFv4.2\n#CHROM\tPOS\n")
    v = nv.read_vcf(str(tmp_path / "empty.vcf"), ["chr1"])
    a = pv.read_vcf(str(tmp_path / "empty.vcf"), ["chr1"])
    assert v.table.n == 0 and a.table.n == 0 and v.header == a.header
    none = S.FilterResult(np.zeros(0, np.float32), np.zeros(0, np.uint8), np.zeros(0, np.uint8))
    pv.write_filtered_vcf(str(tmp_path / "e_py.vcf.gz"), a, none)
    nv.write_filtered_vcf(str(tmp_path / "e_nv.vcf.gz"), v, none)
    assert open(str(tmp_path / "e_py.vcf.gz"), "rb").read() == open(str(tmp_path / "e_nv.vcf.gz"), "rb").read()
    with pytest.raises(RuntimeError, match="do not match"):
        nv.write_filtered_vcf(str(tmp_path / "x.vcf"), v, S.FilterResult(np.zeros(2, np.float32), np.zeros(2, np.uint8),
                                                                      np.zeros(2, np.uint8)))


def test_tree_score_text_is_numpys_shortest_round_trip():
    rng = np.random.default_rng(11)
    xs = np.concatenate([rng.random(60_000).astype(np.float32),
                         (rng.random(20_000) * 1e-4).astype(np.float32),
                         np.array([0, 1, 0.5, 0.1, 1 / 3, 2 / 3, 1e-7, 1e-10, 3.4e38, 16777216, 0.975, 0.025, 1.17549435e-38],
                                  np.float32),
                         (rng.integers(0, 41, 20_000) / np.float64(40)).astype(np.float32)])
    for x in xs:
        want = np.format_float_positional(x, unique=True, trim="0")
        got = nv.format_f32(float(x))
        assert got == want, (float(x), got, want)
        assert np.float32(got) == x


def test_block_boundaries_follow_the_python_writer(tmp_path):
    """A stream that is an exact multiple of the BGZF block size ends without an empty data block."""
    line = b"chr1\t%d\t.\tA\tC\t1\t.\tK=" + b"x" * 30 + b"\n"
    p = str(tmp_path / "m.vcf")
    with open(p, "wb") as fh:
        fh.write(b"#CHROM\n")
        for k in range(20_000):
            fh.write(line % (k + 1))
    a, b = pv.read_vcf(p, ["chr1"]), nv.read_vcf(p, ["chr1"])
    n = a.table.n
    res = S.FilterResult(np.full(n, 0.5, np.float32), np.zeros(n, np.uint8), np.zeros(n, np.uint8))
    pv.write_filtered_vcf(str(tmp_path / "a.vcf.gz"), a, res)
    nv.write_filtered_vcf(str(tmp_path / "b.vcf.gz"), b, res)
    A = open(str(tmp_path / "a.vcf.gz"), "rb").read()
    assert A == open(str(tmp_path / "b.vcf.gz"), "rb").read()
    # every data block but the last inflates to exactly 65280 bytes
    off, sizes = 0, []
    while off < len(A):
        bsize = int.from_bytes(A[off + 16: off + 18], "little") + 1
        sizes.append(len(zlib.decompress(A[off + 18: off + bsize - 8], -15)))
        off += bsize
    assert sizes[-1] == 0 and all(s == 65280 for s in sizes[:-2]) and 0 < sizes[-2] <= 65280


@pytest.mark.parametrize("batch", [1, 7, 1000, 19_999, 20_000])
def test_writer_batch_seams(tmp_path, monkeypatch, batch):
    """The writer formats records in batches (2 M records; UGVC_VCF_WRITE_BATCH for this test) and carries the unfinished
    BGZF block from one batch into the next: file and index are the same bytes for any batch size, gz and plain."""
    line = b"chr1\t%d\t.\tA\tC\t1\t.\tK=" + b"x" * 30 + b"\n"
    p = str(tmp_path / "m.vcf")
    with open(p, "wb") as fh:
        fh.write(b"##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
        for k in range(20_000):
            fh.write(line % (k + 1))
    b = nv.read_vcf(p, ["chr1"])
    n = b.table.n
    rng = np.random.default_rng(3)
    res = S.FilterResult(rng.random(n).astype(np.float32), (rng.random(n) < 0.5).astype(np.uint8), rng.integers(0, 8, n).astype(np.uint8))
    monkeypatch.delenv("UGVC_VCF_WRITE_BATCH", raising=False)
    for ext in (".vcf.gz", ".vcf"):
        nv.write_filtered_vcf(str(tmp_path / ("whole" + ext)), b, res, n_threads=4)
    monkeypatch.setenv("UGVC_VCF_WRITE_BATCH", str(batch))
    for ext in (".vcf.gz", ".vcf"):
        nv.write_filtered_vcf(str(tmp_path / ("cut" + ext)), b, res, n_threads=4)
        assert open(str(tmp_path / ("cut" + ext)), "rb").read() == open(str(tmp_path / ("whole" + ext)), "rb").read()
    assert open(str(tmp_path / "cut.vcf.gz.tbi"), "rb").read() == open(str(tmp_path / "whole.vcf.gz.tbi"), "rb").read()


def test_fasta_reader_matches_the_python_reference(tmp_path):
    from variantcalling_amd.io import fasta
    text = (">chr1 first record description\nACGTNacgtn\nRYKM\n\n>chr2\r\nAC\r\nGT\r\n>empty\n>chrM\tmito\nTTTT")
    for name, data in (("a.fa", text.encode()), ("a.fa.gz", gzip.compress(text.encode()))):
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        a, b = fasta.read_fasta(p), nv.read_fasta(p)
        assert a.names == b.names == ["chr1", "chr2", "empty", "chrM"]
        assert np.array_equal(a.codes, b.codes) and np.array_equal(a.contig_off, b.contig_off)
        assert b.contig_off.tolist() == [0, 14, 18, 18, 22] and b.codes[:10].tolist() == [1, 2, 3, 4, 0, 1, 2, 3, 4, 0]
        sub_a, sub_b = fasta.read_fasta(p, contigs=["chrM", "chr2"]), nv.read_fasta(p, contigs=["chrM", "chr2"])
        assert sub_a.names == sub_b.names == ["chr2", "chrM"] and np.array_equal(sub_a.codes, sub_b.codes)
        assert np.array_equal(sub_a.contig_off, sub_b.contig_off)
    cs = synth.make_callset(2000, genome_len=3_000_000, n_contigs=3, seed=2)
    for name in ("g.fa", "g.fa.gz"):
        p = str(tmp_path / name)
        fasta.write_fasta(p, cs.ref)
        for th in (1, 0):
            b = nv.read_fasta(p, n_threads=th)
            assert b.names == cs.ref.names and np.array_equal(b.codes, cs.ref.codes) and np.array_equal(b.contig_off, cs.ref.contig_off)
    bad = str(tmp_path / "bad.fa")
    open(bad, "w").write("ACGT\n")
    for mod in (fasta, nv):
        with pytest.raises(ValueError, match="not a FASTA"):
            mod.read_fasta(bad)


def test_interval_reader_matches_the_python_reference(tmp_path):
    from variantcalling_amd.io import bed
    names = ["chr1", "chr2"]
    files = {
        "t.bed": "track name=x\nbrowser position\n#c\nchr1\t10\t20\textra\nchr2\t5\t9\nchrUn\t1\t2\nchr1\t15\t30\nchr1\t30\t40\n\nchr1\t100\t100\n",
        "s.bed": "chr1 10 20\nchr2   7\t9\n",
        "p.interval_list": "@HD\tVN:1.6\n@SQ\tSN:chr1\nchr1\t11\t20\t+\tx\nchr2\t6\t9\t+\ty\r\n",
        "h.bed": "chr1\t1\t5\n@late header turns one-based on\nchr1\t11\t20\n",
    }
    for fn, txt in files.items():
        for gz in (False, True):
            p = str(tmp_path / (fn + (".gz" if gz else "")))
            open(p, "wb").write(gzip.compress(txt.encode()) if gz else txt.encode())
            for merge in (True, False):
                a, b = bed.read_intervals(p, names, merge=merge), nv.read_intervals(p, names, merge=merge)
                assert np.array_equal(a.starts, b.starts) and np.array_equal(a.ends, b.ends), (fn, gz, merge)
                assert np.array_equal(a.contig_ptr, b.contig_ptr) and a.name == b.name
    t = nv.read_intervals(str(tmp_path / "t.bed"), names)
    assert t.starts.tolist() == [10, 5] and t.ends.tolist() == [40, 9] and t.contig_ptr.tolist() == [0, 1, 2]
    assert nv.read_intervals(str(tmp_path / "p.interval_list"), names).starts.tolist() == [10, 5]
    cs = synth.make_callset(3000, genome_len=3_000_000, n_contigs=3, seed=4)
    p = str(tmp_path / "big.bed")
    bed.write_bed(p, cs.tracks[2], cs.ref.names)
    for th in (1, 0):
        y = nv.read_intervals(p, cs.ref.names, n_threads=th)
        assert np.array_equal(y.starts, cs.tracks[2].starts) and np.array_equal(y.ends, cs.tracks[2].ends)
    badp = str(tmp_path / "bad.bed")
    open(badp, "w").write("chr1\t10\n")
    with pytest.raises(ValueError, match="fewer than 3"):
        nv.read_intervals(badp, names)
    open(badp, "w").write("chr1\tx\t20\n")
    with pytest.raises(ValueError, match="not integers"):
        nv.read_intervals(badp, names)


def test_native_track_equals_the_numpy_statement_on_random_rows(tmp_path):
    """ugvc_intervals_track (round 6: sort check / stable sort, empty rows dropped, merge by the running maximum of the ends,
    row range per contig - in C++) against bed.track_from_arrays on the SAME tokenised rows: sorted, shuffled, nested, book-ended,
    duplicated, zero-length and reversed intervals, contigs without rows, both merge modes."""
    names = [f"chr{k}" for k in range(1, 8)]
    rng = np.random.default_rng(11)
    for trial in range(12):
        n = int(rng.integers(0, 4000))
        c = rng.integers(0, 6, n)                                 # (chr7 never has rows: an empty contig at the end)
        s = rng.integers(0, 5000 if trial % 2 else 200_000, n)    # dense trials: plenty of overlap and nesting
        ln = rng.integers(-2, 60, n)                              # end <= start for some rows
        e = s + ln
        if trial % 3 == 0:                                        # a sorted file, as a rule
            o = np.lexsort((e, s, c))
            c, s, e = c[o], s[o], e[o]
        if trial == 5:                                            # book-ended rows
            s = np.arange(n) * 10
            e = s + 10
            c = np.zeros(n, np.int64)
        p = str(tmp_path / f"r{trial}.bed")
        with open(p, "w") as fh:
            for k in range(n):
                fh.write(f"{names[int(c[k])]}\t{int(s[k])}\t{int(e[k])}\n")
        for merge in (True, False):
            a = nv.read_intervals(p, names, merge=merge, native_track=False)
            b = nv.read_intervals(p, names, merge=merge)
            assert np.array_equal(a.starts, b.starts) and np.array_equal(a.ends, b.ends), (trial, merge)
            assert np.array_equal(a.contig_ptr, b.contig_ptr) and a.name == b.name
            assert b.starts.dtype == np.int32 and b.contig_ptr.dtype == np.int32 and b.contig_ptr.size == len(names) + 1


def _random_vcf(rng, n):
    nums = ["0", "1", "7", "42", "1e2", "1E-3", ".5", "5.", "+3", "-2", "nan", "NaN", "inf", "-Inf", "Infinity", ".", "", "abc",
            " 12", "12 ", "1_0", "0x10", "1e400", "-1e400", "1,2", "00012", "3.99", "-0.0", "2147483648", "-2147483649", "1e-320",
            # (round 4: plain decimals take the integer / power-of-ten path - fifteen digits at most - everything else strtod)
            "123456789012345", "1234567890123456", "0.000000000000001", "999999999999999.", "4.35", "0.1", "1.005", "-.5", "+.5",
            "1..2", "1.2.3", "-", "+", "-.", "0.30000000000000004", "72057594037927.93", "9007199254740993", "1.0000000000000002"]
    gts = ["0/1", "1/1", "1|1", "0|0", "./.", "1", "0", "1/0", "1/1/1", "1|0|1", ".", "", "2/1", "1/2"]
    chroms = ["chr1", "chr2", "chr3", "chrUn"]
    lines = ["##fileformat=VCFv4.2", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ts1\ts2"]
    for _ in range(n):
        c = chroms[int(rng.integers(0, 3))]
        pos = str(int(rng.integers(1, 5000)))
        vid = rng.choice([".", "rs1", ""])
        ref = "".join(rng.choice(list("ACGTNacgtn*"), size=int(rng.integers(1, 5))))
        alt = ",".join("".join(rng.choice(list("ACGTN<>*"), size=int(rng.integers(1, 4)))) for _ in range(int(rng.integers(1, 3))))
        qual = rng.choice(nums)
        flt = rng.choice([".", "PASS", "LowQual", "a;b", ""])
        info = ";".join(rng.choice(["DP=3", "SOR=" + rng.choice(nums), "TLOD=" + ",".join(rng.choice(nums, size=int(rng.integers(1, 4)))),
                                    "FLAG", "", "TREE_SCORE=0.5", "HPOL_RUN", "SOR", "TLOD="], size=int(rng.integers(0, 5))))
        if rng.random() < 0.1:
            info = "."
        fields = [c, pos, vid, ref, alt, qual, flt, info]
        if rng.random() < 0.85:
            keys = list(rng.choice(["GT", "AD", "DP", "GQ", "PL", "XX"], size=int(rng.integers(1, 6))))
            def sample():
                vals = []
                for k in keys[: int(rng.integers(1, len(keys) + 1))]:
                    if k == "GT": vals.append(rng.choice(gts))
                    elif k == "AD": vals.append(",".join(rng.choice(nums, size=int(rng.integers(1, 4)))))
                    else: vals.append(rng.choice(nums))
                return ":".join(vals)
            fields += [":".join(keys)] + [sample() for _ in range(int(rng.integers(1, 3)))]
        lines.append("\t".join(fields))
    return ("\n".join(lines) + "\n").encode()


@pytest.mark.parametrize("seed", list(range(8)))
def test_random_records_parse_like_the_python_reference(tmp_path, seed):
    """Token soup: numbers python's float() takes or refuses, odd genotypes, missing and extra fields, on both
    sample columns and in mutect mode.  Rows where the Python reference itself raises are dropped first."""
    rng = np.random.default_rng(500 + seed)
    data = _random_vcf(rng, 400)
    names = ["chr1", "chr2", "chr3"]
    p = str(tmp_path / "r.vcf")
    # keep only the records the reference can read (int(float('nan')) and int32 overflow raise in numpy / python)
    head, body = data.split(b"\n")[:2], [ln for ln in data.split(b"\n")[2:] if ln]
    good = []
    for ln in body:
        open(p, "wb").write(b"\n".join(head + [ln]) + b"\n")
        try:
            for sample in (0, 1):
                pv.read_vcf(p, names, sample=sample)
            good.append(ln)
        except (ValueError, OverflowError):
            pass
    assert len(good) > 150
    open(p, "wb").write(b"\n".join(head + good) + b"\n")
    for sample in (0, 1):
        for mutect in (False, True):
            a = pv.read_vcf(p, names, is_mutect=mutect, sample=sample)
            b = nv.read_vcf(p, names, is_mutect=mutect, sample=sample)
            _same_file(a, b, mutect)
    n = a.table.n
    res = S.FilterResult(rng.random(n).astype(np.float32), rng.integers(0, 2, n).astype(np.uint8), rng.integers(0, 4, n).astype(np.uint8))
    pv.write_filtered_vcf(str(tmp_path / "a.vcf"), a, res)
    nv.write_filtered_vcf(str(tmp_path / "b.vcf"), b, res)
    assert open(str(tmp_path / "a.vcf"), "rb").read() == open(str(tmp_path / "b.vcf"), "rb").read()


# ------------------------------------------------------------------ tabix index of the written file
def _read_tbi(path):
    import struct
    data = gzip.open(path, "rb").read()
    assert data[:4] == b"TBI\x01"
    n_ref, fmt, col_seq, col_beg, col_end, meta, skip, l_nm = struct.unpack_from("<8i", data, 4)
    names = data[36:36 + l_nm].split(b"\0")[:-1]
    off = 36 + l_nm
    refs = []
    for _ in range(n_ref):
        n_bin, = struct.unpack_from("<i", data, off); off += 4
        bins = {}
        for _ in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", data, off); off += 8
            bins[b] = [struct.unpack_from("<QQ", data, off + 16 * i) for i in range(n_chunk)]
            off += 16 * n_chunk
        n_intv, = struct.unpack_from("<i", data, off); off += 4
        lin = list(struct.unpack_from(f"<{n_intv}Q", data, off)); off += 8 * n_intv
        refs.append((bins, lin))
    n_no_coor, = struct.unpack_from("<Q", data, off); off += 8
    assert off == len(data) and n_no_coor == 0
    return [n.decode() for n in names], refs, (fmt, col_seq, col_beg, col_end, meta, skip)


def _reg2bins(beg, end):
    end -= 1
    out = [0]
    for shift, first in ((26, 1), (23, 9), (20, 73), (17, 585), (14, 4681)):
        out += list(range(first + (beg >> shift), first + (end >> shift) + 1))
    return out


class _Bgzf:
    """Random access by virtual offset over a BGZF file held in memory."""
    def __init__(self, raw):
        self.blocks, off = {}, 0
        while off < len(raw):
            bsize = int.from_bytes(raw[off + 16: off + 18], "little") + 1
            self.blocks[off] = (zlib.decompress(raw[off + 18: off + bsize - 8], -15), bsize)
            off += bsize

    def lines(self, v_beg, v_end):
        co, uo = v_beg >> 16, v_beg & 0xFFFF
        while (co << 16 | uo) < v_end and co in self.blocks:
            buf = b""
            while True:
                data, bsize = self.blocks[co]
                nl = data.find(b"\n", uo)
                if nl >= 0:
                    buf += data[uo:nl]; uo = nl + 1
                    if uo == len(data):
                        co, uo = co + bsize, 0
                    break
                buf += data[uo:]; co, uo = co + bsize, 0
                if co not in self.blocks:
                    break
            yield buf


def test_tabix_index_answers_region_queries(tmp_path):
    """The .tbi beside the written .vcf.gz (tabix specification: binning + linear index over BGZF virtual offsets)
    returns exactly the records a scan of the file finds, for random regions, long REF alleles, INFO/END records and
    windows without records; unsorted input gets no index."""
    rng = np.random.default_rng(21)
    names = ["chr1", "chr2", "chrM"]
    lens = [3_000_000, 900_000, 16_000]
    lines = ["##fileformat=VCFv4.2", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ts1"]
    recs = []
    for c, L in zip(names, lens):
        for p in np.sort(rng.integers(1, L, 6000 if c != "chrM" else 300)):
            kind = rng.random()
            if kind < 0.9:
                ref, alt, info, end = "A", "C", "SOR=1.0", int(p)
            elif kind < 0.97:
                k = int(rng.choice([2, 40, 3000, 20_000]))
                ref, alt, info, end = "A" * k, "A", "SOR=0.5", int(p) - 1 + k
            else:
                e = int(p) + int(rng.choice([10, 50_000, 400_000]))
                ref, alt, info, end = "N", "<DEL>", f"SVTYPE=DEL;END={e};SOR=2", e
            lines.append(f"{c}\t{p}\t.\t{ref}\t{alt}\t50\t.\t{info}\tGT:AD:DP:GQ\t0/1:5,5:10:50")
            recs.append((c, int(p) - 1, end))
    p_in = str(tmp_path / "in.vcf")
    open(p_in, "w").write("\n".join(lines) + "\n")
    b = nv.read_vcf(p_in, names)
    n = b.table.n
    res = S.FilterResult(rng.random(n).astype(np.float32), rng.integers(0, 2, n).astype(np.uint8), np.zeros(n, np.uint8))
    out = str(tmp_path / "out.vcf.gz")
    assert nv.write_filtered_vcf(out, b, res) is True
    tbi_names, refs, cols = _read_tbi(out + ".tbi")
    assert tbi_names == names and cols == (2, 1, 2, 0, ord("#"), 0)
    raw = open(out, "rb").read()
    bg = _Bgzf(raw)
    all_lines = [ln for ln in gzip.decompress(raw).split(b"\n") if ln and not ln.startswith(b"#")]
    assert len(all_lines) == n

    def span(line):
        f = line.split(b"\t")
        beg = int(f[1]) - 1
        end = beg + len(f[3])
        m = re.search(rb"(?:^|;)END=(\d+)", f[7])
        if m and int(m.group(1)) > beg:
            end = int(m.group(1))
        return f[0].decode(), beg, end

    spans = [span(ln) for ln in all_lines]
    for _ in range(300):
        c = int(rng.integers(0, 3))
        qb = int(rng.integers(0, lens[c]))
        qe = qb + int(rng.choice([1, 100, 16_384, 20_000, 500_000]))
        bins, lin = refs[c]
        w = qb >> 14
        min_off = lin[w] if w < len(lin) else (lin[-1] if lin else 0)
        got = set()
        for bn in _reg2bins(qb, qe):
            for cb, ce in bins.get(bn, []):
                if ce <= min_off:
                    continue
                for ln in bg.lines(cb, ce):
                    cc, sb, se = span(ln)
                    if cc == names[c] and sb < qe and se > qb:
                        got.add(ln)
        want = {ln for ln, (cc, sb, se) in zip(all_lines, spans) if cc == names[c] and sb < qe and se > qb}
        assert got == want, (names[c], qb, qe, len(got), len(want))
    # every chunk starts on a record boundary of its contig and the linear index never decreases
    for c, (bins, lin) in enumerate(refs):
        assert all(x <= y for x, y in zip(lin, lin[1:]))
        for chunks in bins.values():
            for cb, ce in chunks:
                first = next(bg.lines(cb, ce))
                assert first.split(b"\t")[0].decode() == names[c]
    # unsorted input: the file is written, the index is not
    open(p_in, "w").write("\n".join(lines[:2] + lines[2:][::-1]) + "\n")
    b2 = nv.read_vcf(p_in, names)
    out2 = str(tmp_path / "out2.vcf.gz")
    assert nv.write_filtered_vcf(out2, b2, res) is False and not os.path.exists(out2 + ".tbi")
    assert len(gzip.open(out2, "rb").read().splitlines()) == n + 7
    # plain-text output: nothing to index
    assert nv.write_filtered_vcf(str(tmp_path / "out3.vcf"), b, res) is False


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_corrupted_inputs_never_crash(tmp_path, seed):
    """Byte-mutated inputs (flips, truncations, dropped / duplicated spans, spliced noise) in every container the native
    library reads - and in the HDF5 reader - end in a Python exception or a parsed result, never in a dead process."""
    import sys
    from variantcalling_amd.io import bed as pbed, h5
    cs = synth.make_callset(400, genome_len=60_000, n_contigs=3, seed=seed)
    d = tmp_path
    pv.write_vcf_from_table(str(d / "a.vcf"), cs.variants, cs.ref.names)
    pv.write_vcf_from_table(str(d / "b.vcf.gz"), cs.variants, cs.ref.names)
    with open(d / "r.fa", "w") as fh:
        for c, name in enumerate(cs.ref.names):
            fh.write(f">{name} extra\n")
            seq = S.decode_bases(cs.ref.codes[cs.ref.contig_off[c]:cs.ref.contig_off[c + 1]])
            for k in range(0, len(seq), 60):
                fh.write(seq[k:k + 60] + "\n")
    with open(d / "r.fa", "rb") as src, gzip.open(d / "r2.fa.gz", "wb") as dst:
        dst.write(src.read())
    pbed.write_bed(str(d / "t.bed"), cs.tracks[0], cs.ref.names)
    h5.write_hdf(str(d / "f.h5"), {"k": h5.Frame([("chrom", np.array(["chr1"] * 30, dtype=object)), ("pos", np.arange(30)),
                                                    ("q", np.arange(30.0))], index=[np.array(["chr1"] * 30, dtype=object), np.arange(30)],
                                                   index_names=["chrom", "pos"])})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_corrupt_worker.py"), str(d), str(seed), "240"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"worker died with {r.returncode}:\n{r.stderr[-1500:]}"
    m = re.search(r"done ok=(\d+) err=(\d+)", r.stdout)
    assert m and int(m.group(1)) + int(m.group(2)) >= 240 and int(m.group(2)) > 20 and int(m.group(1)) > 20, r.stdout[-300:]


REF_TREE = "/root/reference/test/resources"


@pytest.mark.skipif(not os.path.exists(REF_TREE), reason="reference tree not present")
def test_real_fixtures_of_the_reference_tree(tmp_path):
    """The two real host-format files the reference tree carries for this path: a bgzipped FASTA with its .fai
    (unit/filtering/test_spandel/ref_fragment.fa.gz) and a production VCF header of 3 402 lines, 3 366 of them contigs
    (unit/vcfbed/test_vcftools/header.txt - the field dictionary SURVEY.md 8(c) names)."""
    fa = f"{REF_TREE}/unit/filtering/test_spandel/ref_fragment.fa.gz"
    name, length, *_ = open(fa + ".fai").read().split("\t")
    ref = nv.read_fasta(fa)
    assert ref.names == [name] and ref.contig_len(0) == int(length) == 660000
    seq = b"".join(gzip.open(fa, "rb").read().split(b"\n")[1:])
    assert np.array_equal(ref.codes, S._ASCII_TO_CODE[np.frombuffer(seq, np.uint8)]) and int((ref.codes == 0).sum()) == 150000     # a 150 kb N run
    from variantcalling_amd.io import fasta as pfa
    pref = pfa.read_fasta(fa)
    assert pref.names == ref.names and np.array_equal(pref.codes, ref.codes)

    hdr = open(f"{REF_TREE}/unit/vcfbed/test_vcftools/header.txt").read().rstrip("\n").split("\n")
    contigs = [h.split("ID=")[1].split(",")[0] for h in hdr if h.startswith("##contig")]
    assert len(hdr) == 3402 and len(contigs) == 3366 and hdr[-1].startswith("#CHROM") and contigs[0] == "chr1"
    names = contigs[:200]                                          # the contig column is u8
    rng = np.random.default_rng(2)
    cidx = np.sort(rng.integers(0, 200, 500))                      # records grouped by contig, positions rising
    body = []
    for k, c in enumerate(names[i] for i in cidx):
        body.append(f"{c}\t{1000 + 7 * k}\t.\tA\tG\t{30 + k % 9}.5\tRefCall\tSOR=1.25;X_CSS=non-skip;VARIANT_TYPE=SNP\tGT:AD:DP:GQ:PL:VAF\t"
                    f"0/1:10,{k % 13}:{10 + k % 13}:{k % 60}:30,0,40:0.5")
    src = str(tmp_path / "real_header.vcf.gz")
    w = pv._BgzfWriter(src)
    w.write(("\n".join(hdr + body) + "\n").encode())
    w.close()
    a, b = nv.read_vcf(src, names), pv.read_vcf(src, names)
    _same_file(a, b)
    assert a.header == hdr and a.table.n == 500 and a.table.gq.max() == 59 and set(a.orig_filter) == {"RefCall"}
    res = S.FilterResult((np.arange(500) / 500).astype(np.float32), (np.arange(500) % 2).astype(np.uint8), (np.arange(500) % 4).astype(np.uint8))
    out_n, out_p = str(tmp_path / "n.vcf.gz"), str(tmp_path / "p.vcf.gz")
    assert nv.write_filtered_vcf(out_n, a, res) is True
    pv.write_filtered_vcf(out_p, b, res)
    assert open(out_n, "rb").read() == open(out_p, "rb").read()
    text = gzip.open(out_n, "rb").read().decode().split("\n")
    new = [x for x in text if x.startswith("##") and x not in hdr]
    assert len(new) == 5 and text[len(hdr) + 5 - 1].startswith("#CHROM")     # the five tags join the 3 401 meta lines
    native_index = open(out_n + ".tbi", "rb").read()
    os.remove(out_n + ".tbi")
    assert pv.tabix_index(out_n) and open(out_n + ".tbi", "rb").read() == native_index


@pytest.mark.parametrize("world", [1, 2, 3, 7])
def test_part_reads_are_the_equal_count_slices_of_the_callset(tmp_path, world):
    """read_vcf(part=(r, world)) - what rank r of the multi-process tool tokenises - returns records [b[r], b[r + 1]) of the
    file with b = shard.shard_bounds(n, world); a sorted file's parts concatenate to the whole table; an unsorted slice says so;
    a part cannot be written back."""
    from variantcalling_amd import shard
    cs = synth.make_callset(5_003, genome_len=4_000_000, n_contigs=3, seed=17)
    p = str(tmp_path / "in.vcf.gz")
    pv.write_vcf_from_table(p, cs.variants, cs.ref.names)
    full = nv.read_vcf(p, cs.ref.names)
    b = shard.shard_bounds(full.n, world)
    assert full.n_total == full.n and full.part_lo == 0 and full.sorted_in_file
    for r in range(world):
        part = nv.read_vcf(p, cs.ref.names, part=(r, world))
        lo, hi = int(b[r]), int(b[r + 1])
        assert (part.n_total, part.part_lo, part.n) == (full.n, lo, hi - lo) and part.sorted_in_file
        want = full.table.slice(lo, hi)
        for c in COLS:
            assert np.array_equal(getattr(part.table, c), getattr(want, c), equal_nan=getattr(want, c).dtype.kind == "f"), c
        assert part.header == full.header
        if hi > lo:
            assert part.record_line(0) == full.record_line(lo)
        if world > 1:
            res = S.FilterResult(np.zeros(part.n, np.float32), np.zeros(part.n, np.uint8), np.zeros(part.n, np.uint8))
            with pytest.raises(RuntimeError, match="whole file"):
                nv.write_filtered_vcf(str(tmp_path / "no.vcf"), part, res)
        part.close()
    full.close()
    # an unsorted file: the slice is sorted on its own, and says that it was not in file order
    lines = gzip.open(p, "rt").read().split("\n")
    hdr = [l for l in lines if l.startswith("#")]
    rec = [l for l in lines if l and not l.startswith("#")]
    rec[10], rec[20] = rec[20], rec[10]
    q = str(tmp_path / "unsorted.vcf")
    open(q, "w").write("\n".join(hdr + rec) + "\n")
    part = nv.read_vcf(q, cs.ref.names, part=(0, 2))
    assert not part.sorted_in_file
    part.close()
    with pytest.raises(ValueError, match="part"):
        nv.read_vcf(p, cs.ref.names, part=(2, 2))


def test_count_hook_is_called_once_with_the_record_count(tmp_path):
    """ugvc_vcf_set_count_hook (round 4): the reader tells the tool how many records it holds as soon as the lines are counted -
    once, on the calling thread, for the next read of that thread only (filter_variants_pipeline starts Engine.reserve there)."""
    from variantcalling_amd import synth
    from variantcalling_amd.io import vcf as pyvcf, vcf_native
    cs = synth.make_callset(3000, genome_len=2_000_000, n_contigs=3, seed=5)
    path = str(tmp_path / "calls.vcf.gz")
    pyvcf.write_vcf_from_table(path, cs.variants, cs.ref.names)
    seen = []
    a = vcf_native.read_vcf(path, list(cs.ref.names), on_count=lambda n, tb: seen.append((n, tb)))
    assert len(seen) == 1 and seen[0][0] == a.table.n and seen[0][1] > 0
    b = vcf_native.read_vcf(path, list(cs.ref.names))                      # (the hook does not outlive its read)
    assert len(seen) == 1 and b.table.n == a.table.n
    part = vcf_native.read_vcf(path, list(cs.ref.names), part=(1, 2), on_count=lambda n, tb: seen.append((n, tb)))
    assert len(seen) == 2 and seen[1][0] == part.table.n                  # (a part read reports the part's records)
