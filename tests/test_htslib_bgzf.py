"""A real htslib artefact against the codec: `somatic_test.ann.CHR19.vcf.gz.csi` of the reference's own test
resources (/root/reference/test/resources/system/test_no_gt_report/, committed under tests/golden/htslib/ as the data file it is) is
a genuine BGZF stream written by htslib - the one thing in the reference tree that pins the codec's BGZF framing to something the
builder did not write: the native reader's inflate with BOTH deflate back ends and the pure-Python statement against Python's gzip,
the block header and end-of-file marker the writers emit against htslib's bytes, and the binning arithmetic of the index writer
against the bins a real index holds (CSI / tabix specification, samtools.github.io/hts-specs)."""
import gzip
import os
import struct

import numpy as np
import pytest

from variantcalling_amd.io import vcf as pyvcf
from variantcalling_amd.io import vcf_native

HERE = os.path.dirname(os.path.abspath(__file__))
CSI = os.path.join(HERE, "golden", "htslib", "somatic_test.ann.CHR19.vcf.gz.csi")


def _members(raw: bytes):
    """(offset, total size, header bytes) of every gzip member of a BGZF stream, by the BSIZE field alone."""
    off, out = 0, []
    while off < len(raw):
        assert raw[off:off + 4] == b"\x1f\x8b\x08\x04", f"member at {off} is not gzip + FEXTRA"
        xlen = struct.unpack_from("<H", raw, off + 10)[0]
        assert raw[off + 12:off + 14] == b"BC" and struct.unpack_from("<H", raw, off + 14)[0] == 2 and xlen == 6
        bsize = struct.unpack_from("<H", raw, off + 16)[0] + 1
        out.append((off, bsize, raw[off:off + 16]))
        off += bsize
    assert off == len(raw)
    return out


def test_real_htslib_stream_inflates_identically_with_every_back_end(tmp_path):
    raw = open(CSI, "rb").read()
    want = gzip.decompress(raw)                              # Python's zlib walks the members itself
    assert want[:4] == b"CSI\x01" and len(want) > len(raw)
    used = []
    for backend in ("zlib", "libdeflate"):
        try:
            got_backend = vcf_native.set_deflate(backend)
        except RuntimeError:
            continue                                         # (a host without libdeflate: the zlib leg still runs)
        used.append(got_backend)
        try:
            for threads in (1, 4):
                assert vcf_native.bgzf_read(CSI, threads) == want, f"{backend}, {threads} threads"
        finally:
            vcf_native.set_deflate("auto")
    assert "zlib" in used
    # the pure-Python statement of the reader (io/vcf.py: the native codec's checker) on the same stream
    with pyvcf._open(CSI) as fh:
        assert fh.read() == want
    # a truncated copy (the end-of-file marker and half of the last data member cut off) is refused, not half-read
    cut = tmp_path / "cut.csi"
    cut.write_bytes(raw[:len(raw) - 28 - 200])
    with pytest.raises(ValueError):
        vcf_native.bgzf_read(str(cut))


def test_block_header_and_eof_marker_are_htslibs(tmp_path):
    raw = open(CSI, "rb").read()
    mem = _members(raw)
    assert len(mem) >= 2
    # htslib ends every BGZF file with one empty member: 28 bytes
    assert mem[-1][1] == 28
    eof = raw[-28:]
    assert eof == pyvcf._BGZF_EOF
    # what the codec's writers put in front of a block: the 16 header bytes of htslib's members (MTIME 0, XFL 0, OS 255, 'BC' 2)
    head = {h for _, _, h in mem}
    assert head == {pyvcf._bgzf_block(b"x")[:16]}
    # ... and the native writer: a small filtered VCF written by libugvc_vcf.so starts with the same 16 bytes, ends with the same 28
    from variantcalling_amd import schema as S
    from variantcalling_amd import synth
    cs = synth.make_callset(300, genome_len=200_000, n_contigs=2, seed=5)
    src = tmp_path / "in.vcf.gz"
    pyvcf.write_vcf_from_table(str(src), cs.variants, cs.ref.names)
    v = vcf_native.read_vcf(str(src), cs.ref.names)
    res = S.FilterResult(np.linspace(0, 1, v.table.n, dtype=np.float32), np.zeros(v.table.n, np.uint8), np.zeros(v.table.n, np.uint8))
    out = tmp_path / "out.vcf.gz"
    vcf_native.write_filtered_vcf(str(out), v, res)
    v.close()
    got = out.read_bytes()
    assert got[:16] == mem[0][2] and got[-28:] == eof
    for _, _, h in _members(got):
        assert h == mem[0][2]
    assert gzip.decompress(got).startswith(b"##fileformat=VCF")


def _reg2bin_spec(beg: int, end: int, min_shift: int, depth: int) -> int:
    """reg2bin of the CSI specification (hts-specs CSIv1.pdf), which tabix's fixed scheme is the (14, 5) case of."""
    end -= 1
    s, t = min_shift, ((1 << depth * 3) - 1) // 7
    for level in range(depth, 0, -1):
        if beg >> s == end >> s:
            return t + (beg >> s)
        s += 3
        t -= 1 << (level - 1) * 3
    return 0


def test_binning_arithmetic_against_a_real_index():
    d = gzip.decompress(open(CSI, "rb").read())
    magic, min_shift, depth, l_aux = struct.unpack_from("<4siii", d, 0)
    assert (magic, min_shift, depth) == (b"CSI\x01", 14, 6)
    # aux = the tabix header: format 2 = VCF, sequence column 1, begin column 2, no end column, '#' comments
    fmt, col_seq, col_beg, col_end, meta, skip, l_nm = struct.unpack_from("<7i", d, 16)
    assert (fmt, col_seq, col_beg, col_end, meta, skip) == (2, 1, 2, 0, ord("#"), 0)
    names = d[16 + 28:16 + 28 + l_nm].split(b"\x00")[:-1]
    off = 16 + l_aux
    n_ref = struct.unpack_from("<i", d, off)[0]
    off += 4
    assert n_ref == len(names) and b"chr19" in names
    n_bins_max = ((1 << (depth + 1) * 3) - 1) // 7                # bins 0 .. n_bins_max - 1, pseudo-bin n_bins_max + 1
    level_first = [((1 << 3 * lv) - 1) // 7 for lv in range(depth + 2)]
    seen_chunks = 0
    for _ in range(n_ref):
        n_bin = struct.unpack_from("<i", d, off)[0]
        off += 4
        for _ in range(n_bin):
            b, loffset, n_chunk = struct.unpack_from("<IQi", d, off)
            off += 16
            chunks = [struct.unpack_from("<QQ", d, off + 16 * k) for k in range(n_chunk)]
            off += 16 * n_chunk
            if b == n_bins_max + 1:                               # the pseudo-bin: {file range, mapped / unmapped counts}
                assert n_chunk == 2
                continue
            assert b < n_bins_max
            lv = max(k for k in range(depth + 1) if level_first[k] <= b)
            # a bin at level lv covers [k << s, (k + 1) << s) with s = min_shift + 3 (depth - lv): reg2bin of the spec, which the
            # codec's index writer implements for (14, 5) (io/vcf.py: _reg2bin == csrc_host/vcf_codec.cpp: reg2bin), lands on it
            s = min_shift + 3 * (depth - lv)
            k = b - level_first[lv]
            beg = k << s
            assert _reg2bin_spec(beg, beg + (1 << s), min_shift, depth) == b
            for cb, ce in chunks:
                assert cb < ce and (cb & 0xFFFF) < 65536              # virtual offsets ascend inside a chunk
                assert loffset <= cb or lv < depth                 # (a leaf's loffset is the start of its first record)
            seen_chunks += n_chunk
    assert seen_chunks > 0 and off + 8 >= len(d) - 8              # (n_no_coor may follow)
    # the fixed tabix scheme of the codec against the general formula, on random regions
    rng = np.random.default_rng(3)
    for beg, ln in zip(rng.integers(0, 1 << 29, 3000), rng.integers(1, 1 << 20, 3000)):
        beg, end = int(beg), int(min(beg + ln, 1 << 29))
        assert pyvcf._reg2bin(beg, end) == _reg2bin_spec(beg, end, 14, 5)
