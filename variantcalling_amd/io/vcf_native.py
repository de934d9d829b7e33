"""ctypes binding of the native VCF <-> SoA codec (include/ugvc_vcf.h, csrc_host/vcf_codec.cpp).

`read_vcf` / `write_filtered_vcf` have the signatures and results of `io.vcf` (the pure-Python host
reference of the same two loops, which tests/test_vcf_native.py compares against byte for byte); the
pipelines call these.  The library must have been built (`__graft_entry__.build()` /
`make -C variantcalling_amd/csrc_host`): a missing library raises, there is no silent fallback."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .. import schema as S

# (UGVC_VCF_LIB: another build of the same library - the sanitizer build of tools/codec_asan.sh)
LIB_PATH = os.environ.get("UGVC_VCF_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libugvc_vcf.so")
_lib = None


class _View(C.Structure):
    _fields_ = [("n", C.c_int64), ("pool_bytes", C.c_int64),
                ("contig", C.c_void_p), ("pos", C.c_void_p), ("ref_len", C.c_void_p), ("alt_len", C.c_void_p),
                ("ref_off", C.c_void_p), ("alt_off", C.c_void_p), ("alleles", C.c_void_p),
                ("qual", C.c_void_p), ("sor", C.c_void_p), ("dp", C.c_void_p), ("ad_ref", C.c_void_p),
                ("ad_alt", C.c_void_p), ("gq", C.c_void_p), ("gt", C.c_void_p), ("tlod", C.c_void_p),
                ("has_id", C.c_void_p), ("order", C.c_void_p),
                ("header", C.c_void_p), ("header_bytes", C.c_int64),
                ("text", C.c_void_p), ("filter_off", C.c_void_p), ("filter_len", C.c_void_p),
                ("n_alt", C.c_void_p), ("rec_off", C.c_void_p), ("rec_len", C.c_void_p)]


class _FastaView(C.Structure):
    _fields_ = [("total", C.c_int64), ("n_contigs", C.c_int32), ("codes", C.c_void_p), ("contig_off", C.c_void_p),
                ("names", C.c_void_p), ("names_bytes", C.c_int64)]


class _IntervalsView(C.Structure):
    _fields_ = [("n", C.c_int64), ("contig", C.c_void_p), ("start", C.c_void_p), ("end", C.c_void_p)]


class _TrackView(C.Structure):
    _fields_ = [("n", C.c_int64), ("start", C.c_void_p), ("end", C.c_void_p), ("ptr", C.c_void_p)]


_COUNT_HOOK = C.CFUNCTYPE(None, C.c_int64, C.c_int64, C.c_void_p)


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` "
                               "(make -C variantcalling_amd/csrc_host)")
        lib = C.CDLL(LIB_PATH)
        lib.ugvc_vcf_last_error.restype = C.c_char_p
        lib.ugvc_vcf_read_part.restype = C.c_int
        lib.ugvc_vcf_read_part.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.POINTER(C.c_void_p)]
        lib.ugvc_vcf_set_count_hook.restype = None
        lib.ugvc_vcf_set_count_hook.argtypes = [_COUNT_HOOK, C.c_void_p]
        lib.ugvc_vcf_part_info.restype = C.c_int
        lib.ugvc_vcf_part_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        lib.ugvc_vcf_read.restype = C.c_int
        lib.ugvc_vcf_read.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(C.c_void_p)]
        lib.ugvc_vcf_get_view.restype = C.c_int
        lib.ugvc_vcf_get_view.argtypes = [C.c_void_p, C.POINTER(_View)]
        lib.ugvc_vcf_write_filtered.restype = C.c_int
        lib.ugvc_vcf_write_filtered.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_int64, C.c_int, C.c_int]
        lib.ugvc_vcf_free.restype = None
        lib.ugvc_vcf_free.argtypes = [C.c_void_p]
        lib.ugvc_vcf_format_f32.restype = C.c_int
        lib.ugvc_vcf_format_f32.argtypes = [C.c_float, C.c_char_p, C.c_int]
        lib.ugvc_vcf_abi_version.restype = C.c_int
        lib.ugvc_vcf_set_deflate.restype = C.c_int
        lib.ugvc_vcf_set_deflate.argtypes = [C.c_int]
        lib.ugvc_fasta_read.restype = C.c_int
        lib.ugvc_fasta_read.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        lib.ugvc_fasta_get_view.restype = C.c_int
        lib.ugvc_fasta_get_view.argtypes = [C.c_void_p, C.POINTER(_FastaView)]
        lib.ugvc_fasta_free.restype = None
        lib.ugvc_fasta_free.argtypes = [C.c_void_p]
        lib.ugvc_intervals_read.restype = C.c_int
        lib.ugvc_intervals_read.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        lib.ugvc_intervals_get_view.restype = C.c_int
        lib.ugvc_intervals_get_view.argtypes = [C.c_void_p, C.POINTER(_IntervalsView)]
        lib.ugvc_intervals_track.restype = C.c_int
        lib.ugvc_intervals_track.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(_TrackView)]
        lib.ugvc_intervals_free.restype = None
        lib.ugvc_intervals_free.argtypes = [C.c_void_p]
        lib.ugvc_bgzf_read.restype = C.c_int
        lib.ugvc_bgzf_read.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        lib.ugvc_blob_free.restype = None
        lib.ugvc_blob_free.argtypes = [C.c_void_p]
        _lib = lib
    return _lib


def _err(lib) -> str:
    return (lib.ugvc_vcf_last_error() or b"").decode(errors="replace")


def _arr(ptr, n, dtype):
    """Copy of a library-owned column (the handle may be freed before the table is)."""
    if n == 0 or not ptr:
        return np.zeros(0, dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


class _Owner:
    """Keeps a native handle alive for as long as a numpy view of its memory is (no copy of a 3 GB reference)."""

    def __init__(self, lib, handle, free):
        self._lib, self._h, self._free = lib, handle, free

    def __del__(self):
        try:
            if self._h is not None:
                getattr(self._lib, self._free)(self._h)
                self._h = None
        except Exception:
            pass


def _view(ptr, n, dtype, owner):
    """numpy view (read-only) of library-owned memory; `owner` (an _Owner) is released with the last view."""
    if n == 0 or not ptr:
        return np.zeros(0, dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    buf._owner = owner
    a = np.frombuffer(buf, dtype=dtype, count=n)
    a.flags.writeable = False
    return a


class NativeVcfFile:
    """Same attributes as io.vcf.VcfFile; `records` / `orig_filter` are materialised on first use."""

    def __init__(self, handle, lib, is_mutect):
        self._h, self._lib = handle, lib
        v = _View()
        if lib.ugvc_vcf_get_view(handle, C.byref(v)):
            raise RuntimeError(_err(lib))
        n = int(v.n)
        self.n = n
        self.header = C.string_at(v.header, v.header_bytes).decode().split("\n") if v.header_bytes else []
        # The columns are read-only VIEWS of the handle's memory (round 4: copying twenty 5 M-element columns and checking them
        # again in numpy was a fifth of the reader's time); the handle lives until close() AND the last view are gone.  What
        # schema.VariantTable.validate checks holds by construction - dtypes and shapes are the library's, rows are in (contig,
        # pos) order (a stable sort on that key), alleles are non-empty and inside the pool (rejected per record by the
        # tokeniser) - and is checked again, per row, by the engine at upload (csrc/host_rows.cpp).
        own = self._owner = _Owner(lib, handle, "ugvc_vcf_free")
        self.order = _view(v.order, n, np.int64, own)
        self.ids = _view(v.has_id, n, np.uint8, own).astype(bool)
        self.tlod = _view(v.tlod, n, np.float32, own) if is_mutect else None
        self.table = S.VariantTable(
            contig=_view(v.contig, n, np.uint16, own), pos=_view(v.pos, n, np.int32, own),
            ref_len=_view(v.ref_len, n, np.uint16, own), alt_len=_view(v.alt_len, n, np.uint16, own),
            ref_off=_view(v.ref_off, n, np.uint32, own), alt_off=_view(v.alt_off, n, np.uint32, own),
            alleles=_view(v.alleles, int(v.pool_bytes), np.uint8, own), qual=_view(v.qual, n, np.float32, own),
            sor=_view(v.sor, n, np.float32, own), dp=_view(v.dp, n, np.int32, own), ad_ref=_view(v.ad_ref, n, np.int32, own),
            ad_alt=_view(v.ad_alt, n, np.int32, own), gq=_view(v.gq, n, np.uint8, own), gt=_view(v.gt, n, np.uint8, own))
        self._orig_filter = None
        self.n_alt = _view(v.n_alt, n, np.uint8, own)
        self._rec_off, self._rec_len = _view(v.rec_off, n, np.int64, own), _view(v.rec_len, n, np.int32, own)

    def record_line(self, k: int) -> bytes:
        """The text of the record behind table row k (multi-allelic expansion reads the few rows it needs)."""
        v = _View()
        self._lib.ugvc_vcf_get_view(self._h, C.byref(v))
        return C.string_at(v.text + int(self._rec_off[k]), int(self._rec_len[k]))

    @property
    def orig_filter(self) -> list:
        if self._orig_filter is None:
            v = _View()
            self._lib.ugvc_vcf_get_view(self._h, C.byref(v))
            off = _arr(v.filter_off, self.n, np.int64)
            ln = _arr(v.filter_len, self.n, np.int32)
            self._orig_filter = [C.string_at(v.text + int(o), int(l)).decode() for o, l in zip(off, ln)]
        return self._orig_filter

    def close(self):
        # (the native handle is freed with the last view of its memory: _Owner)
        self._h = None
        self._owner = None


def read_vcf(path: str, contig_names: list, is_mutect: bool = False, sample: int = 0, n_threads: int = 0,
             part: tuple | None = None, on_count=None) -> NativeVcfFile:
    """The whole callset, or - `part = (rank, world)` - the equal-count slice of its records (file order) that one rank of a
    multi-process run scores: `.n_total` records are in the file, `.part_lo` is the file-order index of the slice's first
    record, `.sorted_in_file` says whether the slice was sorted by (contig, pos) as it stood (the caller checks the seams
    between ranks: pipelines/filter_variants_pipeline.py).  A part cannot be written back."""
    lib = load_library()
    names = (C.c_char_p * len(contig_names))(*[n.encode() for n in contig_names])
    h = C.c_void_p()
    rank, world = part if part is not None else (0, 1)
    hook = None
    if on_count is not None:                                # on_count(n_records, text_bytes): once, when the lines are counted
        hook = _COUNT_HOOK(lambda n, tb, _u: on_count(int(n), int(tb)))
        lib.ugvc_vcf_set_count_hook(hook, None)
    try:
        rc = lib.ugvc_vcf_read_part(os.fsencode(path), names, len(contig_names), int(is_mutect), int(sample), int(n_threads),
                                    int(rank), int(world), C.byref(h))
    finally:
        if hook is not None:
            lib.ugvc_vcf_set_count_hook(_COUNT_HOOK(0), None)
    if rc:
        raise ValueError(_err(lib))
    f = NativeVcfFile(h, lib, is_mutect)
    nt, lo = C.c_int64(), C.c_int64()
    lib.ugvc_vcf_part_info(h, C.byref(nt), C.byref(lo))
    f.n_total, f.part_lo = int(nt.value), int(lo.value)
    f.sorted_in_file = bool(f.n == 0 or (int(f.order[0]) == 0 and int(f.order[-1]) == f.n - 1 and bool(np.all(f.order[1:] > f.order[:-1]))))
    return f


def write_filtered_vcf(path: str, vcf: NativeVcfFile, res: S.FilterResult, blacklist_cg: np.ndarray | None = None,
                       n_threads: int = 0, index: bool = True) -> bool:
    """Returns True when a tabix index `path`.tbi was written beside a .gz output (sorted input), else False."""
    lib = load_library()
    ts = np.ascontiguousarray(res.tree_score, np.float32)
    fl = np.ascontiguousarray(res.filter, np.uint8)
    fg = np.ascontiguousarray(res.flags, np.uint8)
    cg = None if blacklist_cg is None else np.ascontiguousarray(np.asarray(blacklist_cg).astype(bool), np.uint8)
    rc = lib.ugvc_vcf_write_filtered(vcf._h, os.fsencode(path), ts.ctypes.data, fl.ctypes.data, fg.ctypes.data,
                                     None if cg is None else cg.ctypes.data, int(ts.size), int(n_threads), int(index))
    if rc < 0:
        raise RuntimeError(_err(lib))
    return bool(index) and path.endswith(".gz") and rc == 0


def format_f32(x: float) -> str:
    buf = C.create_string_buffer(64)
    n = load_library().ugvc_vcf_format_f32(float(x), buf, 64)
    if n < 0:
        raise RuntimeError("format_f32 failed")
    return buf.value.decode()


def read_fasta_names(path: str) -> list:
    """Contig names in file order: from the .fai index beside the FASTA when there is one, else from the file."""
    fai = path + ".fai"
    if os.path.exists(fai):
        with open(fai) as fh:
            return [ln.split("\t")[0] for ln in fh if ln.strip()]
    return list(read_fasta(path).names)


def read_fasta(path: str, contigs: list | None = None, n_threads: int = 0) -> S.Reference:
    """io.fasta.read_fasta through the native reader (threaded inflate + encode)."""
    lib = load_library()
    h = C.c_void_p()
    if lib.ugvc_fasta_read(os.fsencode(path), int(n_threads), C.byref(h)):
        raise ValueError(_err(lib))
    owner = _Owner(lib, h, "ugvc_fasta_free")
    v = _FastaView()
    lib.ugvc_fasta_get_view(h, C.byref(v))
    names = C.string_at(v.names, v.names_bytes).decode().split("\n") if v.n_contigs else []
    off = _arr(v.contig_off, int(v.n_contigs) + 1, np.int64)
    if contigs is None:
        codes = _view(v.codes, int(v.total), np.uint8, owner)     # the encoded genome stays where the reader put it
    else:
        keep = [k for k, nm in enumerate(names) if nm in contigs]
        if not keep:
            raise ValueError(f"{path}: no sequences read")
        whole = _view(v.codes, int(v.total), np.uint8, owner)
        codes = np.concatenate([whole[off[k]: off[k + 1]] for k in keep])
        sizes = [int(off[k + 1] - off[k]) for k in keep]
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        names = [names[k] for k in keep]
    return S.Reference(codes, off, names)


def read_intervals(path: str, contig_names: list, merge: bool = True, n_threads: int = 0, native_track: bool = True) -> S.IntervalTrack:
    """io.bed.read_intervals through the native tokeniser AND the native sort / merge (ugvc_intervals_track, round 6);
    `native_track=False`: the rows go through bed.track_from_arrays instead - its checker (tests/test_vcf_native.py)."""
    from . import bed
    lib = load_library()
    names = (C.c_char_p * len(contig_names))(*[n.encode() for n in contig_names])
    stem = os.path.basename(path)
    for suf in (".gz", ".bed", ".interval_list"):
        if stem.endswith(suf):
            stem = stem[: -len(suf)]
    h = C.c_void_p()
    if lib.ugvc_intervals_read(os.fsencode(path), names, len(contig_names), int(n_threads), C.byref(h)):
        raise ValueError(_err(lib))
    try:
        if native_track:
            t = _TrackView()
            if lib.ugvc_intervals_track(h, len(contig_names), int(bool(merge)), C.byref(t)):
                raise ValueError(_err(lib))
            n = int(t.n)
            return S.IntervalTrack(_arr(t.start, n, np.int32), _arr(t.end, n, np.int32), _arr(t.ptr, len(contig_names) + 1, np.int32), stem)
        v = _IntervalsView()
        lib.ugvc_intervals_get_view(h, C.byref(v))
        n = int(v.n)
        c, s, e = _arr(v.contig, n, np.int64), _arr(v.start, n, np.int64), _arr(v.end, n, np.int64)
    finally:
        lib.ugvc_intervals_free(h)
    return bed.track_from_arrays(c, s, e, len(contig_names), stem, merge)


def bgzf_read(path: str, n_threads: int = 0) -> bytes:
    """The (inflated) bytes of a plain / gzip / BGZF file through the codec's own reader (ugvc_bgzf_read) - htslib's bgzf_read."""
    lib = load_library()
    h, data, n = C.c_void_p(), C.c_void_p(), C.c_int64()
    if lib.ugvc_bgzf_read(os.fsencode(path), int(n_threads), C.byref(h), C.byref(data), C.byref(n)):
        raise ValueError(_err(lib))
    try:
        return C.string_at(data, n.value) if n.value else b""
    finally:
        lib.ugvc_blob_free(h)


def set_deflate(backend: str = "auto") -> str:
    """Which deflate implementation the BGZF reader / writer use: "auto" (libdeflate when the host has it, else zlib), "zlib"
    (level 6: the compressed bytes are then those of the pure-Python reference codec) or "libdeflate".  Returns the one in use."""
    lib = load_library()
    rc = lib.ugvc_vcf_set_deflate({"auto": 0, "zlib": 1, "libdeflate": 2}[backend])
    if rc < 0:
        raise RuntimeError(lib.ugvc_vcf_last_error().decode())
    return "libdeflate" if rc == 2 else "zlib"
