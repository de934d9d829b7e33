"""ctypes binding of libugvc_mi355x.so (include/ugvc_mi355x.h) - the only compute path.

There is NO CPU fallback: if the shared library is missing or no MI355X is visible this
module raises, loudly.  Host side mirrors the reference's pandas-level calls
(`annotate_concordance(df, fasta, ...)`, blacklist apply, `model.predict`;
call pattern ugvc/pipelines/run_no_gt_report.py:92-94,314 and SURVEY.md §3.1) on SoA tables.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import schema as S

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libugvc_mi355x.so")

_u8p, _u16p, _u32p, _u64p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint16, C.c_uint32, C.c_uint64))
_i32p, _i64p, _f32p, _f64p = (C.POINTER(t) for t in (C.c_int32, C.c_int64, C.c_float, C.c_double))


class CVariants(C.Structure):
    _fields_ = [("n", C.c_int64), ("contig", _u16p), ("pos", _i32p), ("ref_len", _u16p), ("alt_len", _u16p),
                ("ref_off", _u32p), ("alt_off", _u32p), ("alleles", _u8p), ("alleles_len", C.c_int64),
                ("qual", _f32p), ("sor", _f32p), ("dp", _i32p), ("ad_ref", _i32p), ("ad_alt", _i32p),
                ("gq", _u8p)]


class CResults(C.Structure):
    _fields_ = [("tree_score", _f32p), ("filter", _u8p), ("flags", _u8p)]


class CPileupOut(C.Structure):
    _fields_ = [(k, _i32p) for k in ("ref_fwd", "ref_rev", "alt_fwd", "alt_rev", "other", "dp", "bq_ref", "bq_alt")] + \
               [("vaf", _f32p), ("sor", _f32p)]


class CBridgingParams(C.Structure):
    _fields_ = [("min_initial_qual", C.c_double), ("min_tumor_vaf", C.c_double), ("max_normal_vaf", C.c_double),
                ("min_query_hmer_size", C.c_int), ("min_normal_depth", C.c_int), ("min_distance_from_edge", C.c_int)]


# every symbol include/ugvc_mi355x.h declares: (restype, argtypes)
_ctx = C.c_void_p
ABI = {
    "ugvc_abi_version": (C.c_int, []),
    "ugvc_last_error": (C.c_char_p, []),
    "ugvc_ctx_create": (C.c_int, [C.c_int, C.POINTER(_ctx)]),
    "ugvc_ctx_destroy": (C.c_int, [_ctx]),
    "ugvc_device_info": (C.c_int, [_ctx, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "ugvc_device_attr": (C.c_int, [_ctx, C.c_int, C.POINTER(C.c_int64)]),
    "ugvc_reserve": (C.c_int, [_ctx, C.c_int64, C.c_int64]),
    "ugvc_sync": (C.c_int, [_ctx]),
    "ugvc_resident_count": (C.c_int, [_ctx, _i64p, C.POINTER(C.c_int)]),
    "ugvc_selftest": (C.c_int, [_ctx, C.c_int64]),
    "ugvc_ref_upload": (C.c_int, [_ctx, _u8p, C.c_int64, _i64p, C.c_int]),
    "ugvc_runs_upload": (C.c_int, [_ctx, _i32p, _i32p, _i32p, C.c_int64, C.c_int, C.c_int, C.c_int]),
    "ugvc_track_upload": (C.c_int, [_ctx, C.c_int, _i32p, _i32p, _i32p, C.c_int64]),
    "ugvc_set_n_tracks": (C.c_int, [_ctx, C.c_int]),
    "ugvc_blacklist_upload": (C.c_int, [_ctx, _u64p, C.c_int64]),
    "ugvc_set_flow_order": (C.c_int, [_ctx, C.c_char_p]),
    "ugvc_model_upload": (C.c_int, [_ctx, C.c_int, C.c_int, _i32p, _f32p, _i32p, _i32p, C.c_int32, _i32p,
                                    C.c_int32, _f64p, C.c_int32, C.c_int32, C.c_float, C.c_int32]),
    "ugvc_model_clear": (C.c_int, [_ctx, C.c_int]),
    "ugvc_filter_variants": (C.c_int, [_ctx, C.POINTER(CVariants), C.POINTER(CResults)]),
    "ugvc_variants_upload": (C.c_int, [_ctx, C.POINTER(CVariants)]),
    "ugvc_filter_resident": (C.c_int, [_ctx]),
    "ugvc_results_download": (C.c_int, [_ctx, C.POINTER(CResults)]),
    "ugvc_timed_filter": (C.c_int, [_ctx, C.c_int, _f32p]),
    "ugvc_timed_steps": (C.c_int, [_ctx, C.c_int, C.c_int64, C.c_int, _f32p, _f32p]),
    "ugvc_set_step_events": (C.c_int, [_ctx, C.c_int]),
    "ugvc_pass_clock": (C.c_int, [_ctx, C.c_int, _f64p, _f64p]),
    "ugvc_device_sync": (C.c_int, [_ctx]),
    "ugvc_last_step_ms": (C.c_int, [_ctx, _f32p, C.c_int]),
    "ugvc_feature_matrix": (C.c_int, [_ctx, _f32p, _u8p]),
    "ugvc_n_features": (C.c_int, [_ctx]),
    "ugvc_forest_gemm": (C.c_int, [_ctx, C.c_int, _i32p, C.c_int64, C.c_int, C.c_int, _f32p, _f32p]),
    "ugvc_forest_gemm3": (C.c_int, [_ctx, C.POINTER(_i32p), _i64p, C.c_int, _f32p, _f32p]),
    "ugvc_host_css_lut": (C.c_int, [C.c_char_p, _u8p]),
    "ugvc_set_kernel_variant": (C.c_int, [_ctx, C.c_int]),
    "ugvc_debug_phase_clocks": (C.c_int, [_ctx, C.POINTER(C.c_uint64), C.c_int]),
    "ugvc_eval_counts": (C.c_int, [_ctx, C.POINTER(C.c_int8), C.POINTER(C.c_uint16), C.POINTER(C.c_int64)]),
    "ugvc_pr_curve": (C.c_int, [_ctx, C.POINTER(C.c_double), _u8p, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                _i32p, _f32p]),
    "ugvc_pileup_tally": (C.c_int, [_ctx, _i64p, _u16p, C.c_int64, C.POINTER(CPileupOut)]),
    "ugvc_pileup_upload": (C.c_int, [_ctx, _i64p, _u16p, C.c_int64]),
    "ugvc_timed_pileup": (C.c_int, [_ctx, C.c_int, _f32p]),
    "ugvc_sec_likelihood_ratio": (C.c_int, [_ctx, _i32p, _i32p, C.c_int64, C.c_int, _f64p, _f64p]),
    "ugvc_sec_db_build": (C.c_int, [_ctx, _u64p, _i32p, C.c_int64, C.c_int, _u64p, _i32p, _i64p]),
    "ugvc_sec_db_upload": (C.c_int, [_ctx, _u64p, _i32p, C.c_int64, C.c_int]),
    "ugvc_sec_apply": (C.c_int, [_ctx, C.c_double, C.c_int, C.c_int, _f64p, _u8p]),
    "ugvc_timed_sec_apply": (C.c_int, [_ctx, C.c_double, C.c_int, C.c_int, _f32p]),
    "ugvc_bridging_snvs": (C.c_int, [_ctx, C.POINTER(CVariants), _u8p, _i32p, _i32p, _i32p,
                                     C.POINTER(CBridgingParams), _u8p, _u8p]),
    "ugvc_timed_feature_matrix": (C.c_int, [_ctx, C.c_int, _f32p]),
    "ugvc_comm_unique_id": (C.c_int, [_u8p]),
    "ugvc_comm_init": (C.c_int, [_ctx, _u8p, C.c_int, C.c_int]),
    "ugvc_comm_info": (C.c_int, [_ctx, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ugvc_comm_destroy": (C.c_int, [_ctx]),
    "ugvc_allgather_resident": (C.c_int, [_ctx, C.c_int64]),
    "ugvc_gather_fence": (C.c_int, [_ctx]),
    "ugvc_gather_target": (C.c_int, [_ctx, C.c_int64, C.POINTER(_f32p), C.POINTER(_u8p), C.POINTER(_u8p)]),
    "ugvc_gather_launch": (C.c_int, [_ctx, C.c_int64]),
    "ugvc_gathered_download": (C.c_int, [_ctx, C.c_int64, C.c_int, C.POINTER(CResults)]),
}

_lib = None


def load_library(path: str = LIB_PATH):
    """dlopen the engine and bind every ABI symbol; raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C variantcalling_amd/csrc`.  There is no CPU fallback.")
    # (multi-process GPU work on this stack needs dmabuf IPC: without it RCCL fails with `hipIpcGetMemHandle: invalid argument`)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in ABI.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.ugvc_abi_version() != 2:
        raise RuntimeError("libugvc_mi355x.so ABI version mismatch")
    _lib = lib
    return lib


def _p(a: np.ndarray, typ):
    return a.ctypes.data_as(typ)


def _col(a, dtype) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=dtype)


class Engine:
    """One GPU context (= one ugvc_ctx).  Methods raise RuntimeError(ugvc_last_error())."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = _ctx()
        self._h = None
        self._check(self.lib.ugvc_ctx_create(device, C.byref(h)))
        self._h = h
        self.n_tracks = 0
        self.n = 0
        self._keep = []   # host arrays that must outlive async calls

    def _check(self, rc: int):
        if rc != 0:
            raise RuntimeError(self.lib.ugvc_last_error().decode() or f"ugvc error {rc}")

    def close(self):
        if self._h is not None:
            self.lib.ugvc_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- info
    def device_info(self) -> dict:
        name = C.create_string_buffer(256)
        cus, mem = C.c_int(), C.c_int64()
        self._check(self.lib.ugvc_device_info(self._h, name, 256, C.byref(cus), C.byref(mem)))
        return dict(name=name.value.decode(), n_cus=cus.value, hbm_bytes=mem.value)

    def device_attr(self, what: str) -> int:
        """One integer property of the device: "clock_khz", "n_cus", "mem_clock_khz", "lds_bytes"."""
        out = C.c_int64()
        self._check(self.lib.ugvc_device_attr(self._h, ("clock_khz", "n_cus", "mem_clock_khz", "lds_bytes").index(what), C.byref(out)))
        return int(out.value)

    def selftest(self, n: int = 1024) -> None:
        """Device canary: copy round trip + the library's prefix-sum kernel over n words, checked on the host.  Raises with
        the failing step's name when the GPU cannot be used at all (a dead box is not a kernel bug)."""
        self._check(self.lib.ugvc_selftest(self._h, n))

    def sync(self):
        self._check(self.lib.ugvc_sync(self._h))

    # ---- resident tables
    def set_reference(self, ref: S.Reference):
        codes = _col(ref.codes, np.uint8)
        off = _col(ref.contig_off, np.int64)
        buf = codes if codes.size else np.zeros(1, np.uint8)         # (an empty array has no address to hand over)
        self._check(self.lib.ugvc_ref_upload(self._h, _p(buf, _u8p), codes.size, _p(off, _i64p), off.size - 1))
        self.n_contigs = off.size - 1

    def set_contigs(self, names: list):
        """Contig dictionary without bases - for the tools that join on (contig, pos) only (SEC apply): the variant upload
        validates contig indices against it."""
        self.set_reference(S.Reference(np.zeros(0, np.uint8), np.zeros(len(names) + 1, np.int64), list(names)))

    def set_runs(self, runs: S.IntervalTrack, min_len: int = 10, max_dist: int = 10, mark_hpol: bool = True):
        s, e, p = _col(runs.starts, np.int32), _col(runs.ends, np.int32), _col(runs.contig_ptr, np.int32)
        self._check(self.lib.ugvc_runs_upload(self._h, _p(s, _i32p), _p(e, _i32p), _p(p, _i32p), s.size,
                                              int(min_len), int(max_dist), int(bool(mark_hpol))))

    def set_tracks(self, tracks: list):
        if len(tracks) > S.MAX_TRACKS:
            raise ValueError(f"at most {S.MAX_TRACKS} annotation tracks")
        for t, tr in enumerate(tracks):
            s, e, p = _col(tr.starts, np.int32), _col(tr.ends, np.int32), _col(tr.contig_ptr, np.int32)
            self._check(self.lib.ugvc_track_upload(self._h, t, _p(s, _i32p), _p(e, _i32p), _p(p, _i32p), s.size))
        self._check(self.lib.ugvc_set_n_tracks(self._h, len(tracks)))
        self.n_tracks = len(tracks)

    def set_track(self, t: int, tr: S.IntervalTrack):
        """One annotation track into slot t (a tool uploads each as its reader finishes; set_n_tracks when all are there)."""
        if not 0 <= t < S.MAX_TRACKS:
            raise ValueError(f"at most {S.MAX_TRACKS} annotation tracks")
        s, e, p = _col(tr.starts, np.int32), _col(tr.ends, np.int32), _col(tr.contig_ptr, np.int32)
        self._check(self.lib.ugvc_track_upload(self._h, t, _p(s, _i32p), _p(e, _i32p), _p(p, _i32p), s.size))

    def set_n_tracks(self, n: int):
        self._check(self.lib.ugvc_set_n_tracks(self._h, int(n)))
        self.n_tracks = int(n)

    def set_blacklist(self, keys: np.ndarray | None):
        k = _col(keys if keys is not None else np.zeros(0, np.uint64), np.uint64)
        self._check(self.lib.ugvc_blacklist_upload(self._h, _p(k, _u64p), k.size))

    def set_flow_order(self, flow: str):
        self._check(self.lib.ugvc_set_flow_order(self._h, flow.encode()))

    def set_model(self, group: int, f: S.FlatForest):
        feat, thr = _col(f.feature, np.int32), _col(f.threshold, np.float32)
        left, right = _col(f.left, np.int32), _col(f.right, np.int32)
        roots, leaves = _col(f.tree_root, np.int32), _col(f.leaf_value, np.float64)
        # the C ABI takes bare pointers: shapes are checked here (a hand-made or damaged model file must not make the
        # library read past an array)
        if not (feat.ndim == thr.ndim == left.ndim == right.ndim == roots.ndim == 1):
            raise ValueError("model node / root arrays must be one-dimensional")
        if not (feat.size == thr.size == left.size == right.size):
            raise ValueError(f"model node arrays differ in length: feature {feat.size}, threshold {thr.size}, "
                             f"left {left.size}, right {right.size}")
        if leaves.ndim != 2 or leaves.shape[1] != 2:
            raise ValueError(f"leaf_value must be [n_leaves, 2], got shape {leaves.shape}")
        self._check(self.lib.ugvc_model_upload(
            self._h, group, f.kind, _p(feat, _i32p), _p(thr, _f32p), _p(left, _i32p), _p(right, _i32p),
            feat.size, _p(roots, _i32p), roots.size, _p(leaves, _f64p), leaves.shape[0], f.n_features,
            float(f.base_score), 0))

    def set_models(self, forests: list):
        for g, f in enumerate(forests):
            if f is not None:
                self.set_model(g, f)
            else:                           # a re-configured context must not keep the previous job's model
                self._check(self.lib.ugvc_model_clear(self._h, g))
        for g in range(len(forests), S.N_GROUPS):
            self._check(self.lib.ugvc_model_clear(self._h, g))

    def reserve(self, n_variants: int, alleles_len: int):
        """Allocate what `filter_variants` allocates once per callset size, and load the kernels, without touching data
        (a tool calls it from a helper thread while it reads / uploads its other inputs)."""
        self._check(self.lib.ugvc_reserve(self._h, int(n_variants), int(alleles_len)))
        self.n = self.resident_count()[0]           # (a reservation that re-allocates a resident column empties the context)

    def _follow_resident_count(self):
        """The cached row count follows what the context holds (0 after a failed upload / boundary call); never raises - it runs
        in `finally` blocks and must not replace the error that is on its way out."""
        try:
            self.n = self.resident_count()[0]
        except Exception:                                       # noqa: BLE001
            self.n = 0

    def resident_count(self):
        """(rows resident, whether the resident result columns hold a scoring pass over them)."""
        n, sc = C.c_int64(), C.c_int()
        self._check(self.lib.ugvc_resident_count(self._h, C.byref(n), C.byref(sc)))
        return int(n.value), bool(sc.value)

    def set_kernel_variant(self, v: int):
        self._check(self.lib.ugvc_set_kernel_variant(self._h, v))

    # ---- score evaluation (consumers of the resident FILTER / tree_score columns)
    def eval_counts(self, label: np.ndarray, cat_bits: np.ndarray) -> np.ndarray:
        """int64 [16, 4] = per category {true, false, true & PASS, false & PASS} over the resident FILTER column;
        label: 1 true / 0 false / < 0 unlabelled, cat_bits: u16 category membership bits."""
        lab = np.ascontiguousarray(label, np.int8)
        cb = np.ascontiguousarray(cat_bits, np.uint16)
        if lab.size != self.n or cb.size != self.n:
            raise ValueError("label / cat_bits must have one entry per resident variant")
        out = np.zeros((16, 4), np.int64)
        self._check(self.lib.ugvc_eval_counts(self._h, lab.ctypes.data_as(C.POINTER(C.c_int8)),
                                              cb.ctypes.data_as(C.POINTER(C.c_uint16)),
                                              out.ctypes.data_as(C.POINTER(C.c_int64))))
        return out

    def pr_curve(self, score: np.ndarray, cls: np.ndarray, initial_tp: int, initial_fp: int, initial_fn: int,
                 want_order: bool = False):
        """(sorted score, recall, precision, f1[, order], device ms): evaluate.calc_performance's curve on the GPU."""
        s = np.ascontiguousarray(score, np.float64)
        c = np.ascontiguousarray(cls, np.uint8)
        n = s.size
        out = [np.zeros(n, np.float64) for _ in range(4)]
        order = np.zeros(n, np.int32) if want_order else None
        ms = C.c_float()
        dp = C.POINTER(C.c_double)
        self._check(self.lib.ugvc_pr_curve(self._h, s.ctypes.data_as(dp), _p(c, _u8p), n, int(initial_tp), int(initial_fp),
                                           int(initial_fn), *[o.ctypes.data_as(dp) for o in out],
                                           None if order is None else _p(order, _i32p), C.byref(ms)))
        return (*out, order, ms.value)

    def phase_clocks(self, reset: bool = True) -> list:
        out = (C.c_uint64 * 8)()
        self._check(self.lib.ugvc_debug_phase_clocks(self._h, out, int(reset)))
        return list(out)

    # ---- hot path
    def _cvariants(self, vt: S.VariantTable) -> CVariants:
        cols = {c: _col(getattr(vt, c), S.VariantTable.DTYPES[c]) for c in S.VariantTable.COLS if c != "gt"}
        alle = _col(vt.alleles, np.uint8)
        self._keep = [cols, alle]
        return CVariants(vt.n, _p(cols["contig"], _u16p), _p(cols["pos"], _i32p), _p(cols["ref_len"], _u16p),
                         _p(cols["alt_len"], _u16p), _p(cols["ref_off"], _u32p), _p(cols["alt_off"], _u32p),
                         _p(alle, _u8p), alle.size, _p(cols["qual"], _f32p), _p(cols["sor"], _f32p),
                         _p(cols["dp"], _i32p), _p(cols["ad_ref"], _i32p), _p(cols["ad_alt"], _i32p),
                         _p(cols["gq"], _u8p))

    @staticmethod
    def _alloc_results(n: int):
        res = S.FilterResult(np.zeros(n, np.float32), np.zeros(n, np.uint8), np.zeros(n, np.uint8))
        return res, CResults(_p(res.tree_score, _f32p), _p(res.filter, _u8p), _p(res.flags, _u8p))

    def filter_variants(self, vt: S.VariantTable, out: S.FilterResult | None = None) -> S.FilterResult:
        """featurize -> lookup -> score -> FILTER for one table (H2D + kernel + D2H).  `out`: result arrays of the caller
        (C-contiguous f32 / u8 / u8 of vt.n rows) to fill instead of fresh ones - a caller that scores callset after
        callset keeps them: three fresh arrays are 30 MB of first-touch page faults per 5 M variants."""
        cv = self._cvariants(vt)
        if out is None:
            res, cr = self._alloc_results(vt.n)
        else:
            for a, dt in ((out.tree_score, np.float32), (out.filter, np.uint8), (out.flags, np.uint8)):
                if a.dtype != dt or a.shape != (vt.n,) or not a.flags.c_contiguous or not a.flags.writeable:
                    raise ValueError("out: need writable C-contiguous tree_score f32 / filter u8 / flags u8 arrays of vt.n rows")
            res, cr = out, CResults(_p(out.tree_score, _f32p), _p(out.filter, _u8p), _p(out.flags, _u8p))
        # (the C side leaves the context EMPTY on any failed upload / boundary call: the cached row count follows what is
        # resident on every way out, so that a later download / feature_matrix cannot size its buffers by a stale count - ADVICE r5)
        try:
            self._check(self.lib.ugvc_filter_variants(self._h, C.byref(cv), C.byref(cr)))
        finally:
            self._follow_resident_count()
        return res

    def upload_variants(self, vt: S.VariantTable):
        cv = self._cvariants(vt)
        try:
            self._check(self.lib.ugvc_variants_upload(self._h, C.byref(cv)))
        finally:
            self._follow_resident_count()

    def filter_resident(self):
        self._check(self.lib.ugvc_filter_resident(self._h))

    def download_results(self) -> S.FilterResult:
        res, cr = self._alloc_results(self.n)
        self._check(self.lib.ugvc_results_download(self._h, C.byref(cr)))
        return res

    def timed_filter(self, iters: int) -> float:
        """Total milliseconds of `iters` back-to-back kernel launches (hipEvents on the stream)."""
        ms = C.c_float()
        self._check(self.lib.ugvc_timed_filter(self._h, iters, C.byref(ms)))
        return ms.value

    def timed_steps(self, iters: int, shard_cap: int = 0, gather: bool = False, per_step_events: bool = True):
        """(ms_total, ms_kernel_sum) of `iters` steps = kernel [+ RCCL all-gather] on the stream.  per_step_events=False: one
        event pair around the whole run instead of one per step (a marker between two launches costs ~9 us: the passes then run
        back to back as a production stream issues them; ms_kernel_sum = ms_total, `last_step_ms` holds the mean)."""
        tot, ker = C.c_float(), C.c_float()
        self._check(self.lib.ugvc_set_step_events(self._h, int(per_step_events)))
        try:
            self._check(self.lib.ugvc_timed_steps(self._h, iters, shard_cap, int(gather), C.byref(tot), C.byref(ker)))
        finally:
            self.lib.ugvc_set_step_events(self._h, 1)
        return tot.value, ker.value

    def pass_clock_ghz(self, passes: int = 40):
        """(shader clock in GHz the resident scoring pass sustains, span of the probed wave in ms): `passes` passes back to back,
        the last one read by the kernel's own counters (s_memtime against the constant 100 MHz s_memrealtime)."""
        ghz, ms = C.c_double(), C.c_double()
        self._check(self.lib.ugvc_pass_clock(self._h, int(passes), C.byref(ghz), C.byref(ms)))
        return float(ghz.value), float(ms.value)

    def last_step_ms(self, n: int) -> np.ndarray:
        """Per-step kernel milliseconds of the last timed_steps call."""
        out = np.zeros(max(n, 1), np.float32)
        k = self.lib.ugvc_last_step_ms(self._h, _p(out, _f32p), int(n))
        if k < 0:
            self._check(k)
        return out[:k]

    def device_sync(self):
        self._check(self.lib.ugvc_device_sync(self._h))

    def n_features(self) -> int:
        return self.lib.ugvc_n_features(self._h)

    def feature_matrix(self, vt: S.VariantTable | None = None):
        """(X float32 [n, F], group u8 [n]) - the matrix train_models_pipeline fits on."""
        if vt is not None:
            self.upload_variants(vt)
        F = self.n_features()
        X = np.zeros((self.n, F), np.float32)
        g = np.zeros(self.n, np.uint8)
        self._check(self.lib.ugvc_feature_matrix(self._h, _p(X, _f32p), _p(g, _u8p)))
        return X, g

    def timed_feature_matrix(self, iters: int) -> float:
        """ms per build of the resident N x F feature matrix (no download)."""
        ms = C.c_float()
        self._check(self.lib.ugvc_timed_feature_matrix(self._h, int(iters), C.byref(ms)))
        return ms.value / iters

    def forest_gemm(self, group: int, rows: np.ndarray | None = None, use_mfma: bool = True, iters: int = 1):
        """(f32 margins, ms per launch) of group `group`'s additive ensemble on rows of the resident feature
        matrix: leaf-matrix GEMM on MFMA (use_mfma) or row traversal."""
        n = self.n if rows is None else int(rows.size)
        out = np.zeros(n, np.float32)
        ms = C.c_float()
        r = None if rows is None else _col(rows, np.int32)
        self._check(self.lib.ugvc_forest_gemm(self._h, group, None if r is None else _p(r, _i32p), n, int(use_mfma),
                                              iters, _p(out, _f32p), C.byref(ms)))
        return out, ms.value

    def forest_gemm3(self, rows_by_group: list, iters: int = 1):
        """(f32 margin per row of the resident feature matrix, ms per launch): every variant-type group's additive ensemble on
        its rows (rows_by_group[k]: int32 row numbers, or None / empty) in ONE launch of the leaf-matrix GEMM (round 5)."""
        keep = [None if r is None else _col(r, np.int32) for r in rows_by_group]
        ptrs = (_i32p * 3)(*[C.cast(None, _i32p) if r is None or r.size == 0 else _p(r, _i32p) for r in keep])
        counts = np.array([0 if r is None else r.size for r in keep], np.int64)
        out = np.zeros(self.n, np.float32)
        ms = C.c_float()
        self._check(self.lib.ugvc_forest_gemm3(self._h, ptrs, _p(counts, _i64p), iters, _p(out, _f32p), C.byref(ms)))
        return out, ms.value

    # ---- pileup
    @staticmethod
    def _check_csr(off, ob):
        if off.ndim != 1 or off.size < 1 or ob.ndim != 1:
            raise ValueError("offsets must be a 1-D array of n_loci + 1 entries, obs 1-D")
        if off[0] != 0 or off[-1] != ob.size or (off.size > 1 and np.any(np.diff(off) < 0)):
            raise ValueError("offsets must rise from 0 to len(obs)")

    def pileup_tally(self, offsets: np.ndarray, obs: np.ndarray) -> dict:
        off, ob = _col(offsets, np.int64), _col(obs, np.uint16)
        self._check_csr(off, ob)
        n = off.size - 1
        out = {k: np.zeros(n, np.int32) for k in ("ref_fwd", "ref_rev", "alt_fwd", "alt_rev", "other", "dp",
                                                  "bq_ref", "bq_alt")}
        out["vaf"] = np.zeros(n, np.float32)
        out["sor"] = np.zeros(n, np.float32)
        co = CPileupOut(*[_p(out[k], _i32p) for k in ("ref_fwd", "ref_rev", "alt_fwd", "alt_rev", "other", "dp",
                                                      "bq_ref", "bq_alt")],
                        _p(out["vaf"], _f32p), _p(out["sor"], _f32p))
        self._check(self.lib.ugvc_pileup_tally(self._h, _p(off, _i64p), _p(ob, _u16p), n, C.byref(co)))
        out["ad_ref"] = out["ref_fwd"] + out["ref_rev"]
        out["ad_alt"] = out["alt_fwd"] + out["alt_rev"]
        return out

    def upload_pileup(self, offsets, obs):
        off, ob = _col(offsets, np.int64), _col(obs, np.uint16)
        self._check_csr(off, ob)
        self._check(self.lib.ugvc_pileup_upload(self._h, _p(off, _i64p), _p(ob, _u16p), off.size - 1))

    def timed_pileup(self, iters: int) -> float:
        ms = C.c_float()
        self._check(self.lib.ugvc_timed_pileup(self._h, iters, C.byref(ms)))
        return ms.value

    # ---- SEC statistic
    def sec_likelihood_ratio(self, actual: np.ndarray, expected: np.ndarray):
        a, e = _col(actual, np.int32), _col(expected, np.int32)
        if a.ndim != 2 or a.shape != e.shape:
            raise ValueError("actual/expected must be [n_loci, k] of equal shape")
        lik, ratio = np.zeros(a.shape[0]), np.zeros(a.shape[0])
        self._check(self.lib.ugvc_sec_likelihood_ratio(self._h, _p(a, _i32p), _p(e, _i32p), a.shape[0], a.shape[1],
                                                       _p(lik, _f64p), _p(ratio, _f64p)))
        return lik, ratio

    # ---- SEC database (builder-defined around the in-tree statistic; include/ugvc_mi355x.h)
    def sec_db_build(self, keys: np.ndarray, counts: np.ndarray):
        """Cohort observations (u64 locus keys in any order, [n_obs, k] counts) -> (sorted unique keys, summed counts)."""
        k_ = np.ascontiguousarray(keys, np.uint64)
        c = _col(counts, np.int32)
        if c.ndim != 2 or c.shape[0] != k_.size:
            raise ValueError("counts must be [n_obs, k] with one row per key")
        ok, oe, on = np.zeros(k_.size, np.uint64), np.zeros(c.shape, np.int32), C.c_int64()
        self._check(self.lib.ugvc_sec_db_build(self._h, _p(k_, _u64p), _p(c, _i32p), k_.size, c.shape[1], _p(ok, _u64p),
                                               _p(oe, _i32p), C.byref(on)))
        return ok[:on.value].copy(), oe[:on.value].copy()

    def set_sec_db(self, keys: np.ndarray, expected: np.ndarray):
        k_ = np.ascontiguousarray(keys, np.uint64)
        e = _col(expected, np.int32)
        if e.ndim != 2 or e.shape[0] != k_.size:
            raise ValueError("expected must be [n_db, k] with one row per key")
        self._check(self.lib.ugvc_sec_db_upload(self._h, _p(k_, _u64p), _p(e, _i32p), k_.size, e.shape[1]))

    def sec_apply(self, min_ratio: float = 0.05, scale_expected: bool = True, mark: bool = False, download: bool = True):
        """(ratio f64 [n] - NaN off the database, is_sec bool [n]) for the resident variants; mark=True also sets the
        SEC bit in the resident flags column (needs a scoring pass first).  download=False leaves everything on the
        device (with mark=True the verdict is in the resident flags) and returns None."""
        if not download:
            self._check(self.lib.ugvc_sec_apply(self._h, float(min_ratio), int(scale_expected), int(mark), None, None))
            return None
        ratio, hit = np.zeros(self.n, np.float64), np.zeros(self.n, np.uint8)
        self._check(self.lib.ugvc_sec_apply(self._h, float(min_ratio), int(scale_expected), int(mark), _p(ratio, _f64p), _p(hit, _u8p)))
        return ratio, hit.astype(bool)

    # ---- calibrate_bridging_snvs
    def timed_sec_apply(self, iters: int, min_ratio: float = 0.05, scale_expected: bool = True) -> float:
        """Total milliseconds of `iters` mark-only SEC applications (device events on the context stream)."""
        ms = C.c_float()
        self._check(self.lib.ugvc_timed_sec_apply(self._h, float(min_ratio), int(scale_expected), int(iters), C.byref(ms)))
        return float(ms.value)

    def bridging_snvs(self, vt: S.VariantTable, is_pass, ad_alt_sum, bg_ad_alt_sum, bg_dp,
                      min_query_hmer_size=5, min_initial_qual=5, min_tumor_vaf=0.2, max_normal_vaf=0.1,
                      min_normal_depth=10, min_distance_from_edge=0):
        cv = self._cvariants(vt)
        ip, a, b, d = (_col(is_pass, np.uint8), _col(ad_alt_sum, np.int32), _col(bg_ad_alt_sum, np.int32),
                       _col(bg_dp, np.int32))
        for name, x in (("is_pass", ip), ("ad_alt_sum", a), ("bg_ad_alt_sum", b), ("bg_dp", d)):
            if x.shape != (vt.n,):                         # the C ABI takes bare pointers: sizes are checked here
                raise ValueError(f"{name} must have one entry per variant ({vt.n}), got shape {x.shape}")
        prm = CBridgingParams(min_initial_qual, min_tumor_vaf, max_normal_vaf, min_query_hmer_size,
                              min_normal_depth, min_distance_from_edge)
        oh, op = np.zeros(vt.n, np.uint8), np.zeros(vt.n, np.uint8)
        try:
            self._check(self.lib.ugvc_bridging_snvs(self._h, C.byref(cv), _p(ip, _u8p), _p(a, _i32p), _p(b, _i32p),
                                                    _p(d, _i32p), C.byref(prm), _p(oh, _u8p), _p(op, _u8p)))
        finally:
            self._follow_resident_count()
        return oh.astype(bool), op.astype(bool)

    # ---- multi-GPU
    def comm_unique_id(self) -> bytes:
        buf = np.zeros(128, np.uint8)
        self._check(self.lib.ugvc_comm_unique_id(_p(buf, _u8p)))
        return buf.tobytes()

    def comm_init(self, uid: bytes, rank: int, world: int):
        buf = np.frombuffer(uid, dtype=np.uint8).copy()
        self._check(self.lib.ugvc_comm_init(self._h, _p(buf, _u8p), rank, world))
        self.rank, self.world = rank, world

    def comm_info(self) -> dict:
        """What RCCL reports about this context's communicator: {"nranks", "rank", "device"}."""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self._check(self.lib.ugvc_comm_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(nranks=a.value, rank=b.value, device=c.value)

    def allgather_resident(self, shard_cap: int):
        self._check(self.lib.ugvc_allgather_resident(self._h, shard_cap))

    def gathered_download(self, shard_cap: int, world: int, counts: list) -> S.FilterResult:
        tot = shard_cap * world
        res, cr = self._alloc_results(tot)
        self._check(self.lib.ugvc_gathered_download(self._h, shard_cap, world, C.byref(cr)))
        sel = np.concatenate([np.arange(r * shard_cap, r * shard_cap + c) for r, c in enumerate(counts)])
        return S.FilterResult(res.tree_score[sel], res.filter[sel], res.flags[sel])


def configure(engine: Engine, ref, runs, tracks, blacklist, forests, flow_order="TGCA", hpol_len=10,
              hpol_dist=10, mark_hpol=True):
    """Load every resident table of one filtering job (the reference builds the same state from
    --reference_file/--runs_file/--annotate_intervals/--blacklist/--model_file)."""
    engine.set_reference(ref)
    if runs is None:                       # a context may be re-configured: no runs file = an empty runs table
        runs = S.IntervalTrack(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(ref.n_contigs + 1, np.int32), "runs")
    engine.set_runs(runs, hpol_len, hpol_dist, mark_hpol)
    engine.set_tracks(tracks or [])
    engine.set_blacklist(blacklist)
    engine.set_flow_order(flow_order)
    engine.set_models(forests)
    return engine
