"""CPU tests of the host-side logic and of the C-ABI library as an artefact (no compute calls:
there is no GPU here).  The library must load, export every symbol include/ugvc_mi355x.h
declares, and FAIL LOUDLY - not fall back - when asked for a device that is not there."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from oracle import oracle as O
from variantcalling_amd import engine as eng_mod
from variantcalling_amd import model_io, schema as S, shard, synth

HEADER = os.path.join(ROOT, "include", "ugvc_mi355x.h")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(eng_mod.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "variantcalling_amd", "csrc"), "-j4"], check=True)
    return eng_mod.load_library()


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ugvc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(lib):
    names = _declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ugvc_mi355x.h but not exported"
    # the ctypes binding table covers exactly the header
    assert sorted(eng_mod.ABI) == names
    assert lib.ugvc_abi_version() == 2
    out = subprocess.run(["nm", "-D", "--defined-only", eng_mod.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (ugvc_\w+)", out))
    assert set(names) <= exported


def test_library_is_gfx950_only(lib):
    """One code object, gfx950: no multi-arch fat binary, no host fallback kernels."""
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o",
                          f"--input={eng_mod.LIB_PATH}"], capture_output=True, text=True)
    if out.returncode == 0 and out.stdout.strip():
        targets = [t for t in out.stdout.split() if "amdgcn" in t]
        assert targets and all("gfx950" in t for t in targets), targets
    else:                                  # bundle section not listable on a .so: look for the ISA name
        blob = open(eng_mod.LIB_PATH, "rb").read()
        # offload-bundle entry ids name the ISA of every embedded code object (rocPRIM's host-side target-name
        # table mentions other ISAs as plain strings; those are not code objects)
        isas = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", blob))
        assert isas == {b"gfx950"}, isas


def test_no_gpu_fails_loudly(lib):
    """No silent CPU path: creating a context without a device is an error with a message."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    h = C.c_void_p()
    rc = lib.ugvc_ctx_create(0, C.byref(h))
    assert rc != 0 and lib.ugvc_last_error()
    with pytest.raises(RuntimeError):
        eng_mod.Engine(0)
    assert lib.ugvc_filter_resident(None) != 0 and b"NULL" in lib.ugvc_last_error()


def test_product_path_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under variantcalling_amd/ may import it."""
    pkg = os.path.join(ROOT, "variantcalling_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(d, f)
                assert "from oracle" not in src and "import oracle" not in src, os.path.join(d, f)
    code = ("import sys; sys.path.insert(0, %r); import variantcalling_amd.engine, variantcalling_amd.dist, "
            "variantcalling_amd.model_io, variantcalling_amd.synth; "
            "assert not [m for m in sys.modules if m == 'oracle' or m.startswith('oracle.')]" % ROOT)
    subprocess.run([sys.executable, "-c", code], check=True)


# ------------------------------------------------------------------ cycle-skip LUT (host helper of the kernels)
@pytest.mark.parametrize("flow", ["TGCA", "ACGT", "GTAC", "CATG"])
def test_css_lut_matches_full_flow_key_computation(lib, flow):
    """The kernels use a 256-entry table for single-base substitutions; it must equal the oracle's
    full 11-mer flow-key comparison for EVERY context (exhaustive over l1, ref, alt, r1 and random
    over the outer motif bases, which must not matter)."""
    lut = np.zeros(256, np.uint8)
    assert lib.ugvc_host_css_lut(flow.encode(), lut.ctypes.data_as(C.POINTER(C.c_uint8))) == 0
    fo = S.encode_bases(flow)
    rng = np.random.default_rng(1)
    for l1 in range(4):
        for r in range(4):
            for a in range(4):
                if a == r:
                    continue
                for r1 in range(4):
                    want = None
                    for _ in range(6):
                        left = np.concatenate([rng.integers(1, 5, 4), [l1 + 1]]).astype(np.uint8)
                        right = np.concatenate([[r1 + 1], rng.integers(1, 5, 4)]).astype(np.uint8)
                        rs = np.concatenate([left, [r + 1], right]).astype(np.uint8)
                        as_ = np.concatenate([left, [a + 1], right]).astype(np.uint8)
                        st = O.cycle_skip_status(rs, as_, fo)
                        assert want is None or st == want, "outer motif bases changed the status"
                        want = st
                    assert lut[(l1 << 6) | (r << 4) | (a << 2) | r1] == want
    for bad in (b"AAGT", b"AXGT", b"ACG", b"ACGTA"):
        assert lib.ugvc_host_css_lut(bad, lut.ctypes.data_as(C.POINTER(C.c_uint8))) != 0
        assert b"permutation" in lib.ugvc_last_error()


# ------------------------------------------------------------------ sharding (multi-GPU partition)
def test_shard_bounds_and_reassembly():
    for n, w in ((0, 1), (1, 8), (7, 8), (8, 8), (5_000_000, 8), (1_000_003, 4), (13, 2)):
        b = shard.shard_bounds(n, w)
        sizes = np.diff(b)
        assert b[0] == 0 and b[-1] == n and sizes.max() - sizes.min() <= 1 and (sizes >= 0).all()
        cap = shard.shard_cap(n, w)
        assert cap >= max(int(sizes.max()), 1) and cap % 256 == 0 and cap - int(sizes.max()) <= 256
    with pytest.raises(ValueError):
        shard.shard_bounds(10, 0)
    cs = synth.make_callset(5000, genome_len=2_000_000, n_contigs=3, seed=3)
    vt = cs.variants
    parts = [shard.shard_of(vt, r, 3) for r in range(3)]
    assert sum(p.n for p in parts) == vt.n
    for p in parts:
        p.validate()
    assert np.array_equal(np.concatenate([p.pos for p in parts]), vt.pos)
    # allele pools are re-based: every shard decodes the same alleles as the full table
    lo = 0
    for p in parts:
        for k in (0, p.n // 2, p.n - 1):
            i = lo + k
            assert np.array_equal(p.alleles[p.alt_off[k]: p.alt_off[k] + p.alt_len[k]],
                                  vt.alleles[vt.alt_off[i]: vt.alt_off[i] + vt.alt_len[i]])
        lo += p.n
    rng = np.random.default_rng(0)
    full = S.FilterResult(rng.random(vt.n).astype(np.float32), rng.integers(0, 2, vt.n).astype(np.uint8),
                          rng.integers(0, 255, vt.n).astype(np.uint8))
    b = shard.shard_bounds(vt.n, 3)
    cap = shard.shard_cap(vt.n, 3)
    padded = [shard.pad_result(S.FilterResult(full.tree_score[b[r]:b[r + 1]], full.filter[b[r]:b[r + 1]],
                                              full.flags[b[r]:b[r + 1]]), cap) for r in range(3)]
    back = shard.reassemble(padded, [int(b[r + 1] - b[r]) for r in range(3)])
    assert np.array_equal(back.tree_score, full.tree_score) and np.array_equal(back.flags, full.flags)


# ------------------------------------------------------------------ model import
def test_f32_threshold_rounding_decides_like_f64():
    rng = np.random.default_rng(2)
    thr = np.concatenate([rng.normal(size=2000), rng.normal(size=2000).astype(np.float32).astype(np.float64),
                          [0.0, -0.0, 1e-40, 3.4e38, 0.1, 0.5]])
    fl, ce = model_io.f32_floor(thr), model_io.f32_ceil(thr)
    assert (fl.astype(np.float64) <= thr).all() and (ce.astype(np.float64) >= thr).all()
    x = np.concatenate([fl, ce, np.nextafter(fl, np.float32(np.inf)), np.nextafter(ce, np.float32(-np.inf))])
    for t, f, c in zip(thr[::37], fl[::37], ce[::37]):
        assert np.array_equal(x.astype(np.float64) <= t, x <= f)
        assert np.array_equal(x.astype(np.float64) < t, x < c)


def test_model_file_round_trip(tmp_path, frozen_models):
    import pickle

    from sklearn.ensemble import RandomForestClassifier
    p = tmp_path / "m.npz"
    model_io.save_models(str(p), frozen_models)
    back = model_io.load_models(str(p))
    for name in frozen_models:
        for a, b in zip(frozen_models[name], back[name]):
            for k in ("feature", "threshold", "left", "right", "tree_root", "leaf_value"):
                assert np.array_equal(getattr(a, k), getattr(b, k))
            assert (a.kind, a.n_features, a.max_depth, a.base_score) == (b.kind, b.n_features, b.max_depth, b.base_score)
    rng = np.random.default_rng(0)
    X = rng.normal(size=(500, 20)).astype(np.float32)
    y = (X[:, 0] > 0).astype(int)
    clfs = {g: RandomForestClassifier(n_estimators=3, max_depth=3, random_state=i).fit(X, y)
            for i, g in enumerate(S.GROUP_NAMES)}
    pk = tmp_path / "m.pkl"
    pk.write_bytes(pickle.dumps({"rf_model_ignore_gt_incl_hpol_runs": clfs}))
    forests = model_io.load_model_file(str(pk), "rf_model_ignore_gt_incl_hpol_runs")
    assert len(forests) == 3 and all(f.kind == S.MODEL_RF and f.n_trees == 3 for f in forests)
    with pytest.raises(KeyError):
        model_io.load_model_file(str(pk), "nope")


def test_synthetic_callset_is_deterministic_and_wgs_shaped():
    a = synth.make_callset(20_000, genome_len=12_000_000, n_contigs=4, seed=5)
    b = synth.make_callset(20_000, genome_len=12_000_000, n_contigs=4, seed=5)
    for c in S.VariantTable.COLS:
        assert np.array_equal(getattr(a.variants, c), getattr(b.variants, c)), c
    assert np.array_equal(a.blacklist, b.blacklist) and np.array_equal(a.ref.codes, b.ref.codes)
    vt = a.variants
    vt.validate()
    indel = vt.ref_len != vt.alt_len
    assert 0.15 < indel.mean() < 0.21
    ft = O.featurize(vt, a.ref, a.runs, a.tracks)
    assert 0.45 < (ft["group"][indel] == S.GROUP_HINDEL).mean() < 0.75
    for t in a.tracks + [a.runs]:
        assert (t.starts < t.ends).all() and t.contig_ptr[-1] == t.starts.size
        for c in range(a.ref.n_contigs):
            s, e = t.starts[t.contig_ptr[c]: t.contig_ptr[c + 1]], t.ends[t.contig_ptr[c]: t.contig_ptr[c + 1]]
            assert (np.diff(s) > 0).all() and (e[:-1] <= s[1:]).all()
    assert np.isin(vt.keys(), a.blacklist).sum() > 0
    snv = synth.make_callset(5000, genome_len=3_000_000, n_contigs=2, seed=5, snv_only=True).variants
    assert (snv.ref_len == 1).all() and (snv.alt_len == 1).all()


def test_variant_table_validation():
    cs = synth.make_callset(300, genome_len=2_000_000, n_contigs=2, seed=9)
    import copy
    vt = copy.copy(cs.variants)
    vt.pos = vt.pos[::-1].copy()
    with pytest.raises(ValueError, match="sorted"):
        vt.validate()
    vt = copy.copy(cs.variants)
    vt.qual = vt.qual.astype(np.float64)
    with pytest.raises(ValueError, match="qual"):
        vt.validate()
    assert cs.variants.slice(10, 10).n == 0


def test_set_model_rejects_malformed_tables_before_the_abi():
    """The ABI takes bare pointers + one node count: mismatched node arrays or a leaf table that is not [n, 2] would be read
    out of bounds on the host side of the library.  The binding refuses them (no GPU needed: the check precedes the call)."""
    import copy
    from variantcalling_amd import model_io
    from variantcalling_amd.engine import Engine
    f = model_io.load_models(os.path.join(ROOT, "tests", "golden", "synth_rf_v1.npz"))["rf_model_ignore_gt_incl_hpol_runs"][0]
    eng = Engine.__new__(Engine)                        # no context: set_model must fail before touching the library
    eng.lib, eng._h = None, None
    bad = copy.copy(f)
    bad.threshold = f.threshold[:-1]
    with pytest.raises(ValueError, match="differ in length"):
        eng.set_model(0, bad)
    bad = copy.copy(f)
    bad.leaf_value = f.leaf_value[:, 0]
    with pytest.raises(ValueError, match=r"\[n_leaves, 2\]"):
        eng.set_model(0, bad)
    bad = copy.copy(f)
    bad.leaf_value = f.leaf_value[:, :1]
    with pytest.raises(ValueError, match=r"\[n_leaves, 2\]"):
        eng.set_model(0, bad)


def test_frame_to_table_rejects_alleles_longer_than_u16():
    from variantcalling_amd.io import concordance, h5
    fr = h5.Frame([("chrom", np.array(["chr1", "chr1"], object)), ("pos", np.array([10, 20])),
                   ("ref", np.array(["A", "A" * 70000], object)),
                   ("alleles", np.array([("A", "C"), ("A" * 70000, "A")], object))])
    with pytest.raises(ValueError, match="longer than 65535"):
        concordance.frame_to_table(fr, ["chr1"])


def test_host_row_checks_name_the_first_offending_row(lib):
    """csrc/host_rows.cpp: the vectorised block check + scalar re-read behind ugvc_variants_upload and the chunk pipeline
    (internal C++ symbols, no GPU needed) against a row-by-row numpy restatement: clean tables, one planted fault of each
    kind at block edges (block = 8192 rows), several faults (the FIRST is reported), and piece starts in the middle of a
    table (row lo is compared with row lo - 1); the streaming copy equals memcpy at odd sizes and alignments."""
    import ctypes as C
    from variantcalling_amd import engine as EN
    vr = getattr(lib, "_ZN4ugvc13validate_rowsEPK13ugvc_variantslliPlS3_")
    vr.restype = C.c_int
    vr.argtypes = [C.POINTER(EN.CVariants), C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    rng = np.random.default_rng(5)
    n, n_contigs = 40_000, 7
    contig = np.sort(rng.integers(0, n_contigs, n)).astype(np.uint16)
    pos = np.zeros(n, np.int32)
    for c in range(n_contigs):
        m = contig == c
        pos[m] = np.sort(rng.integers(1, 5_000_000, int(m.sum())))
    rl = rng.integers(1, 4, n).astype(np.uint16)
    al = rng.integers(1, 4, n).astype(np.uint16)
    ro = np.cumsum(np.r_[0, (rl + al)[:-1]]).astype(np.uint32)
    ao = (ro + rl).astype(np.uint32)
    alleles_len = int(ro[-1]) + int(rl[-1]) + int(al[-1])

    def run(cols, lo, hi):
        ct, ps, r, a, rof, aof = (np.ascontiguousarray(x) for x in cols)
        z8, zf, zi = np.zeros(1, np.uint8), np.zeros(1, np.float32), np.zeros(1, np.int32)
        cv = EN.CVariants(n, EN._p(ct, EN._u16p), EN._p(ps, EN._i32p), EN._p(r, EN._u16p), EN._p(a, EN._u16p), EN._p(rof, EN._u32p),
                          EN._p(aof, EN._u32p), EN._p(z8, EN._u8p), alleles_len, EN._p(zf, EN._f32p), EN._p(zf, EN._f32p),
                          EN._p(zi, EN._i32p), EN._p(zi, EN._i32p), EN._p(zi, EN._i32p), EN._p(z8, EN._u8p))
        k, row = C.c_int64(0), C.c_int64(-1)
        what = vr(C.byref(cv), lo, hi, n_contigs, C.byref(k), C.byref(row))
        return what, row.value, k.value

    def expect(cols, lo, hi):
        ct, ps, r, a, rof, aof = cols
        for i in range(lo, hi):
            if ct[i] >= n_contigs: return 1, i
            if r[i] == 0 or a[i] == 0: return 2, i
            if int(rof[i]) + int(r[i]) > alleles_len or int(aof[i]) + int(a[i]) > alleles_len: return 3, i
            if ps[i] < 1: return 4, i
            if i and (ct[i] < ct[i - 1] or (ct[i] == ct[i - 1] and ps[i] < ps[i - 1])): return 5, i
        return 0, -1

    base = (contig, pos, rl, al, ro, ao)
    for lo, hi in ((0, n), (1, n), (8191, 8193), (12_345, 33_333), (n - 1, n), (5, 5)):
        what, row, k = run(base, lo, hi)
        assert (what, row) == (0, -1) and k == int((rl[lo:hi] != al[lo:hi]).sum()), (lo, hi)
    plant = {1: lambda c, i: c[0].__setitem__(i, 200), 2: lambda c, i: c[2].__setitem__(i, 0), 3: lambda c, i: c[5].__setitem__(i, 2**32 - 1),
             4: lambda c, i: c[1].__setitem__(i, 0), 5: lambda c, i: c[1].__setitem__(i, max(int(c[1][i - 1]) - 1, 1))}
    for what_planted, f in plant.items():
        for i in (0, 1, 8191, 8192, 8193, 16_384, 29_999, n - 1):
            if what_planted == 5 and (i == 0 or contig[i] != contig[i - 1] or pos[i - 1] <= 1):
                continue
            cols = tuple(x.copy() for x in base)
            f(cols, i)
            for lo, hi in ((0, n), (max(i - 3, 0), min(i + 3, n)), (i, i + 1), (min(i + 1, n), n)):
                got = run(cols, lo, hi)[:2]
                assert got == expect(cols, lo, hi), (what_planted, i, lo, hi, got)
    cols = tuple(x.copy() for x in base)
    for i in (30_000, 9_000, 25_000):
        cols[2][i] = 0
    cols[1][9_500] = -4
    assert run(cols, 0, n)[:2] == (2, 9_000) and run(cols, 9_001, n)[:2] == (4, 9_500)
    cs = getattr(lib, "_ZN4ugvc11copy_streamEPvPKvm")
    cs.restype = None
    cs.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    src = rng.integers(0, 256, 300_000).astype(np.uint8)
    for ln in (0, 1, 4095, 4096, 4097, 70_001, 262_144):
        for da in (0, 1, 33, 63, 64):
            for sa in (0, 5):
                dst = np.full(ln + 200, 0xEE, np.uint8)
                cs(dst.ctypes.data + 100 + da, src.ctypes.data + sa, ln)
                assert np.array_equal(dst[100 + da:100 + da + ln], src[sa:sa + ln]) and dst[99 + da] == 0xEE and dst[100 + da + ln] == 0xEE


def test_abort_names_the_last_launches_and_keeps_pythons_traceback():
    """What a GPU memory fault leaves behind (the HSA runtime aborts the process): with UGVC_BREADCRUMB=1 the library's SIGABRT
    handler prints the last kernel launches and then hands over to the handler that was there before - Python's faulthandler
    under pytest, i.e. the traceback of the test.  No GPU needed: the launch bookkeeping is called directly, then abort()."""
    code = r'''
import os, sys, faulthandler, ctypes
os.environ["UGVC_BREADCRUMB"] = "1"
faulthandler.enable()
sys.path.insert(0, %r)
from variantcalling_amd import engine
lib = engine.load_library()
note = getattr(lib, "_ZN4ugvc11launch_noteEPKcP12ihipStream_t")
note.argtypes = [ctypes.c_char_p, ctypes.c_void_p]
note.restype = None
names = [ctypes.c_char_p(("kernel_%%d" %% k).encode()) for k in range(11)]
for n in names:
    note(n, None)
ctypes.CDLL(None).abort()
''' % ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("UGVC_POISON", "UGVC_DEBUG_SYNC", "UGVC_GUARD")}
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode == -6, (p.returncode, p.stderr[-500:])
    err = p.stderr
    assert "[ugvc] process aborted; last kernel launches (oldest first):" in err
    crumbs = [l.split()[-1] for l in err.splitlines() if l.startswith("[ugvc]   kernel_")]
    assert crumbs == [f"kernel_{k}" for k in range(3, 11)]                  # the ring keeps the last eight, oldest first
    assert err.index("[ugvc] process aborted") < err.index("Fatal Python error: Aborted")


def test_gemm_predicate_listing_check(tmp_path):
    """tools/isa/check_pred_asm.py (run by csrc/Makefile before the library is linked): a listing whose register-indexed predicates
    have `x0` off the first register of the feature tuple - what a compiler update could produce - fails the build."""
    import subprocess
    import sys
    script = os.path.join(ROOT, "tools", "isa", "check_pred_asm.py")
    good = "\n".join(f"\t; ugvc_pred x0=v{a} xv=v[{a}:{a + 31}]\n\ts_set_gpr_idx_on s4, gpr_idx(SRC0)" for a in (12, 12, 40))
    p = tmp_path / "good.s"
    p.write_text(good)
    assert subprocess.run([sys.executable, script, str(p)], capture_output=True).returncode == 0
    for bad in (good + "\n\t; ugvc_pred x0=v13 xv=v[12:43]\n",        # x0 is the tuple's SECOND register
                good + "\n\t; ugvc_pred x0=v12 xv=v[12:41]\n",        # a 30-register tuple
                good + "\n\t; ugvc_pred x0=v12 xv=s[12:43]\n",        # not a VGPR tuple
                "\tv_mov_b32 v0, v1\n"):                              # no instance at all: the check did not see the kernels
        p = tmp_path / "bad.s"
        p.write_text(bad)
        assert subprocess.run([sys.executable, script, str(p)], capture_output=True).returncode != 0, bad[-60:]
