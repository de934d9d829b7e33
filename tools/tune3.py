"""Launch-shape sweep of the v3 scoring pass (profiling aid).  Kernel-variant bits: 1 no forest kernel, 32 K1 without column prefetch, 2048 16 trees in flight,
1024 pair-sum forest kernel, bits 12-13 K1 workgroups per CU (0 = 4), bits 14-15 K2 waves (0 = 16, 1 = 12,
2 = 8, 3 = 4).  Usage: python tools/tune3.py [n_variants ...]"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from variantcalling_amd import model_io, synth  # noqa: E402
from variantcalling_amd.engine import Engine, configure  # noqa: E402

sizes = [int(x) for x in sys.argv[1:]] or [5_000_000, 625_000]
forests = model_io.load_models(os.path.join(ROOT, "tests", "golden", "synth_rf_v1.npz"))[
    "rf_model_ignore_gt_incl_hpol_runs"]
eng = Engine(0)
full = synth.make_callset(max(sizes))
configure(eng, full.ref, full.runs, full.tracks, full.blacklist, forests)


def t(v, iters=10):
    eng.set_kernel_variant(v)
    eng.timed_filter(3)
    return min(eng.timed_filter(iters) / iters for _ in range(3)) * 1e3


for n in sizes:
    vt = full.variants if n >= full.variants.n else full.variants.slice(0, n)
    eng.upload_variants(vt)
    print(f"== {vt.n} variants")
    base = t(0)
    k1 = t(1)
    print(f"single-sum forest: pass {base:8.1f} us   K0+K1 {k1:8.1f} us   K2 {base - k1:8.1f} us", flush=True)
    pair = t(1024)
    print(f"pair-sum forest:   pass {pair:8.1f} us   K2 {pair - k1:8.1f} us", flush=True)
    nt8 = t(2048)
    print(f"16 trees in flight: pass {nt8:8.1f} us   K2 {nt8 - k1:8.1f} us", flush=True)
    nopf = t(1 | 32)
    print(f"K1 without column prefetch: K0+K1 {nopf:8.1f} us", flush=True)
    for bpc in (1, 2, 3):
        print(f"K1 {bpc} workgroups/CU: K0+K1 {t(1 | (bpc << 12)):8.1f} us", flush=True)
    for pick, w in ((1, 12), (2, 8), (3, 4)):
        x = t(pick << 14)
        print(f"K2 {w:2d} waves: pass {x:8.1f} us   K2 {x - k1:8.1f} us", flush=True)
eng.set_kernel_variant(0)
eng.close()
