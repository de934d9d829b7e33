"""Section timing of the v3 scoring pass on the GPU (profiling aid; results are WRONG under ablation).
bits: 1 no forest kernel, 2 no joins/staging, 4 no quantisation, 8 no reference-window features, 16 no record append.
Usage: python tools/ablate3.py [n_variants]"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from variantcalling_amd import model_io, synth  # noqa: E402
from variantcalling_amd.engine import Engine, configure  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
base = int(sys.argv[2]) if len(sys.argv) > 2 else 0        # extra kernel-variant bits, e.g. 128 = the v4 featurize kernel
cs = synth.make_callset(n)
forests = model_io.load_models(os.path.join(ROOT, "tests", "golden", "synth_rf_v1.npz"))[
    "rf_model_ignore_gt_incl_hpol_runs"]
eng = Engine(0)
configure(eng, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
eng.upload_variants(cs.variants)
for v, name in ((0, "full"), (1, "K0+K1 only"), (1 | 2, "K1 - joins"), (1 | 4, "K1 - quantise"), (1 | 8, "K1 - window features"),
                (1 | 16, "K1 - append"), (1 | 2 | 4, "K1 - joins - quantise"), (1 | 2 | 4 | 8, "K1 - joins - quantise - window"),
                (1 | 2 | 4 | 8 | 16, "K1 columns + window load only")):
    eng.set_kernel_variant(v | base)
    eng.timed_filter(3)
    ms = min(eng.timed_filter(10) / 10 for _ in range(3))
    print(f"ablate={v:3d} {name:34s} {ms * 1e3:9.1f} us", flush=True)
