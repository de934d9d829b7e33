"""Kernel-phase ablation on the GPU (profiling aid): times the fused kernel with phases switched
off through the debug `ablate` bits.  Usage: python tools/ablate.py [n_variants] [variants...]"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from variantcalling_amd import model_io, synth  # noqa: E402
from variantcalling_amd.engine import Engine, configure  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
cs = synth.make_callset(n)
forests = model_io.load_models(os.path.join(ROOT, "tests", "golden", "synth_rf_v1.npz"))[
    "rf_model_ignore_gt_incl_hpol_runs"]
eng = Engine(0)
configure(eng, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
eng.upload_variants(cs.variants)
names = {0: "full", 1: "no forest", 2: "no joins", 4: "no cycle-skip", 8: "no ref features", 3: "no forest+joins",
         7: "no forest+joins+css", 15: "columns only", 14: "forest only (+cols)"}
sel = [int(x) for x in sys.argv[2:]] or list(names)
for v in sel:
    eng.set_kernel_variant(v)
    eng.timed_filter(3)
    ms = min(eng.timed_filter(10) / 10 for _ in range(3))
    print(f"ablate={v:3d} {names.get(v, ''):24s} {ms * 1e3:9.1f} us   {cs.variants.n / (ms * 1e-3) / 1e9:7.2f} Gvar/s   "
          f"{121.6 * cs.variants.n / (ms * 1e-3) / 1e9:8.1f} GB/s(alg)")
