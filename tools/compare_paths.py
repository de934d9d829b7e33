"""Times the three scoring paths (v3 / v2 / v1) on the C3 callset (profiling aid).
Usage: python tools/compare_paths.py [n_variants]"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from variantcalling_amd import model_io, synth  # noqa: E402
from variantcalling_amd.engine import Engine, configure  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
cs = synth.make_callset(n)
forests = model_io.load_models(os.path.join(ROOT, "tests", "golden", "synth_rf_v1.npz"))[
    "rf_model_ignore_gt_incl_hpol_runs"]
eng = Engine(0)
configure(eng, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
eng.upload_variants(cs.variants)
ref = None
for v, name in ((0, "v3"), (512, "v2"), (256, "v1")):
    eng.set_kernel_variant(v)
    eng.timed_filter(3)
    ms = min(eng.timed_filter(10) / 10 for _ in range(3))
    res = eng.download_results()
    same = "" if ref is None else f" identical_to_v3={bool(np.array_equal(res.filter, ref.filter) and np.array_equal(res.tree_score, ref.tree_score) and np.array_equal(res.flags, ref.flags))}"
    ref = ref or res
    print(f"{name}: {ms * 1e3:9.1f} us/pass   {cs.variants.n / (ms * 1e-3) / 1e9:7.2f} Gvar/s   "
          f"{121.6 * cs.variants.n / (ms * 1e-3) / 1e9:8.1f} GB/s(alg)  frac={121.6 * cs.variants.n / (ms * 1e-3) / 8e12:.3f}{same}", flush=True)
