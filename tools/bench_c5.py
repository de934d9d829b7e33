"""Config C5 measurement: 2 M labelled variants - on-GPU feature matrix build, then the T = 100, depth-6
additive ensemble evaluated as a leaf-matrix GEMM on MFMA next to the row traversal.
Usage: python tools/bench_c5.py [n_variants]"""
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from variantcalling_amd import model_io, synth  # noqa: E402
from variantcalling_amd.engine import Engine, configure  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
cs = synth.make_callset(n)
forests = model_io.load_models(os.path.join(ROOT, "tests", "golden", "synth_rf_v1.npz"))["xgb_model_ignore_gt_incl_hpol_runs"]
eng = Engine(0)
configure(eng, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
eng.upload_variants(cs.variants)
t0 = time.perf_counter()
X, group = eng.feature_matrix()
t_fm = time.perf_counter() - t0
N, F = X.shape
out = {"n": N, "F": F, "feature_matrix_wall_s_incl_d2h": round(t_fm, 4)}
same = True
tot = {"gemm": 0.0, "traverse": 0.0}
for g in range(3):
    rows = np.flatnonzero(group == g).astype(np.int32)
    a, ms_a = eng.forest_gemm(g, rows, use_mfma=True, iters=5)
    b, ms_b = eng.forest_gemm(g, rows, use_mfma=False, iters=5)
    same &= bool(np.array_equal(a, b))
    tot["gemm"] += ms_a
    tot["traverse"] += ms_b
T, I, L = 100, 64, 64
ops = 2.0 * I * L * T * N
out.update(gemm_ms=round(tot["gemm"], 3), traverse_ms=round(tot["traverse"], 3), identical=same,
           gemm_int8_TOPs=round(ops / (tot["gemm"] * 1e-3) / 1e12, 3), mfma_i8_peak_TOPs=5000.0,
           gemm_variants_per_s=round(N / (tot["gemm"] * 1e-3)), traverse_variants_per_s=round(N / (tot["traverse"] * 1e-3)))
print(json.dumps(out))
