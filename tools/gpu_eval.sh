#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_eval.py tests/test_gpu_pipelines.py -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_eval.log
timeout 300 python - <<'PY' 2>&1 | tee gpurun_out/bench_eval.log
import sys, time, numpy as np
sys.path.insert(0, '.')
from variantcalling_amd import evaluate
from variantcalling_amd.engine import Engine
n = 5_000_000
rng = np.random.default_rng(1)
tp = rng.random(n) < 0.6; fp = ~tp
s = (rng.integers(0, 2**20, n) / 2**20 * (0.4 + 0.6 * tp)).astype(np.float32).astype(np.float64)
cls = np.where(tp, 1, 2).astype(np.uint8)
eng = Engine(0)
eng.pr_curve(s[:1000], cls[:1000], 1, 1, 0)
t0 = time.perf_counter(); out = eng.pr_curve(s, cls, int(tp.sum()), int(fp.sum()), 0); t_gpu = time.perf_counter() - t0
t0 = time.perf_counter()
order = np.argsort(s, kind="stable"); ctp = np.cumsum(tp[order]); cfp = np.cumsum(fp[order])
rec = evaluate.get_recall(ctp, tp.sum() - ctp, np.nan); prec = evaluate.get_precision(fp.sum() - cfp, tp.sum() - ctp, np.nan); f1 = evaluate.get_f1(prec, rec)
t_cpu = time.perf_counter() - t0
print(f"pr_curve 5M rows: device {out[5]:.2f} ms, call incl. transfers {t_gpu*1e3:.1f} ms, host numpy {t_cpu*1e3:.1f} ms; equal: "
      f"{np.array_equal(out[1], rec, equal_nan=True) and np.array_equal(out[2], prec, equal_nan=True) and np.array_equal(out[3], f1, equal_nan=True)}")
PY
