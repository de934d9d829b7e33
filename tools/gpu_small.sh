#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "synthetic_rf or ties or random_models or golden" 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 300 python - <<'PY'
import sys
sys.path.insert(0, '.')
from variantcalling_amd import model_io, synth
from variantcalling_amd.engine import Engine, configure
cs = synth.make_callset(5_000_000)
forests = model_io.load_models("tests/golden/synth_rf_v1.npz")["rf_model_ignore_gt_incl_hpol_runs"]
eng = Engine(0); configure(eng, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests); eng.upload_variants(cs.variants)
def t(v):
    eng.set_kernel_variant(v); eng.timed_filter(3)
    return min(eng.timed_filter(10) / 10 for _ in range(3)) * 1e3
k1 = t(1)
for v, nm in ((0, "top levels by readlane"), (131072, "all levels from LDS")):
    x = t(v)
    print(f"{nm:28s} pass {x:7.1f} us  K2 {x - k1:7.1f} us")
PY
