#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not full_size and not c5 and not rccl and not pileup and not sec and not bridging" 2>&1 | tail -15 | tee gpurun_out/pytest_v4.log
echo "== tune"; timeout 600 python tools/tune3.py 5000000 2>&1 | head -12 | tee gpurun_out/tune4.log
echo "== v3 for comparison"; timeout 300 python - <<'PY'
import os, sys
sys.path.insert(0, '.')
from variantcalling_amd import model_io, synth
from variantcalling_amd.engine import Engine, configure
cs = synth.make_callset(5_000_000)
forests = model_io.load_models("tests/golden/synth_rf_v1.npz")["rf_model_ignore_gt_incl_hpol_runs"]
eng = Engine(0); configure(eng, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests); eng.upload_variants(cs.variants)
for v, nm in ((1, "v4 K0+K1"), (1 | 128, "v3 K0+K1"), (0, "v4 pass"), (128, "v3 pass")):
    eng.set_kernel_variant(v); eng.timed_filter(3)
    print(nm, min(eng.timed_filter(10) / 10 for _ in range(3)) * 1e3, "us")
PY
