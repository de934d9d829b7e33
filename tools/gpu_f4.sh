#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ties or synthetic_rf or golden or edge_cases or gbt or snv_only or fallbacks or ragged" 2>&1 | tail -8 | tee gpurun_out/pytest_f4.log
echo "== tune"; timeout 600 python tools/tune3.py 2>&1 | tail -30 | tee gpurun_out/tune3.log
