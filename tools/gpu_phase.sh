#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== parity"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ties or synthetic_rf" 2>&1 | tail -4
echo "== phases"; timeout 600 python tools/phase3.py 2>&1 | tail -30 | tee gpurun_out/phase3.log
echo "== tune"; timeout 600 python tools/tune3.py 625000 2>&1 | head -5 | tee gpurun_out/tune3_625k.log
