#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
echo "== tune"; timeout 600 python tools/tune3.py 5000000 625000 2>&1 | grep -E "==|single-sum|pair-sum|without" | tee gpurun_out/tune7.log
