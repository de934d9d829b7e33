#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== quick parity"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not full_size" 2>&1 | tail -8 | tee gpurun_out/pytest_gpu_quick.log
echo "== profile"; bash tools/gpu_profile.sh r01_v3a 2>&1 | tail -40
