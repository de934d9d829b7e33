#!/bin/bash
# GPU box: v3 parity (small tests first), path timing, then the full-size test.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== quick parity"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not full_size" 2>&1 | tail -12 | tee gpurun_out/pytest_gpu_quick.log
echo "== compare"; timeout 600 python tools/compare_paths.py 2>&1 | tail -8 | tee gpurun_out/compare_paths.log
echo "== full size"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size" 2>&1 | tail -8 | tee gpurun_out/pytest_gpu_full.log
if [ "${1:-}" != "" ]; then bash tools/gpu_profile.sh $1 2>&1 | tail -22; fi
