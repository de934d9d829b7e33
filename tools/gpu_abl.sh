#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
echo "== parity v4"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "v4 and not full_size" 2>&1 | tail -4
echo "== v4"; timeout 300 python tools/ablate3.py 5000000 128 2>&1 | tail -10 | tee gpurun_out/ablate_v4.log
