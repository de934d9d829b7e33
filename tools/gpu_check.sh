#!/bin/bash
# Runs on the GPU box via gpurun: parity tests, smoke, a bench line; logs under gpurun_out/.
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 3 2>&1 | tail -5 | tee gpurun_out/bench.log
