#!/bin/bash
# Round-end validation: full GPU suite, smoke, default bench line, rocprofv3 kernel trace + FETCH/WRITE passes.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=${1:-r01_final}
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/pytest_gpu_$TAG.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke_$TAG.log
echo "== bench"; timeout 600 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_$TAG.json
echo "== profile"; bash tools/gpu_profile.sh $TAG 2>&1 | tail -25
