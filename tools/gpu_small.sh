#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== bench"; timeout 600 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_r01_final.json | cut -c1-200
echo "== profile"; bash tools/gpu_profile.sh r01_final 2>&1 | tail -22
echo "== aux"; timeout 300 python tools/bench_aux.py 2>&1 | tail -3 | tee gpurun_out/bench_aux.log
