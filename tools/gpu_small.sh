#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
free -g | head -2
( time timeout 900 python bench.py --variants 50000000 --steps 10 --warmup 2 --cpu-sample 0 ) 2>&1 | tail -5 | tee gpurun_out/bench_50M.log | cut -c1-900
