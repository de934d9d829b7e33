#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python bench.py --variants 1000000 --snv-only --steps 50 --warmup 5 2>&1 | tail -1 | tee gpurun_out/bench_c2.json | cut -c1-400
timeout 600 python tools/bench_c5.py 2>&1 | tail -6 | tee gpurun_out/bench_c5.log
