#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python tools/bench_pipeline.py 1000000 2>&1 | grep -v "^INFO\|^DEBUG" | tail -12 | tee gpurun_out/bench_pipeline.log
