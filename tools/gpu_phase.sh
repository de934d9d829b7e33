#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== phases (v4)"; timeout 600 python tools/phase3.py 2>&1 | tail -30 | tee gpurun_out/phase4.log
