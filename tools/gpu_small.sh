#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "shards_scored" 2>&1 | grep -E "passed|failed|Error|AssertionError: seed|FAILED" | tail -30
