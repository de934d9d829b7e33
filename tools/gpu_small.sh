#!/bin/bash
# ad-hoc GPU check: edit, then `gpurun -- 'bash tools/gpu_small.sh'`
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
