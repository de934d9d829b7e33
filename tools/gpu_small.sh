#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not full_size and not c5 and not pileup and not sec and not bridging" 2>&1 | grep -E "passed|failed|Error" | tail -3
rm -rf gpurun_out/prof_small
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_small/trace -o trace -- python bench.py --steps 10 --warmup 2 --cpu-sample 0 --variants 625000 > gpurun_out/prof_small.log 2>&1
find gpurun_out/prof_small -name "*kernel_stats.csv" | head -1 | xargs -r head -5
find gpurun_out/prof_small -type f -size +1M -delete
