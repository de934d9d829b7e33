#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== parity v5"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "v5" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
echo "== v5"; timeout 300 python tools/ablate3.py 5000000 65536 2>&1 | tail -10 | tee gpurun_out/ablate_v5.log
rm -rf gpurun_out/prof_small
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_small/trace -o trace -- python bench.py --steps 10 --warmup 2 --cpu-sample 0 --variant 65536 > gpurun_out/prof_small.log 2>&1
find gpurun_out/prof_small -name "*kernel_stats.csv" | head -1 | xargs -r head -6 | cut -c1-110
find gpurun_out/prof_small -type f -size +1M -delete
