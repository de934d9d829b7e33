#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
for nv in 625000 5000000; do
rm -rf gpurun_out/prof_small
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_small/trace -o trace -- python bench.py --steps 10 --warmup 2 --cpu-sample 0 --variants $nv > gpurun_out/prof_small.log 2>&1
echo "-- $nv"; find gpurun_out/prof_small -name "*kernel_stats.csv" | head -1 | xargs -r grep bracket | cut -c1-110
done
find gpurun_out/prof_small -type f -size +1M -delete
