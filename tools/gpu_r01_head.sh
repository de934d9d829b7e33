#!/bin/bash
# HEAD re-validation: full GPU suite, smoke, bench line, kernel-trace profile (no PMC).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/smoke.log
echo "== bench"; timeout 600 python bench.py 2>&1 | tail -2 | tee gpurun_out/bench.log
echo "== aux"; timeout 300 python tools/bench_aux.py 2>&1 | tail -12 | tee gpurun_out/bench_aux.log
rm -rf gpurun_out/prof_head
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_head/trace -o trace -- python bench.py --steps 20 --warmup 3 --cpu-sample 0 > gpurun_out/prof_head.bench.log 2>&1
tail -1 gpurun_out/prof_head.bench.log
find gpurun_out/prof_head -name "*kernel_stats.csv" | head -1 | xargs -r head -8
find gpurun_out/prof_head -type f -size +2M -delete
