#!/bin/bash
# rocprofv3 kernel trace + PMC passes of the default bench command; summaries -> gpurun_out/
# usage (on the GPU box, via gpurun): bash tools/gpu_profile.sh <tag> [extra bench args]
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=${1:-r01}
shift || true
CMD="python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-e2e $*"
rm -rf gpurun_out/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG/trace -o trace -- $CMD > gpurun_out/prof_$TAG.bench.log 2>&1
tail -2 gpurun_out/prof_$TAG.bench.log
for c in ${PMC_COUNTERS-FETCH_SIZE WRITE_SIZE}; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/prof_$TAG/pmc_$c -o pmc -- $CMD > gpurun_out/prof_$TAG.pmc_$c.log 2>&1
done
find gpurun_out/prof_$TAG -type f | head -30
python tools/summarize_prof.py gpurun_out/prof_$TAG > gpurun_out/prof_${TAG}_summary.txt 2>&1
cat gpurun_out/prof_${TAG}_summary.txt
# keep only small files for the merge back
find gpurun_out/prof_$TAG -type f -size +2M -delete
