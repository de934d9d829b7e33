#!/bin/bash
# rocprofv3 kernel trace of the headline bench (and optional variants): tools/gpu_prof.sh <tag> [bench args...]
tag=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
cd /root/repo
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o prof -- python bench.py --steps 10 --warmup 2 --cpu-sample 0 "$@" > gpurun_out/$tag.bench.log 2>&1
f=$(find gpurun_out/$tag -name "*kernel_stats.csv" | head -1)
echo "== $tag $@"; head -12 "$f" | cut -d, -f1-8
tail -c 400 gpurun_out/$tag.bench.log | grep -o '"ms_per_step": [0-9.]*'
find gpurun_out/$tag -name "*.db" -delete; find gpurun_out/$tag -name "*kernel_trace.csv" -size +5M -delete
