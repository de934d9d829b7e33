#!/bin/bash
# quick A/B of the working tree's library against $BASE (default tools/ab/base_walk6.so) at 5 M and 625 k + per-wave clocks; SKIP_TESTS=1 skips the suite
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp UGVC_SYNTH_CACHE=/tmp/ugvc_synth
TAG=${1:-quick}
BASE=${BASE:-tools/ab/base_walk6.so}
mkdir -p gpurun_out
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/r04_${TAG}_pytest.raw 2>&1
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r04_${TAG}_pytest.raw | tail -25 > gpurun_out/r04_${TAG}_pytest.txt
tail -4 gpurun_out/r04_${TAG}_pytest.txt
fi
{
echo "# 'other' = $BASE, 'new' = the working tree; bench.py --steps 40 --warmup 5 (150-pass spin-up); kernel ms = HIP events around the pass"
bash tools/ab/ab.sh $BASE
echo "# --variants 625000"
bash tools/ab/ab.sh $BASE --variants 625000
} > gpurun_out/r04_${TAG}_ab.txt 2>&1
cat gpurun_out/r04_${TAG}_ab.txt
WCLK_OUT=r04_wave_clk_${TAG}.txt bash tools/gpu_r04_wclk.sh | grep -E "==|first-tile|^   0|^   8|^  13|workgroup end"
