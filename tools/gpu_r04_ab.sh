#!/bin/bash
# GPU suite on the working tree, then the working tree's library against tools/ab/base_r03.so (alternating, same box), at 5 M and at
# the 8-rank shard size.  bash tools/gpu_r04_ab.sh [tag]   (on the GPU box, via gpurun)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp UGVC_SYNTH_CACHE=/tmp/ugvc_synth
TAG=${1:-ab}
mkdir -p gpurun_out
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/r04_${TAG}_pytest.raw 2>&1
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r04_${TAG}_pytest.raw | tail -25 > gpurun_out/r04_${TAG}_pytest.txt
tail -4 gpurun_out/r04_${TAG}_pytest.txt
fi
{
echo "# tools/ab/ab.sh tools/ab/base_r03.so : 'other' = the library of the previous commit, 'new' = the working tree; bench.py --steps 40 --warmup 5 (150-pass spin-up), kernel ms = HIP events around the pass"
bash tools/ab/ab.sh tools/ab/base_r03.so
echo "# the same at the 8-rank shard size (--variants 625000)"
bash tools/ab/ab.sh tools/ab/base_r03.so --variants 625000
} > gpurun_out/r04_${TAG}_ab.txt 2>&1
cat gpurun_out/r04_${TAG}_ab.txt
