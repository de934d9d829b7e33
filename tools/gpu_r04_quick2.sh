#!/bin/bash
# quick A/B of libraries on one box + the fused kernel's LDS footprint: bash tools/gpu_r04_quick2.sh a.so b.so ...
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp UGVC_SYNTH_CACHE=/tmp/ugvc_synth
mkdir -p gpurun_out
UGVC_DEBUG_SYNC=1 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-e2e 2>&1 | grep -m2 "ugvc v5" 
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
AB_OUT=${AB_OUT:-r04_quick2_ab.txt} bash tools/gpu_r04_multi_ab.sh "$@"
AB_ARGS="--variants 625000" AB_OUT=${AB_OUT:-r04_quick2_ab}_625k.txt bash tools/gpu_r04_multi_ab.sh "$@"
