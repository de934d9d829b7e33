#!/bin/bash
# GPU suite, A/B of libraries on one box at 5 M and 625 k, per-wave clocks of the working-tree library
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp UGVC_SYNTH_CACHE=/tmp/ugvc_synth
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -3
AB_OUT=${AB_OUT:-r04_quick4_ab}.txt bash tools/gpu_r04_multi_ab.sh "$@"
AB_ARGS="--variants 625000" AB_OUT=${AB_OUT:-r04_quick4_ab}_625k.txt bash tools/gpu_r04_multi_ab.sh "$@"
WCLK_OUT=${AB_OUT:-r04_quick4}_wclk.txt bash tools/gpu_r04_wclk.sh | grep -E "==|workgroup end|slowest" -A2 | head -30
