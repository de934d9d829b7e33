#!/bin/bash
# GPU suite (indel-heavy tests first), A/B of libraries on one box, phase clocks of the working-tree library
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp UGVC_SYNTH_CACHE=/tmp/ugvc_synth
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -3
AB_OUT=${AB_OUT:-r04_quick3_ab.txt} bash tools/gpu_r04_multi_ab.sh "$@"
AB_ARGS="--variants 625000" AB_OUT=${AB_OUT:-r04_quick3_ab}_625k.txt bash tools/gpu_r04_multi_ab.sh "$@"
CLK_OUT=r04_phase_clocks_indel2.txt bash tools/gpu_phase_clocks.sh | grep -E "iclk|cut" | cut -c1-300
