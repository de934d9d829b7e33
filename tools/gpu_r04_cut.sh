cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp UGVC_SYNTH_CACHE=/tmp/ugvc_synth
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6
AB_OUT=r04_cut_ab.txt bash tools/gpu_r04_multi_ab.sh tools/ab/gallop.so tools/ab/cut.so
AB_ARGS="--variants 625000" AB_OUT=r04_cut_ab_625k.txt bash tools/gpu_r04_multi_ab.sh tools/ab/gallop.so tools/ab/cut.so
WCLK_N=5000000 WCLK_OUT=r04_wave_clk_cut.txt bash tools/gpu_r04_wclk.sh | grep -E "workgroup end|slowest" -A3 | head -12
