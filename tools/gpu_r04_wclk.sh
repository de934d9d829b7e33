#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp UGVC_SYNTH_CACHE=/tmp/ugvc_synth
mkdir -p gpurun_out
for n in ${WCLK_N:-5000000 625000}; do
UGVC_WAVE_CLK=/tmp/wclk.bin python bench.py --variants $n --steps 3 --warmup 2 --spinup 20 --cpu-sample 0 --no-e2e --check-rows 0 $WCLK_ARGS > /tmp/wclk.json 2>/tmp/wclk.err || tail -3 /tmp/wclk.err
echo "== $n variants" ; python tools/wave_clk.py /tmp/wclk.bin
done > gpurun_out/${WCLK_OUT:-r04_wave_clk.txt}
cat gpurun_out/${WCLK_OUT:-r04_wave_clk.txt}
