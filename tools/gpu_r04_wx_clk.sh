#!/bin/bash
# phase clocks of the feature-matrix launch (the library rebuilt with -DUGVC_PHASE_CLOCK in a scratch copy)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp UGVC_SYNTH_CACHE=/tmp/ugvc_synth
rm -rf /tmp/clk && mkdir -p /tmp/clk && cp -r variantcalling_amd oracle include profiles tests bench.py /tmp/clk/ 2>/dev/null
( cd /tmp/clk/variantcalling_amd/csrc && touch kernels_v5.hip && make EXTRA=-DUGVC_PHASE_CLOCK -j8 > /tmp/clk/build.log 2>&1; tail -2 /tmp/clk/build.log )
( cd /tmp/clk && python bench.py --workload c5_gemm --steps 2 --warmup 0 --cpu-sample 0 2>&1 | grep -E "^clk|^iclk|^  cut" | awk '{k=$1" "$2" "$3; last[k]=$0} END{for (k in last) print last[k]}' | sort ) > gpurun_out/${CLK_OUT:-r04_wx_phase_clocks.txt}
cat gpurun_out/${CLK_OUT:-r04_wx_phase_clocks.txt}
