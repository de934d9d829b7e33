#!/bin/bash
# round-2 side measurements: pass time at the shard sizes of 2 / 4 / 8 ranks, SQ counters, phase clocks (a second
# build of the library with -DUGVC_PHASE_CLOCK into gpurun_out/, loaded through a copy of the package)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
{
  echo "# python bench.py --variants N --steps 40 --warmup 5 --cpu-sample 0 --no-e2e : kernel time of one pass (HIP events), one GPU"
  for nv in 5000000 2500000 1250000 625000; do
    python bench.py --variants $nv --steps 40 --warmup 5 --cpu-sample 0 --no-e2e | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['config']['variants_per_gpu'], 'variants: pass mean %.1f us  p50 %.1f us  step %.1f us  parity %s' % (r['kernel_ms']*1e3, r['kernel_ms_p50']*1e3, d['ms_per_step']*1e3, d['parity']))"
  done
} > gpurun_out/r02_shard_sizes.txt 2>&1
cat gpurun_out/r02_shard_sizes.txt
bash tools/gpu_pmc5.sh p5 0 2>&1 | tail -1 > gpurun_out/r02_sq_counters.txt
bash tools/gpu_pmc_lds.sh 0 2>&1 | grep -E "fused5|forest5" >> gpurun_out/r02_sq_counters.txt
cat gpurun_out/r02_sq_counters.txt
# phase clocks
rm -rf /tmp/clk && mkdir -p /tmp/clk && cp -r variantcalling_amd oracle include profiles bench.py /tmp/clk/ 2>/dev/null
( cd /tmp/clk/variantcalling_amd/csrc && touch kernels_v5.hip && make EXTRA=-DUGVC_PHASE_CLOCK -j8 > /tmp/clk/build.log 2>&1 )
( cd /tmp/clk && python bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-e2e 2>&1 | grep -E "^clk|^iclk|^fclk|issue" | sort | head -16 ) > gpurun_out/r02_phase_clocks.txt
cat gpurun_out/r02_phase_clocks.txt
