#!/bin/bash
# pass time under the fused kernel's profiling bits (bench.py --variant), one box, two rounds
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp UGVC_SYNTH_CACHE=/tmp/ugvc_synth
mkdir -p gpurun_out
{
for rep in 1 2; do
for v in 0 131072 524288 655360 262144 393216 917504 536870912; do
python bench.py --steps 40 --warmup 5 --cpu-sample 0 --no-e2e --variant $v ${AB_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('variant %9d' % $v, 'ms_per_step %.4f kernel_ms %.4f p50 %.4f' % (d['ms_per_step'], r['kernel_ms'], r['kernel_ms_p50']))"
done; done
} > gpurun_out/${AB_OUT:-r04_ablate_times.txt} 2>&1
cat gpurun_out/${AB_OUT:-r04_ablate_times.txt}
