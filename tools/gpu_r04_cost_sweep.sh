#!/bin/bash
# the indel-tile cost of the wave-role split (UGVC_INDEL_COST, profiling knob of v5_fill_args) on ONE box, two rounds, 5 M and 625 k
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp UGVC_SYNTH_CACHE=/tmp/ugvc_synth
mkdir -p gpurun_out
run() {
  local label=$1; shift
  env "$@" python bench.py --steps 40 --warmup 5 --cpu-sample 0 --no-e2e ${AB_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-28s' % '$label', 'ms_per_step %.4f kernel_ms %.4f p5 %.4f p50 %.4f p95 %.4f' % (d['ms_per_step'], r['kernel_ms'], r['kernel_ms_p5'], r['kernel_ms_p50'], r['kernel_ms_p95']), d['parity']['oracle_slice_bit_exact'])"
}
{
for n in 5000000 625000; do
echo "== $n variants"; AB_ARGS="--variants $n"
for rep in 1 2; do
run cost_1.0 X=1
run cost_0.9 UGVC_INDEL_COST=0.9
run cost_0.85 UGVC_INDEL_COST=0.85
run cost_0.8 UGVC_INDEL_COST=0.8
run cost_0.7 UGVC_INDEL_COST=0.7
run cost_0.6 UGVC_INDEL_COST=0.6
done; done
} > gpurun_out/${AB_OUT:-r04_cost_sweep.txt} 2>&1
cat gpurun_out/${AB_OUT:-r04_cost_sweep.txt}
