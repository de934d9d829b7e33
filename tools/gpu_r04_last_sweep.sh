#!/bin/bash
# env-only knobs of the evidence tree on ONE box, three rounds: indel cost of the role split, one-tile moves between SNP waves
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp UGVC_SYNTH_CACHE=/tmp/ugvc_synth
mkdir -p gpurun_out
run() {
  local label=$1; shift
  env "$@" python bench.py --steps 40 --warmup 5 --cpu-sample 0 --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-28s' % '$label', 'ms_per_step %.4f kernel_ms %.4f p5 %.4f p50 %.4f p95 %.4f' % (d['ms_per_step'], r['kernel_ms'], r['kernel_ms_p5'], r['kernel_ms_p50'], r['kernel_ms_p95']), d['parity']['oracle_slice_bit_exact'])"
}
{
for rep in 1 2 3; do
run default X=1
run cost_0.8 UGVC_INDEL_COST=0.8
run cost_0.9 UGVC_INDEL_COST=0.9
run w8_to_w12 UGVC_SNP_W=20,20,20,20,20,20,20,20,19,20,20,20,11
run w8_w10_to_w12 UGVC_SNP_W=20,20,20,20,20,20,20,20,19,20,19,20,12
run w8to11_to_w12 UGVC_SNP_W=20,20,20,20,20,20,20,20,19,19,19,19,14
done
} > gpurun_out/r04_last_sweep.txt 2>&1
cat gpurun_out/r04_last_sweep.txt
