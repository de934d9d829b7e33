#!/bin/bash
# SNP-wave tile shares on ONE box: UGVC_SNP_W / UGVC_INDEL_COST (profiling knobs of model_pack.hip: v5_fill_args), interleaved, two rounds
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp UGVC_SYNTH_CACHE=/tmp/ugvc_synth
mkdir -p gpurun_out
run() {  # label, env assignments...
  local label=$1; shift
  env "$@" python bench.py --steps 40 --warmup 5 --cpu-sample 0 --no-e2e ${AB_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-28s' % '$label', 'ms_per_step %.4f kernel_ms %.4f p5 %.4f p50 %.4f p95 %.4f' % (d['ms_per_step'], r['kernel_ms'], r['kernel_ms_p5'], r['kernel_ms_p50'], r['kernel_ms_p95']), d['parity']['oracle_slice_bit_exact'])"
}
{
for rep in 1 2; do
run base X=1
run w8_minus1_to_w12 UGVC_SNP_W=20,20,20,20,20,20,20,20,19,20,20,20,11
run simd0_minus3 UGVC_SNP_W=19,20,20,20,19,20,20,20,19,20,20,20,14
run w12_16 UGVC_SNP_W=39,39,39,39,39,39,39,39,39,39,39,39,32
run even UGVC_SNP_W=1,1,1,1,1,1,1,1,1,1,1,1,1
run age_21_20_19 UGVC_SNP_W=21,21,21,21,20,20,20,20,19,19,19,19,10
run age_22_20_18 UGVC_SNP_W=22,22,22,22,20,20,20,20,18,18,18,18,10
run young_more UGVC_SNP_W=19,19,19,19,20,20,20,20,21,21,21,21,10
run four_indel_waves UGVC_INDEL_COST=1.4
run four_indel_w_age UGVC_INDEL_COST=1.4 UGVC_SNP_W=22,22,22,22,21,21,21,21,20,20,20,20
run w12_last_simd0_heavy UGVC_SNP_W=21,20,20,20,21,20,20,20,21,20,20,20,8
done
} > gpurun_out/${AB_OUT:-r04_shares2.txt} 2>&1
cat gpurun_out/${AB_OUT:-r04_shares2.txt}
