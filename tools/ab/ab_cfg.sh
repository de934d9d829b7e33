#!/bin/bash
# Interleaved A/B of configurations on ONE box: tools/ab/ab_cfg.sh <sizes> <reps> <cfg> [<cfg> ...]
#   cfg = "label|so path or - (the tree's library)|ENV=.. ENV=..|extra bench args"
sizes=$1; reps=$2; shift 2
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export UGVC_SYNTH_CACHE=/tmp/ugvc_synth
cp variantcalling_amd/libugvc_mi355x.so /tmp/ab_tree.so
for n in $sizes; do
for i in $(seq $reps); do
  for cfg in "$@"; do
    IFS='|' read -r label so envs args <<< "$cfg"
    if [ "$so" = "-" ]; then cp /tmp/ab_tree.so variantcalling_amd/libugvc_mi355x.so; else cp "$so" variantcalling_amd/libugvc_mi355x.so; fi
    env $envs python bench.py --variants $n --steps 40 --warmup 5 --cpu-sample 0 --no-e2e --no-other $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$n', '$label', 'ms_per_step %.4f kernel_ms %.4f p5 %.4f p50 %.4f p95 %.4f' % (d['ms_per_step'], r['kernel_ms'], r['kernel_ms_p5'], r['kernel_ms_p50'], r['kernel_ms_p95']), d['parity']['oracle_slice_bit_exact'])"
  done
done
done
cp /tmp/ab_tree.so variantcalling_amd/libugvc_mi355x.so
