#!/bin/bash
# A/B of two builds of libugvc_mi355x.so on the same box, interleaved: tools/ab/ab.sh <other.so> [bench args]
other=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp variantcalling_amd/libugvc_mi355x.so /tmp/ab_new.so
for i in 1 2 3; do
  for which in other new; do
    if [ $which = other ]; then cp "$other" variantcalling_amd/libugvc_mi355x.so; else cp /tmp/ab_new.so variantcalling_amd/libugvc_mi355x.so; fi
    python bench.py --steps 40 --warmup 5 --cpu-sample 0 --no-e2e --no-other "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$which', 'ms_per_step %.4f kernel_ms %.4f p5 %.4f p50 %.4f p95 %.4f' % (d['ms_per_step'], r['kernel_ms'], r['kernel_ms_p5'], r['kernel_ms_p50'], r['kernel_ms_p95']), d['parity'])"
  done
done
cp /tmp/ab_new.so variantcalling_amd/libugvc_mi355x.so
