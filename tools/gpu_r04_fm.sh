#!/bin/bash
# the C5 feature-matrix build: libraries A/B on one box, then the indel-tile weight of its wave split (UGVC_FM_INDEL_W)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp UGVC_SYNTH_CACHE=/tmp/ugvc_synth
mkdir -p gpurun_out
cp variantcalling_amd/libugvc_mi355x.so /tmp/keep.so
fm() {
  local label=$1; shift
  env "$@" python bench.py --workload c5_gemm --steps 10 --warmup 2 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['roofline']['feature_build']
print('%-24s' % '$label', 'feature build %.4f ms  %.0f GB/s  frac %.3f | gemm %.3f ms' % (f['ms'], f['achieved'], f['frac'], d['roofline']['kernel_ms']), d['parity'])"
}
{
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipelines.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -2
for rep in 1 2; do for lib in "$@"; do cp "$lib" variantcalling_amd/libugvc_mi355x.so; fm "$lib" X=1; done; done
cp /tmp/keep.so variantcalling_amd/libugvc_mi355x.so
for w in 1.5 2 2.5 3 4; do fm "indel_w=$w" UGVC_FM_INDEL_W=$w; done
} > gpurun_out/${AB_OUT:-r04_fm_ab.txt} 2>&1
cp /tmp/keep.so variantcalling_amd/libugvc_mi355x.so
cat gpurun_out/${AB_OUT:-r04_fm_ab.txt}
