#!/bin/bash
# several libraries on ONE box, interleaved: bash tools/gpu_r04_multi_ab.sh a.so b.so c.so ...   (bench args via AB_ARGS)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp UGVC_SYNTH_CACHE=/tmp/ugvc_synth
mkdir -p gpurun_out
cp variantcalling_amd/libugvc_mi355x.so /tmp/keep.so
{
for rep in 1 2 3; do
for lib in "$@"; do
cp "$lib" variantcalling_amd/libugvc_mi355x.so
python bench.py --steps 40 --warmup 5 --cpu-sample 0 --no-e2e ${AB_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$lib', 'ms_per_step %.4f kernel_ms %.4f p5 %.4f p50 %.4f p95 %.4f' % (d['ms_per_step'], r['kernel_ms'], r['kernel_ms_p5'], r['kernel_ms_p50'], r['kernel_ms_p95']), d['parity']['oracle_slice_bit_exact'])"
done; done
} > gpurun_out/${AB_OUT:-r04_multi_ab.txt} 2>&1
cp /tmp/keep.so variantcalling_amd/libugvc_mi355x.so
cat gpurun_out/${AB_OUT:-r04_multi_ab.txt}
