#!/bin/bash
# Round-3 evidence set from ONE tree (profiles/HEAD names the commit): GPU test suite, headline bench, the other workloads,
# rocprofv3 kernel trace + PMC passes.  bash tools/gpu_r03_final.sh   (on the GPU box, via gpurun)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r03_pytest_gpu.raw 2>&1
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r03_pytest_gpu.raw | tail -15 > gpurun_out/r03_pytest_gpu.txt
tail -3 gpurun_out/r03_pytest_gpu.txt
python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
tail -c 600 gpurun_out/r03_bench.json
for w in c2 pileup sec_apply c5_gemm; do
  python bench.py --workload $w --steps 30 --warmup 5 --cpu-sample 0 > gpurun_out/r03_bench_$w.json 2> gpurun_out/r03_bench_$w.err
  python -c "
import json
d=json.loads(open('gpurun_out/r03_bench_$w.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$w', round(d['ms_per_step'],4), round(r['frac'],4), (r.get('feature_build') or {}).get('frac'), d.get('parity'))"
done
bash tools/gpu_profile.sh r03 > gpurun_out/r03_profile.log 2>&1
head -5 gpurun_out/prof_r03_summary.txt
