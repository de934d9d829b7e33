#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp UGVC_SYNTH_CACHE=/tmp/ugvc_synth
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "c5 or feature_matrix" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8
python bench.py --workload c5_gemm --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/r04_bench_c5_gemm_try.json 2> gpurun_out/r04_bench_c5_gemm_try.err
python -c "
import json
d=json.loads(open('gpurun_out/r04_bench_c5_gemm_try.json').read().strip().splitlines()[-1]); r=d['roofline']
print('gemm ms', r['kernel_ms'], 'TOP/s', r['achieved'], 'frac', r['frac'], 'vs measured ceiling', r['tops_vs_measured_i8_ceiling'], 'traversal ms', r['traversal_ms'], 'feature build', r['feature_build']['ms'], r['feature_build']['frac'], d['parity'])"
tail -3 gpurun_out/r04_bench_c5_gemm_try.err
