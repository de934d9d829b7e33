#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_pipelines.py -m gpu -q -x -k "sec or two_ranks or filter_variants" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4
UGVC_VCF_TRACE=1 python tools/bench_pipeline.py 5000000 > gpurun_out/r04_c1_pipeline_5M.txt 2>&1
UGVC_DEFLATE=zlib python tools/bench_pipeline.py 5000000 2>/dev/null | sed 's/^/[UGVC_DEFLATE=zlib] /' >> gpurun_out/r04_c1_pipeline_5M.txt
grep -v "^\[vcf\] write   \|gather parts\|write format\|write deflate" gpurun_out/r04_c1_pipeline_5M.txt
