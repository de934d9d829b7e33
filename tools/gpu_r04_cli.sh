#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out
UGVC_VCF_TRACE=1 python tools/bench_pipeline.py 5000000 > gpurun_out/r04_c1_pipeline_5M.txt 2>&1
UGVC_DEFLATE=zlib python tools/bench_pipeline.py 5000000 2>/dev/null | sed 's/^/[UGVC_DEFLATE=zlib] /' >> gpurun_out/r04_c1_pipeline_5M.txt
cat gpurun_out/r04_c1_pipeline_5M.txt
