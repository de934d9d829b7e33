#!/bin/bash
# the CLI stage table (5 M-record BGZF VCF) twice, and the tool tests
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_pipelines.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -2
UGVC_VCF_TRACE=1 python tools/bench_pipeline.py 5000000 > gpurun_out/r04_c1_pipeline_5M_b.txt 2>&1
python tools/bench_pipeline.py 5000000 2>/dev/null | sed 's/^/[second run] /' >> gpurun_out/r04_c1_pipeline_5M_b.txt
grep -v "^\[vcf\] write   \|gather parts\|write format" gpurun_out/r04_c1_pipeline_5M_b.txt
