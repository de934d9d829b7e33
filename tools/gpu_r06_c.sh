#!/bin/bash
# round 6, lease C: dense-occupancy feature-matrix kernel A/B (UGVC_FM_WPE=6) with its parity tests, the CLI with the new writer
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp UGVC_SYNTH_CACHE=/tmp/ugvc_synth; O=gpurun_out; mkdir -p $O
nolog() { grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl"; }
{ echo "== parity of the feature matrix under UGVC_FM_WPE=6"
  UGVC_FM_WPE=6 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_annotate.py -m gpu -x -q -k "feature or matrix or c5 or annotate or train" 2>&1 | nolog | tail -4
  for rep in 1 2 3; do for wpe in 0 6; do
    UGVC_FM_WPE=$wpe python bench.py --workload c5_gemm --steps 20 --warmup 3 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['roofline']['feature_build']
print('wpe $wpe feature build ms %.4f frac %.4f' % (f['ms'], f['frac']), d['parity'])"
  done; done; } > $O/r06_fm_dense_ab.txt 2>&1
cat $O/r06_fm_dense_ab.txt
{ for rep in 1 2 3; do UGVC_VCF_TRACE=1 python tools/bench_pipeline.py 5000000 2>&1 | nolog | grep -v "^\[vcf\] \(read\|fasta\)" | tail -70; done; } > $O/r06_c1_pipeline_5M_raw2.txt 2>&1
grep -v "^\[vcf\]" $O/r06_c1_pipeline_5M_raw2.txt; grep "all flushes\|last batch\|tabix" $O/r06_c1_pipeline_5M_raw2.txt
