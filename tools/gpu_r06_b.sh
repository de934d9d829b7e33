#!/bin/bash
# round 6, lease B: smoke first (fresh lease), the gemm3 fallback test, the CLI with the writer's trace, the large-N bench lines
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp UGVC_SYNTH_CACHE=/tmp/ugvc_synth; O=gpurun_out; mkdir -p $O
nolog() { grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl"; }
{ echo "== smoke (first GPU process of the lease)"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | nolog | tail -4
  echo "== gemm3 fallback + c5"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gemm" 2>&1 | nolog | tail -4; } > $O/r06_lease_b.txt 2>&1
cat $O/r06_lease_b.txt | tail -8
{ for rep in 1 2 3; do UGVC_VCF_TRACE=1 python tools/bench_pipeline.py 5000000 2>&1 | nolog | grep -v "^\[vcf\] \(read\|fasta\)" | tail -70; done; } > $O/r06_c1_pipeline_5M_raw.txt 2>&1
grep -v "^\[vcf\]" $O/r06_c1_pipeline_5M_raw.txt; grep "all flushes\|last batch\|tabix" $O/r06_c1_pipeline_5M_raw.txt
unset UGVC_SYNTH_CACHE
for n in 50000000 500000000; do
  steps=10; [ $n = 500000000 ] && steps=4
  timeout 1500 python bench.py --variants $n --steps $steps --warmup 2 --spinup 10 --cpu-sample 0 --no-e2e --no-other --check-rows 20000 > $O/r06_bench_$((n/1000000))M.json 2> $O/r06_bench_$((n/1000000))M.err
  python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[1], 'ms', round(d['ms_per_step'],4), 'frac', round(r['frac'],4), d['parity'], 'setup_s', d['setup_s'])" $O/r06_bench_$((n/1000000))M.json || tail -3 $O/r06_bench_$((n/1000000))M.err
done
