#!/bin/bash
# Round-4 evidence set from ONE tree (profiles/HEAD names the commit): GPU suite, smoke, headline bench, the other workloads
# (each with its CPU baseline), shard sizes, rocprofv3 kernel traces of every workload, FETCH_SIZE / WRITE_SIZE passes of every
# workload (profiles/hbm_traffic.json, keyed by workload), SQ counters, per-wave clocks, the CLI stage table.
# bash tools/gpu_r04_final.sh   (on the GPU box, via gpurun; ~15 min)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp UGVC_SYNTH_CACHE=/tmp/ugvc_synth
O=gpurun_out
mkdir -p $O
nolog() { grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl"; }
# ---- 1. tests + smoke
timeout 1500 python -m pytest tests -m gpu -q > $O/r04_pytest_gpu.raw 2>&1
nolog < $O/r04_pytest_gpu.raw | tail -15 > $O/r04_pytest_gpu.txt; tail -2 $O/r04_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | nolog | tail -2 | tee $O/r04_smoke.txt
# ---- 2. headline + the other workloads
python bench.py > $O/r04_bench.json 2> $O/r04_bench.err; tail -c 400 $O/r04_bench.json
for w in c2 pileup sec_apply c5_gemm; do
  python bench.py --workload $w --steps 30 --warmup 5 > $O/r04_bench_$w.json 2> $O/r04_bench_$w.err
  python -c "
import json
d=json.loads(open('$O/r04_bench_$w.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$w', 'ms', round(d['ms_per_step'],4), 'kernel_ms', round(r.get('kernel_ms',0),4), 'frac', round(r['frac'],4), (r.get('feature_build') or {}).get('frac'), d.get('parity'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))"
done
# ---- 3. shard sizes (the strong-scaling floor)
{ echo "# python bench.py --variants N --steps 40 --warmup 5 --cpu-sample 0 --no-e2e : kernel time of one pass (HIP events, after the 150-pass spin-up), one GPU, round 4"
for n in 5000000 2500000 1250000 625000; do
python bench.py --variants $n --steps 40 --warmup 5 --cpu-sample 0 --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['config']['variants_per_gpu'], 'variants: pass mean %.1f us  p50 %.1f us  step %.1f us  parity' % (r['kernel_ms']*1e3, r['kernel_ms_p50']*1e3, d['ms_per_step']*1e3), d['parity'])"
done; } > $O/r04_shard_sizes.txt; cat $O/r04_shard_sizes.txt
# ---- 4. rocprofv3: kernel trace + PMC passes per workload
prof() {  # tag, key, kernel filter, bench args...
  local tag=$1 key=$2 filt=$3; shift 3
  local CMD="python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-e2e $*"
  rm -rf $O/prof_$tag
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag/trace -o trace -- $CMD > $O/prof_$tag.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/prof_$tag/pmc_$c -o pmc -- $CMD > $O/prof_$tag.pmc_$c.log 2>&1
  done
  python tools/make_traffic_json.py $O/prof_$tag --key "$key" --kernels "$filt" --dst $O/hbm_traffic.json > /dev/null
  python - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
print("==", tag, ": name | calls | average us | total %")
for f in glob.glob(f"gpurun_out/prof_{tag}/trace/**/*kernel_stats.csv", recursive=True):
    for k, r in enumerate(csv.DictReader(open(f))):
        if k < 6: print(f"{r['Name'][:84]:84s} | {r['Calls']:>5s} | {float(r['AverageNs'])/1e3:10.1f} | {r.get('Percentage','')}")
PY
  find $O/prof_$tag -type f -size +2M -delete
}
cp profiles/hbm_traffic.json $O/hbm_traffic.json 2>/dev/null
{ prof filter "filter:4999706:1" "fused5,forest5"
  prof c2 "c2:1000000:1" "fused5,forest5" --workload c2
  prof pileup "pileup:5000000:1" "pileup_kernel" --workload pileup
  prof sec_apply "sec_apply:4999706:1" "sec_apply" --workload sec_apply
  prof c5 "c5_feature_build:1999870:1" "true>" --workload c5_gemm; } > $O/r04_kernel_stats.txt 2>&1
cat $O/r04_kernel_stats.txt
python tools/summarize_prof.py $O/prof_filter > $O/r04_rocprof_summary.txt 2>&1; head -12 $O/r04_rocprof_summary.txt
python -c "
import json; d=json.load(open('$O/hbm_traffic.json'))
for k,v in d['workloads'].items(): print(k, v.get('bytes_per_launch'), v.get('commit'))"
# ---- 5. SQ counters of the pass (all of it / without the SNP walk)
: > $O/r04_sq_counters.txt
for var in 0 131072; do
  rm -rf $O/pm
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pm -o pmc -- python bench.py --steps 4 --warmup 1 --spinup 0 --cpu-sample 0 --no-e2e --check-rows 0 --variant $var > $O/pm.log 2>&1 || tail -3 $O/pm.log
  python - "$var" <<'PY' >> gpurun_out/r04_sq_counters.txt
import csv, glob, sys, collections
var = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:30]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    if "fused5" not in k and "forest5" not in k: continue
    print(f"variant {var:>7s} {k:30s} " + "  ".join(f"{c[3:]}={sum(v)/len(v)/1e6:.2f}M" for c, v in sorted(d.items())))
PY
done
rm -rf $O/pm; cat $O/r04_sq_counters.txt
# ---- 6. per-wave clocks, CLI
WCLK_OUT=r04_wave_clk_final.txt bash tools/gpu_r04_wclk.sh > /dev/null; grep -E "==|workgroup end|^   0|^   8|^  13" $O/r04_wave_clk_final.txt
UGVC_VCF_TRACE=1 python tools/bench_pipeline.py 5000000 > $O/r04_c1_pipeline_5M.txt 2>&1
UGVC_DEFLATE=zlib python tools/bench_pipeline.py 5000000 2>/dev/null | sed 's/^/[UGVC_DEFLATE=zlib] /' >> $O/r04_c1_pipeline_5M.txt
grep -v "^\[vcf\]" $O/r04_c1_pipeline_5M.txt
