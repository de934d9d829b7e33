#!/bin/bash
# The evidence set of a round from ONE tree (profiles/HEAD names the commit), on the GPU box via gpurun:
#   bash tools/gpu_evidence.sh <tag> <step> [<step> ...]       e.g.  tools/gpu_evidence.sh r05 bench shards prof sq
# steps:  tests    whole -m gpu suite + smoke                       -> <tag>_pytest_gpu.txt, <tag>_smoke.txt
#         bench    headline line (driver-style run)                 -> <tag>_bench.json
#         work     the other workloads: c2 pileup sec_apply c5_gemm -> <tag>_bench_<w>.json
#         shards   5 M / 2.5 M / 1.25 M / 625 k variants per pass   -> <tag>_shard_sizes.txt  (the strong-scaling floor)
#         prof     rocprofv3 --kernel-trace --stats + FETCH_SIZE / WRITE_SIZE passes of every workload
#                                                                   -> <tag>_kernel_stats.txt, <tag>_rocprof_summary.txt, hbm_traffic.json
#         sq       SQ instruction counters of the pass (all / without the SNP walk) -> <tag>_sq_counters.txt
#         c5sq     SQ counters of the C5 GEMM kernels (vector vs MFMA work)  -> <tag>_c5_sq_counters.txt
#         wclk     per-wave clocks (UGVC_WAVE_CLK, tools/wave_clk.py) -> <tag>_wave_clk.txt
#         cli      filter_variants_pipeline on a 5 M-record VCF, stage table -> <tag>_c1_pipeline_5M.txt
# Everything lands in gpurun_out/; copy what is to be judged into profiles/.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp UGVC_SYNTH_CACHE=/tmp/ugvc_synth
O=gpurun_out; mkdir -p $O
T=$1; shift
nolog() { grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl"; }
line() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[2], 'ms', round(d['ms_per_step'],4), 'kernel_ms', round(r.get('kernel_ms',0),4), 'frac', round(r['frac'],4), 'traffic', r.get('traffic'), (r.get('feature_build') or {}).get('frac'), d.get('parity'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))" "$1" "$2"; }
prof() {  # tag, traffic key, kernel filter, bench args...
  local tag=$1 key=$2 filt=$3; shift 3
  local CMD="python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-e2e --no-other $*"
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
for step in "$@"; do case $step in
tests)
  timeout 1500 python -m pytest tests -m gpu -q > $O/${T}_pytest_gpu.raw 2>&1
  nolog < $O/${T}_pytest_gpu.raw | tail -15 > $O/${T}_pytest_gpu.txt; tail -2 $O/${T}_pytest_gpu.txt
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | nolog | tail -12 | tee $O/${T}_smoke.txt ;;
bench)
  python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err; tail -c 600 $O/${T}_bench.json; line $O/${T}_bench.json filter ;;
work)
  for w in c2 pileup sec_apply c5_gemm; do
    python bench.py --workload $w --steps 30 --warmup 5 > $O/${T}_bench_$w.json 2> $O/${T}_bench_$w.err; line $O/${T}_bench_$w.json $w
  done ;;
shards)
  { echo "# python bench.py --variants N --steps 40 --warmup 5 --cpu-sample 0 --no-e2e : kernel time of one pass (HIP events, after the spin-up), one GPU"
  for n in 5000000 2500000 1250000 625000; do
  python bench.py --variants $n --steps 40 --warmup 5 --cpu-sample 0 --no-e2e --no-other 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['config']['variants_per_gpu'], 'variants: pass mean %.1f us  p50 %.1f us  step %.1f us  parity' % (r['kernel_ms']*1e3, r['kernel_ms_p50']*1e3, d['ms_per_step']*1e3), d['parity'])"
  done; } > $O/${T}_shard_sizes.txt; cat $O/${T}_shard_sizes.txt ;;
prof)
  cp profiles/hbm_traffic.json $O/hbm_traffic.json 2>/dev/null
  { prof filter "filter:4999706:1" "fused5,forest5"
    prof c2 "c2:1000000:1" "fused5,forest5" --workload c2
    prof pileup "pileup:5000000:1" "pileup_kernel" --workload pileup
    prof sec_apply "sec_apply:4999706:1" "sec_apply" --workload sec_apply
    prof c5 "c5_feature_build:1999870:1" "true>" --workload c5_gemm; } > $O/${T}_kernel_stats.txt 2>&1
  cat $O/${T}_kernel_stats.txt
  python tools/summarize_prof.py $O/prof_filter > $O/${T}_rocprof_summary.txt 2>&1; head -12 $O/${T}_rocprof_summary.txt
  python -c "
import json; d=json.load(open('$O/hbm_traffic.json'))
for k,v in d['workloads'].items(): print(k, v.get('bytes_per_launch'), v.get('commit'))" ;;
sq)
  : > $O/${T}_sq_counters.txt
  for var in ${SQ_VARIANTS:-0 131072}; do      # (131072 no SNP walk, 262144 no indel tiles, 524288 no side-table joins: sums of them ablate several)
    rm -rf $O/pm
    timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pm -o pmc -- python bench.py --steps 4 --warmup 1 --spinup 0 --cpu-sample 0 --no-e2e --no-other --check-rows 0 --variant $var > $O/pm.log 2>&1 || tail -3 $O/pm.log
    python - "$var" "$T" <<'PY'
import csv, glob, sys, collections
var, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:30]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"gpurun_out/{tag}_sq_counters.txt", "a") as out:
    for k, d in sorted(agg.items()):
        if "fused5" not in k and "forest5" not in k: continue
        out.write(f"variant {var:>7s} {k:30s} " + "  ".join(f"{c[3:]}={sum(v)/len(v)/1e6:.2f}M" for c, v in sorted(d.items())) + "\n")
PY
  done
  rm -rf $O/pm; cat $O/${T}_sq_counters.txt ;;
c5sq)
  rm -rf $O/pm5
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pm5 -o pmc -- python bench.py --workload c5_gemm --steps 4 --warmup 1 --cpu-sample 0 > $O/pm5.log 2>&1 || tail -3 $O/pm5.log
  python - "$T" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pm5/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"gpurun_out/{tag}_c5_sq_counters.txt", "w") as out:
    out.write("# rocprofv3 --pmc ... -- python bench.py --workload c5_gemm: mean per launch, millions (SQ_* summed over the chip)\n")
    for k, d in sorted(agg.items()):
        if "forest_" not in k: continue
        out.write(f"{k:48s} launches {len(next(iter(d.values()))):3d}  " + "  ".join(f"{c[3:]}={sum(v)/len(v)/1e6:.2f}M" for c, v in sorted(d.items())) + "\n")
PY
  rm -rf $O/pm5; cat $O/${T}_c5_sq_counters.txt ;;
wclk)
  for n in 5000000 625000; do
    UGVC_WAVE_CLK=/tmp/wclk.bin python bench.py --variants $n --steps 3 --warmup 2 --spinup 20 --cpu-sample 0 --no-e2e --no-other --check-rows 0 > /tmp/wclk.json 2>/tmp/wclk.err || tail -3 /tmp/wclk.err
    echo "== $n variants"; python tools/wave_clk.py /tmp/wclk.bin
  done > $O/${T}_wave_clk.txt; grep -E "==|workgroup end" $O/${T}_wave_clk.txt ;;
cli)
  UGVC_VCF_TRACE=1 python tools/bench_pipeline.py 5000000 > $O/${T}_c1_pipeline_5M.txt 2>&1
  grep -v "^\[vcf\]" $O/${T}_c1_pipeline_5M.txt ;;
esac; done
