#!/bin/bash
# Round-4 first GPU call: the GPU suite on this round's HEAD, the headline bench, and the instruction counters of the fused
# kernel under its ablation bits (which part of the pass issues what).  bash tools/gpu_r04_diag.sh  (on the GPU box, via gpurun)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp UGVC_SYNTH_CACHE=/tmp/ugvc_synth
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/r04_pytest_gpu.raw 2>&1
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r04_pytest_gpu.raw | tail -15 > gpurun_out/r04_pytest_gpu_first.txt
tail -3 gpurun_out/r04_pytest_gpu_first.txt
python bench.py > gpurun_out/r04_bench_first.json 2> gpurun_out/r04_bench_first.err
tail -c 900 gpurun_out/r04_bench_first.json
: > gpurun_out/r04_sq_ablation.txt
for var in 0 131072 393216 917504 262144; do
  rm -rf gpurun_out/pm
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pm -o pmc -- python bench.py --steps 4 --warmup 1 --spinup 0 --cpu-sample 0 --no-e2e --check-rows 0 --variant $var > gpurun_out/pm.log 2>&1 || tail -3 gpurun_out/pm.log
  python - "$var" <<'PY' >> gpurun_out/r04_sq_ablation.txt
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
rm -rf gpurun_out/pm
cat gpurun_out/r04_sq_ablation.txt
