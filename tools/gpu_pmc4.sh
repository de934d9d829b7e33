#!/bin/bash
# SQ instruction counters of the v4 featurize kernel (same counter groups as tools/gpu_pmc.sh, known to collect)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
CMD="python bench.py --steps 4 --warmup 1 --cpu-sample 0 --variant 128"
rm -rf gpurun_out/pmc4
timeout 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc4/g1 -o pmc -- $CMD > gpurun_out/pmc4.g1.log 2>&1
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc4/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "ugvc" not in k: continue
    print("##", k)
    for c, v in sorted(d.items()):
        print(f"   {c:24s} mean={sum(v)/len(v):16.1f} n={len(v)}")
PY
find gpurun_out/pmc4 -type f -size +1M -delete
