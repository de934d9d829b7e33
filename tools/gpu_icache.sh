#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --list-avail 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_LEVEL|SQC_" | head -40 > gpurun_out/avail_icache.txt
cat gpurun_out/avail_icache.txt | cut -c1-160
CMD="python bench.py --steps 4 --warmup 1 --cpu-sample 0"
rm -rf gpurun_out/icache
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/icache/g1 -o pmc -- $CMD > gpurun_out/icache.g1.log 2>&1
tail -2 gpurun_out/icache.g1.log | cut -c1-300
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/icache/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "ugvc" not in k: continue
    print("##", k)
    for c, v in sorted(d.items()):
        print(f"   {c:24s} mean={sum(v)/len(v):16.1f} n={len(v)}")
PY
find gpurun_out/icache -type f -size +1M -delete
