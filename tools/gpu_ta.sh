#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --list-avail 2>/dev/null | grep -E "Counter_Name|^\s+Name" | grep -E "TA_|TCP_" | cut -c1-100 | head -70 > gpurun_out/avail_ta.txt
wc -l gpurun_out/avail_ta.txt
CMD="python bench.py --steps 4 --warmup 1 --cpu-sample 0 --variant 128"
rm -rf gpurun_out/ta
i=0
for grp in "TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/ta/g$i -o pmc -- $CMD > gpurun_out/ta.g$i.log 2>&1
  tail -1 gpurun_out/ta.g$i.log | cut -c1-200
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/ta/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "ugvc" not in k: continue
    print("##", k)
    for c, v in sorted(d.items()):
        print(f"   {c:36s} mean={sum(v)/len(v):16.1f} n={len(v)}")
PY
find gpurun_out/ta -type f -size +1M -delete
