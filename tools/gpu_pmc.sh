#!/bin/bash
# SQ / LDS counters of the default bench command (one --pmc pass per counter group).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=${1:-pmc}
shift || true
CMD="python bench.py --steps 6 --warmup 2 --cpu-sample 0 $*"
rm -rf gpurun_out/$TAG
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/$TAG/g$i -o pmc -- $CMD > gpurun_out/$TAG.g$i.log 2>&1
done
python - <<'PY' "$TAG"
import csv, glob, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "ugvc" not in k: continue
    print("##", k)
    for c, v in sorted(d.items()):
        print(f"   {c:24s} mean={sum(v)/len(v):16.1f} n={len(v)}")
PY
find gpurun_out/$TAG -type f -size +1M -delete
