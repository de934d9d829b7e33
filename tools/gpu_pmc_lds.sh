#!/bin/bash
# LDS / issue counters of the fused kernel: tools/gpu_pmc_lds.sh <variant>
var=${1:-0}
export TMPDIR=/tmp
cd /root/repo; mkdir -p gpurun_out
rocprofv3 -L 2>/dev/null | grep -o -E "SQ_[A-Z_0-9]*" | sort -u > gpurun_out/avail_sq.txt
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
  rm -rf gpurun_out/pl
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pl -o pmc -- python bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-e2e --variant $var > gpurun_out/pl.log 2>&1 || tail -3 gpurun_out/pl.log
  python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pl/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:28]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "fused5" not in k and "forest5" not in k: continue
    print(k, "  ".join(f"{c[3:]}={sum(v)/len(v)/1e6:.2f}M" for c, v in sorted(d.items())))
PY
done
rm -rf gpurun_out/pl
