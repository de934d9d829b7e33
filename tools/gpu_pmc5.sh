#!/bin/bash
# instruction counters of the fused kernel under an ablation: tools/gpu_pmc5.sh <tag> <variant>
tag=$1; var=$2
export TMPDIR=/tmp
cd /root/repo; mkdir -p gpurun_out
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/$tag -o pmc -- python bench.py --steps 4 --warmup 1 --cpu-sample 0 --variant $var > gpurun_out/$tag.log 2>&1
python - "$tag" "$var" <<'PY'
import csv, glob, sys, collections
tag, var = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:28]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "fused5" not in k: continue
    print(f"variant {var}: " + "  ".join(f"{c[3:]}={sum(v)/len(v)/1e6:.2f}M" for c, v in sorted(d.items())))
PY
rm -rf gpurun_out/$tag
