#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the fused kernel under its ablation bits (what each part of a pass reads from HBM)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for var in 0 524288 262144 $((262144+524288)); do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/fa
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/fa -o pmc -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-e2e --variant $var > gpurun_out/fa.log 2>&1
    python - "$var" "$c" <<'PY'
import csv, glob, sys, collections
var, c = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/fa/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:24]].append(float(r["Counter_Value"]))
print(f"variant {var:>7} {c}: " + "  ".join(f"{k}={sum(v)/len(v)/1024:.1f} MB" for k, v in sorted(agg.items()) if "ugvc" in k))
PY
  done
done
rm -rf gpurun_out/fa
