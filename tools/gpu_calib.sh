#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration (tools/calib/calib.hip) under rocprofv3; summary -> gpurun_out/calib_summary.txt
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/calib
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/calib/$c -o pmc -- tools/calib/calib > gpurun_out/calib.$c.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/calib/trace -o trace -- tools/calib/calib > gpurun_out/calib.trace.log 2>&1
python - <<'PY' > gpurun_out/calib_summary.txt
import csv, glob, collections
GiB = 1 << 30
known = {"calib_read4": GiB, "calib_read16": GiB, "calib_gather": 1600000 * 48, "calib_write4": GiB}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/calib/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        if k in known:
            m = sum(v) / len(v)
            print(f"{c:10s} {k:14s} counter={m:12.1f} KB  = {m * 1024 / known[k]:.3f} x the {known[k]} bytes the kernel moves")
for f in glob.glob("gpurun_out/calib/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].split("(")[0]
        if n in known:
            print(f"time {n:14s} avg {float(r['AverageNs']) / 1e3:9.1f} us -> {known[n] / float(r['AverageNs']):8.1f} GB/s")
PY
cat gpurun_out/calib_summary.txt
find gpurun_out/calib -type f -size +1M -delete
