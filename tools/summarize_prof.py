"""Condense rocprofv3 output directories into a small text summary (per-kernel stats + PMC)."""
import csv
import glob
import os
import sys

d = sys.argv[1]
for f in sorted(glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)):
    print("## kernel stats:", f)
    for row in list(csv.reader(open(f)))[:12]:
        print(",".join(row))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in sorted(glob.glob(os.path.join(d, f"pmc_{c}", "**", "*counter_collection.csv"), recursive=True)):
        rows = list(csv.DictReader(open(f)))
        agg = {}
        for r in rows:
            k = r.get("Kernel_Name", "?")[:70]
            agg.setdefault(k, []).append(float(r.get("Counter_Value", 0)))
        print(f"## {c}: {f}")
        for k, v in agg.items():
            print(f"{k}: launches={len(v)} mean={sum(v) / len(v):.1f} min={min(v):.1f} max={max(v):.1f}")
