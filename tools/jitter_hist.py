"""Per-dispatch durations of the scoring pass from a rocprofv3 kernel trace (where does the 13 % p5-p95
spread of a resident, identical input come from?).  Usage: python tools/jitter_hist.py <dir with *kernel_trace.csv> [skip]
Prints, per kernel, the distribution of dispatch durations and of the gaps to the previous dispatch, and the duration
as a function of the position in the run (clock ramp shows as a trend, contention as outliers)."""
import csv
import glob
import os
import sys

import numpy as np

d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
files = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))
if not files:
    sys.exit("no kernel_trace.csv under " + d)
rows = list(csv.DictReader(open(files[0])))
by = {}
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    by.setdefault(name, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for name, ev in by.items():
    if "fused5" not in name and "forest5" not in name:
        continue
    ev.sort()
    ev = ev[skip:]
    dur = np.array([e - s for s, e in ev]) / 1e3
    print(f"## {name}: {dur.size} dispatches; duration us: min {dur.min():.1f} p5 {np.percentile(dur, 5):.1f} p50 {np.percentile(dur, 50):.1f} "
          f"p95 {np.percentile(dur, 95):.1f} max {dur.max():.1f} mean {dur.mean():.1f} sd {dur.std():.1f}")
    lo, hi = np.floor(dur.min() / 5) * 5, np.ceil(dur.max() / 5) * 5
    h, edges = np.histogram(dur, bins=np.arange(lo, hi + 5, 5))
    for c, e in zip(h, edges):
        if c:
            print(f"   {e:7.0f}-{e + 5:<5.0f} us {'#' * int(np.ceil(60 * c / h.max()))} {c}")
    q = max(dur.size // 5, 1)
    print("   mean by fifth of the run:", " ".join(f"{dur[k * q:(k + 1) * q].mean():.1f}" for k in range(5) if dur[k * q:(k + 1) * q].size))
    if dur.size > 2:
        ac = np.corrcoef(dur[:-1], dur[1:])[0, 1]
        print(f"   lag-1 autocorrelation of the durations: {ac:.2f} (near 0: independent dispatch-to-dispatch noise; near 1: slow drift)")
