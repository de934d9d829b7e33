#!/usr/bin/env python3
"""Per-wave clocks of the fused kernel's last pass (UGVC_WAVE_CLK=<file> bench.py ...): when each wave of a workgroup reached
its first tile and when it finished, relative to the workgroup's first entry, by wave index (mean / max over workgroups)."""
import sys

import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16, 4).astype(np.int64)
entry, first, end, info = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
tiles, role = info & 0xFFFFFFFF, info >> 32
live = end > 0
t0 = np.where(live, entry, np.iinfo(np.int64).max).min(axis=1, keepdims=True)
print(f"{a.shape[0]} workgroups; ticks relative to the workgroup's first wave entry (s_memtime)")
print("wave role tiles(mean)  first-tile mean   end mean    end p95    end max   ticks/tile")
for w in range(16):
    m = live[:, w]
    if not m.any():
        print(f"{w:4d}  -");
        continue
    f, e = (first - t0)[:, w][m], (end - t0)[:, w][m]
    tl = tiles[:, w][m]
    print(f"{w:4d} {'indel' if role[:, w][m].max() else 'snp  '} {tl.mean():8.1f} {f.mean():14.0f} {e.mean():11.0f} {np.percentile(e, 95):10.0f} {e.max():10.0f} {((e - f) / np.maximum(tl, 1)).mean():11.0f}")
wg_end = np.where(live, end - t0, 0).max(axis=1)
print(f"workgroup end: mean {wg_end.mean():.0f}  p5 {np.percentile(wg_end, 5):.0f}  p95 {np.percentile(wg_end, 95):.0f}  max {wg_end.max():.0f}")
gl = np.where(live, end, 0).max() - np.where(live, entry, np.iinfo(np.int64).max).min()
print(f"kernel span (first entry to last end over all workgroups): {gl}")
last = np.where(live, end - t0, 0).argmax(axis=1)
print("last wave to finish, count by wave index:", np.bincount(last, minlength=16).tolist())
# ---- per workgroup: who is slow, and is it the data or the place
abs0 = np.where(live, entry, np.iinfo(np.int64).max).min()
wg_start = np.where(live, entry, np.iinfo(np.int64).max).min(axis=1) - abs0
wg_abs_end = np.where(live, end, 0).max(axis=1) - abs0
wg_tiles = tiles.sum(axis=1)
wg_indel = (info >> 32).sum(axis=1)
print(f"workgroup start (first wave entry after the launch's first): mean {wg_start.mean():.0f}  max {wg_start.max():.0f}")
print(f"workgroup absolute end: mean {wg_abs_end.mean():.0f}  p95 {np.percentile(wg_abs_end, 95):.0f}  max {wg_abs_end.max():.0f}")
print("by blockIdx % 8 (XCD): mean relative end  |  mean start")
for x in range(8):
    m = np.arange(a.shape[0]) % 8 == x
    print(f"  xcd {x}: {wg_end[m].mean():9.0f} | {wg_start[m].mean():7.0f}")
order = np.argsort(-wg_end)
print("slowest workgroups: blockIdx, relative end, tiles, indel tiles")
for b in order[:12]:
    print(f"  {b:4d} {wg_end[b]:9.0f} {wg_tiles[b]:5d} {wg_indel[b]:4d}")
print("fastest:")
for b in order[-6:]:
    print(f"  {b:4d} {wg_end[b]:9.0f} {wg_tiles[b]:5d} {wg_indel[b]:4d}")
c = np.corrcoef(wg_end, wg_tiles)[0, 1]
ci = np.corrcoef(wg_end, wg_indel)[0, 1]
print(f"correlation of a workgroup's end with its tiles {c:.2f}, with its indel tiles {ci:.2f}")
# quarter by quarter of the grid
q = a.shape[0] // 4
print("mean relative end by quarter of the grid (genome order):", [int(wg_end[k * q:(k + 1) * q].mean()) for k in range(4)])

# ---- the pass before, if it was kept: do the same workgroups finish late again?
import os
if os.path.exists(sys.argv[1] + ".prev"):
    b = np.fromfile(sys.argv[1] + ".prev", dtype=np.uint64).reshape(-1, 16, 4).astype(np.int64)
    if b.shape == a.shape:
        lb = b[..., 2] > 0
        tb = np.where(lb, b[..., 0], np.iinfo(np.int64).max).min(axis=1, keepdims=True)
        eb = np.where(lb, b[..., 2] - tb, 0).max(axis=1)
        print(f"correlation of a workgroup's end with its end in the previous pass: {np.corrcoef(wg_end, eb)[0, 1]:.2f}")
        slow_a, slow_b = set(np.argsort(-wg_end)[:26].tolist()), set(np.argsort(-eb)[:26].tolist())
        print(f"of the 26 slowest workgroups, {len(slow_a & slow_b)} are among the 26 slowest of the previous pass")
        # per-wave correlation inside a workgroup: is it one wave that is late, or the whole workgroup?
        ea = np.where(live, end - t0, 0)
        print("mean over workgroups of (workgroup end - median wave end):", int(np.nanmean(wg_end - np.nanmedian(np.where(live, ea, np.nan), axis=1))))
# ---- the slowest workgroups wave by wave, beside a median one
med = int(np.argsort(wg_end)[len(wg_end) // 2])
for b in list(order[:3]) + [med]:
    ea = np.where(live[b], end[b] - t0[b, 0], 0)
    fa = np.where(live[b], first[b] - t0[b, 0], 0)
    print(f"workgroup {b} (end {wg_end[b]}): wave: first-tile / end / tiles(indel)")
    print("   " + "  ".join(f"{w}:{fa[w] // 1000}k/{ea[w] // 1000}k/{tiles[b, w]}({(info[b, w] >> 32)})" for w in range(16)))
