"""Timing of the two users of the library's own radix sort / scan (csrc/kernels_prims.hip): the cumulative PR curve
(ugvc_pr_curve, 5 M rows) and the SEC database build (ugvc_sec_db_build, 8 M observations -> ~2 M loci), each against
its numpy statement.  Usage: python tools/bench_prims.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from variantcalling_amd import evaluate  # noqa: E402
from variantcalling_amd.engine import Engine  # noqa: E402

rng = np.random.default_rng(1)
n = 5_000_000
score = rng.random(n)
score[::11] = np.round(score[::11], 2)
cls = rng.choice(np.array([0, 1, 1, 1, 2], np.uint8), n)
with Engine(0) as eng:
    eng.pr_curve(score[:1000], cls[:1000], 10, 10, 10)
    best, dev = 1e9, 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        out = eng.pr_curve(score, cls, int((cls == 1).sum()), int((cls == 2).sum()), 1234)
        best = min(best, time.perf_counter() - t0)
        dev = out[5]
    t0 = time.perf_counter()
    order = np.argsort(score, kind="stable")
    ctp, cfp = np.cumsum(cls[order] == 1), np.cumsum(cls[order] == 2)
    i_tp, i_fp = int((cls == 1).sum()), int((cls == 2).sum())
    rec = evaluate.get_recall(1234 + ctp, i_tp - ctp, np.nan)
    prec = evaluate.get_precision(i_fp - cfp, i_tp - ctp, np.nan)
    f1 = evaluate.get_f1(prec, rec)
    t_np = time.perf_counter() - t0
    same = np.array_equal(out[0], score[order]) and np.array_equal(out[1], rec, equal_nan=True) and np.array_equal(out[3], f1, equal_nan=True)
    print(f"ugvc_pr_curve, {n} rows: device {dev:.2f} ms (sort + scan + finish), call incl. PCIe {best * 1e3:.1f} ms; numpy {t_np * 1e3:.0f} ms; bit-equal {same}")
    loci = np.unique(rng.integers(1, 1 << 40, 2_000_000).astype(np.uint64))
    keys = loci[rng.integers(0, loci.size, 8_000_000)]
    counts = rng.integers(0, 60, size=(keys.size, 3)).astype(np.int32)
    eng.sec_db_build(keys[:1000], counts[:1000])
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        uk, ue = eng.sec_db_build(keys, counts)
        best = min(best, time.perf_counter() - t0)
    t0 = time.perf_counter()
    o = np.argsort(keys, kind="stable")
    ks = keys[o]
    head = np.r_[True, ks[1:] != ks[:-1]]
    sums = np.add.reduceat(counts[o].astype(np.int64), np.flatnonzero(head), axis=0)
    t_np = time.perf_counter() - t0
    print(f"ugvc_sec_db_build, {keys.size} observations -> {uk.size} loci: call incl. PCIe {best * 1e3:.1f} ms; numpy {t_np * 1e3:.0f} ms; "
          f"equal {np.array_equal(uk, ks[head]) and np.array_equal(ue, sums.astype(np.int32))}")
