#!/usr/bin/env python
"""Headline benchmark: variants/sec filtered on a 5 M-call WGS-shaped callset (BASELINE.json).

One "step" = one scoring pass (featurize -> lookup -> quantise kernel, then the LDS-resident
forest kernel; featurize -> lookup -> score -> FILTER) over the whole callset (C3: 5 M SNV+indel, 3.1 Gb genome, runs + 3 annotation tracks, 1 M-locus
blacklist, 40-tree depth-8 forest per variant-type group), inputs already resident in HBM.
With --gpus N > 1 (launched by torch.distributed.run, one process per GPU) the SAME callset is
cut into N equal-count shards ("strong" scaling, BASELINE.json config C4) and every step ends
with the RCCL all-gather of the (tree_score, filter, flags) columns over xGMI.

Prints ONE JSON line on rank 0.  roofline.achieved = 121.6 algorithmic bytes/variant
 x variants per launch / mean kernel launch duration measured
with HIP events on the launch stream inside the timed region.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

ALG_BYTES_C3 = 121.6       # B/variant, fused featurize+lookup+score, SNV+indel
ALG_BYTES_C2 = 107.6       # SNV-only
HBM_PEAK_GBPS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec
MODEL = "rf_model_ignore_gt_incl_hpol_runs"


def cpu_baseline(cs, forests, sample_n):
    """Oracle restatement in the reference idiom (pandas per-row apply + tree scoring), one
    process, on the first `sample_n` variants of the same callset.  Checker code only."""
    from oracle import idiom, oracle as O
    sub = cs.variants.slice(0, sample_n)
    fa = idiom.PyFasta(cs.ref)
    for c in np.unique(sub.contig):       # "open the FASTA" outside the timed region
        fa[int(c)]
    t0 = time.perf_counter()
    idiom.filter_variants_idiom(sub, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests, fasta=fa)
    t_idiom = time.perf_counter() - t0
    t0 = time.perf_counter()
    O.filter_variants(sub, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
    t_vec = time.perf_counter() - t0
    return dict(value=sub.n / t_idiom, unit="variants/s", cores=1, kind="port",
                sample=f"first {sub.n} variants of the same callset; oracle/idiom.py (pandas per-row apply, "
                       f"reference idiom B0) single process, {t_idiom:.1f} s",
                vectorised_numpy_value=sub.n / t_vec, vectorised_numpy_seconds=round(t_vec, 2))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--variants", type=int, default=5_000_000)
    ap.add_argument("--snv-only", action="store_true", help="C2 shape instead of C3")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--cpu-sample", type=int, default=100_000, help="variants timed on the CPU baseline (0 = skip)")
    ap.add_argument("--variant", type=int, default=0, help="kernel variant (debug)")
    args = ap.parse_args()

    from variantcalling_amd import dist, model_io, shard, synth
    from variantcalling_amd.engine import Engine, configure

    grp = dist.Group()
    if grp.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={grp.world}: launch with "
                         f"python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus}")
    t_setup = time.perf_counter()
    n_req = args.variants * (grp.world if args.scaling == "weak" else 1)
    cs = synth.make_callset(n_req, snv_only=args.snv_only)
    n_total = cs.variants.n
    forests = model_io.load_models(os.path.join(ROOT, "tests", "golden", "synth_rf_v1.npz"))[MODEL]
    mine = shard.shard_of(cs.variants, grp.rank, grp.world)
    cap = shard.shard_cap(n_total, grp.world)

    eng = Engine(grp.local_rank)
    info = eng.device_info()
    configure(eng, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests, "TGCA", 10, 10, True)
    eng.set_kernel_variant(args.variant)
    eng.upload_variants(mine)
    gather = grp.world > 1
    if gather:
        uid = grp.broadcast_bytes(eng.comm_unique_id() if grp.rank == 0 else None, 0)
        eng.comm_init(uid, grp.rank, grp.world)
    t_setup = time.perf_counter() - t_setup

    # ---- warm-up (untimed)
    if args.warmup > 0:
        eng.timed_steps(args.warmup, cap, gather)
    # ---- timed region: exactly K steps between barrier + device sync on both sides
    eng.device_sync()
    grp.barrier()
    t0 = time.perf_counter()
    ms_total, ms_kernel = eng.timed_steps(args.steps, cap, gather)
    eng.device_sync()
    grp.barrier()
    wall = grp.max_float(time.perf_counter() - t0)
    ms_kernel_max = grp.max_float(ms_kernel)

    # ---- post-run correctness spot check against the oracle (rank 0, small slice; untimed).  With a
    # collective the timed passes wrote straight into the gather buffers: one more pass fills the
    # resident result columns this check reads.
    if gather:
        eng.filter_resident()
    res = eng.download_results()
    check = None
    if grp.rank == 0:
        from oracle import oracle as O
        k = min(5000, mine.n)
        sub = mine.slice(0, k)
        exp = O.filter_variants(sub, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
        check = bool(np.array_equal(res.filter[:k], exp.filter) and np.array_equal(res.flags[:k], exp.flags)
                     and np.array_equal(res.tree_score[:k], exp.tree_score))
    if gather:
        b = shard.shard_bounds(n_total, grp.world)
        allr = eng.gathered_download(cap, grp.world, [int(b[r + 1] - b[r]) for r in range(grp.world)])
        lo = int(b[grp.rank])
        ok = bool(allr.filter.size == n_total and np.array_equal(allr.filter[lo:lo + mine.n], res.filter)
                  and np.array_equal(allr.tree_score[lo:lo + mine.n], res.tree_score))
        ok_all = grp.sum_float(1.0 if ok else 0.0) == grp.world
    else:
        ok_all = True

    if grp.rank == 0:
        alg = ALG_BYTES_C2 if args.snv_only else ALG_BYTES_C3
        kern_ms = ms_kernel_max / args.steps
        achieved = alg * mine.n / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as fh:
                traffic = json.load(fh).get("bytes_per_launch_5M")
        out = dict(
            metric="variants/sec filtered (whole node), 5M-call WGS", value=n_total * args.steps / wall,
            unit="variants/s", n_gpus=grp.world, steps=args.steps, warmup=args.warmup,
            ms_per_step=wall / args.steps * 1e3, higher_is_better=True, scaling=args.scaling,
            vs_baseline=None, dtype="u8/i32 featurize + f32 compare + f64 accumulate", data="synthetic",
            config=dict(workload=("C2 " if args.snv_only else "C3 ") + f"{n_total} variants "
                        f"({'SNV-only' if args.snv_only else '82% SNV / 18% indel'}), 3.1 Gb 24-contig genome tiled from "
                        "real hg38 chr1 blocks, runs + 3 annotation tracks (5.0M intervals), 1M-locus blacklist, "
                        "RF 40 trees depth 8 x 3 groups, F=20; fused featurize+lookup+score+FILTER, inputs resident in HBM",
                        variants_per_gpu=mine.n, model=MODEL, sharding=f"equal-count x{grp.world}",
                        collective="RCCL all-gather (score f32, filter u8, flags u8)" if gather else "none",
                        device=info["name"], kernel_variant=args.variant),
            roofline=dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBPS, unit="GB/s",
                          frac=achieved / HBM_PEAK_GBPS, traffic=traffic,
                          kernel="one scoring pass = bracket3_kernel + featurize3_kernel + forest3_kernel "
                                 "(HIP events around the three launches on the context stream)",
                          kernel_ms=kern_ms, alg_bytes_per_variant=alg,
                          variants_per_launch=mine.n),
            parity=dict(oracle_slice_bit_exact=check, gather_consistent=ok_all),
            setup_s=round(t_setup, 1))
        if grp.world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(cs, forests, min(args.cpu_sample, n_total))
            out["cpu_baseline"]["host_cores_available"] = os.cpu_count()
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    eng.close()
    grp.close()


if __name__ == "__main__":
    main()
