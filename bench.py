#!/usr/bin/env python
"""Headline benchmark: variants/sec filtered on a 5 M-call WGS-shaped callset (BASELINE.json).

Default workload `filter` (config C3): one "step" = one scoring pass over the whole callset - featurize ->
lookup -> score -> FILTER: the two launches of csrc/kernels_v5.hip (fused5 -> forest5) -
on 5 M SNV+indel, 3.1 Gb genome, runs + 3 annotation tracks, 1 M-locus blacklist, 40-tree depth-8 forest per
variant-type group, inputs already resident in HBM.  With --gpus N > 1 (launched by torch.distributed.run, one
process per GPU; the ranks rendezvous over plain TCP, variantcalling_amd/dist.py) the SAME callset is cut into N
equal-count shards ("strong" scaling, BASELINE.json config C4) and every step ends with the RCCL all-gather of the
(tree_score, filter, flags) columns over xGMI.

Other workloads (same JSON shape, their own algorithmic bytes / flops and roofline; N = 1):
  --workload c2          1 M SNV-only callset (config C2)
  --workload pileup      a11 pileup tally, 5 M loci x ~30 observations, 84 B/locus
  --workload sec_apply   SEC database apply on the 5 M resident calls, ~26 B/call
  --workload c5_gemm     config C5: T = 100 depth-6 ensemble as a leaf-matrix GEMM on MFMA (int8), 2 M variants

Prints ONE JSON line on rank 0.  roofline.achieved = algorithmic bytes x units per
launch / mean launch duration measured with HIP events on the launch stream inside the timed region.
"""
import argparse
import json
import os
import subprocess
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

ALG_BYTES_C3 = 121.6       # B/variant, fused featurize+lookup+score, SNV+indel
ALG_BYTES_C2 = 107.6       # SNV-only
ALG_BYTES_PILEUP = 84.0    # B/locus
ALG_BYTES_SEC = 26.0       # B/call: key 8 + counts 12 + depth columns / flags 6
HBM_PEAK_GBPS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_I8_PEAK_TOPS = 3944.0  # dense int8 MFMA: "≥ 3 944 TOPS measured" (MI355X_MICROARCH.md) - the guide's figure;
MFMA_I8_SPEC_TOPS = 5000.0  # the datasheet's dense fp8 / int8 rate, reported beside it
MODEL = "rf_model_ignore_gt_incl_hpol_runs"
PASS_KERNELS = ("one scoring pass = fused5_kernel + forest5_kernel "
                "(csrc/kernels_v5.hip; HIP events around the two launches on the context stream)")

_CPU_JOB = {}


def _cpu_chunk(bounds):
    """One worker of the multi-process CPU baseline: the reference-idiom restatement on rows [lo, hi)."""
    from oracle import idiom
    lo, hi = bounds
    job = _CPU_JOB
    sub = job["vt"].slice(lo, hi)
    cs = job["cs"]
    idiom.filter_variants_idiom(sub, cs.ref, cs.runs, cs.tracks, cs.blacklist, job["forests"], fasta=job["fa"])
    return hi - lo


def _cpu_noop(k):
    return k


def _cpu_chunk_vec(bounds):
    """One worker of the strongest CPU formulation: the vectorised numpy oracle on rows [lo, hi)."""
    from oracle import oracle as O
    lo, hi = bounds
    job = _CPU_JOB
    cs = job["cs"]
    O.filter_variants(job["vt"].slice(lo, hi), cs.ref, cs.runs, cs.tracks, cs.blacklist, job["forests"])
    return hi - lo


def cpu_baseline(cs, forests, sample_n):
    """Oracle restatement in the reference idiom (pandas per-row apply + tree scoring): one process on the first
    `sample_n` variants (the reference tool is single-process), then one process per host core over contig-ordered
    chunks of a larger sample (the `--n_jobs` style of docs/run_comparison_pipeline.md:81).  Checker code only; runs
    BEFORE the GPU context exists (the workers are forked)."""
    import multiprocessing as mp

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
    # `out` starts as the single-process reference idiom; once the all-core runs below have finished the STRONGEST CPU
    # formulation (the vectorised oracle over every host core) becomes the top-level figure - that is what "the same box's
    # host cores" can do at best - and the others stay beside it under their own names
    out = dict(value=sub.n / t_idiom, unit="variants/s", cores=1, kind="port",
               sample=f"first {sub.n} variants of the same callset; oracle/idiom.py (pandas per-row apply, "
                      f"reference idiom B0) single process, {t_idiom:.1f} s",
               vectorised_numpy_value=sub.n / t_vec, vectorised_numpy_seconds=round(t_vec, 2))
    cores = os.cpu_count() or 1
    workers = max(1, min(cores, 256))
    n_multi = int(min(cs.variants.n, max(sample_n, sub.n / t_idiom * 3.0 * workers)))     # ~3 s of work per worker
    chunks = workers                   # one chunk per process: each pays the tool's per-run table preparation once
    edges = np.linspace(0, n_multi, chunks + 1).astype(np.int64)
    for c in np.unique(cs.variants.contig[:n_multi]):   # "open the FASTA" once, before the fork and outside the timed region
        fa[int(c)]
    _CPU_JOB.update(vt=cs.variants, cs=cs, forests=forests, fa=fa)
    try:
        ctx = mp.get_context("fork")
        with ctx.Pool(workers) as pool:
            pool.map(_cpu_noop, range(4 * workers), chunksize=1)         # every worker forked and warm before anything is timed
            t0 = time.perf_counter()
            done = sum(pool.map(_cpu_chunk, [(int(edges[k]), int(edges[k + 1])) for k in range(chunks)], chunksize=1))
            t_multi = time.perf_counter() - t0
            out["multi"] = dict(value=done / t_multi, unit="variants/s", cores=workers,
                                sample=f"first {done} variants in {chunks} position-ordered chunks, one forked process per host core "
                                       f"({workers}), same idiom code, {t_multi:.1f} s wall (processes already started)")
            # the strongest CPU formulation: the VECTORISED oracle over all host cores on the whole callset (the
            # forked idiom run is slower than one core of this) - what "the same box's host cores" can do at best
            edges = np.linspace(0, cs.variants.n, workers + 1).astype(np.int64)
            t0 = time.perf_counter()
            done = sum(pool.map(_cpu_chunk_vec, [(int(edges[k]), int(edges[k + 1])) for k in range(workers)], chunksize=1))
            t_vm = time.perf_counter() - t0
            out["vectorised_multi"] = dict(value=done / t_vm, unit="variants/s", cores=workers,
                                           sample=f"all {done} variants in {workers} chunks over {workers} forked processes, oracle/oracle.py "
                                                  f"(vectorised numpy), {t_vm:.1f} s wall (processes already started)")
    except Exception as e:                # a box that cannot fork this much: report, do not fail the bench
        out.setdefault("multi", dict(value=None, error=repr(e)[:200]))
    finally:
        _CPU_JOB.clear()
    vm = out.get("vectorised_multi")
    if vm and vm.get("value"):
        single = {k: out[k] for k in ("value", "unit", "cores", "kind", "sample")}
        out["single_process_idiom"] = single
        out.update(value=vm["value"], cores=vm["cores"], kind="port", sample=vm["sample"],
                   what="strongest CPU formulation measured: the vectorised numpy oracle forked over every host core; the reference "
                        "tool's own idiom (single process, pandas per-row apply) is `single_process_idiom`, the same idiom over all "
                        "cores `multi`")
    return out


def _git_head():
    """Short commit hash of the tree the bench runs from (the GPU box gets a snapshot without .git: profiles/HEAD, written
    by whoever calls tools/gpu_evidence.sh before the snapshot is taken, stands in)."""
    try:
        h = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip()
        if h:
            return h
    except Exception:
        pass
    p = os.path.join(ROOT, "profiles", "HEAD")
    return open(p).read().strip() if os.path.exists(p) else None


def _traffic(workload, units, world=1):
    """HBM bytes per launch from the PMC passes (tools/make_traffic_json.py) - only for the workload, size and rank count
    they were measured on (key "<workload>:<units>:<world>" of profiles/hbm_traffic.json); anything else is None.
    Carries the commit it was measured at."""
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not os.path.exists(tpath):
        return None, None
    with open(tpath) as fh:
        d = json.load(fh)
    w = (d.get("workloads") or {}).get(f"{workload}:{int(units)}:{int(world)}")
    if not w:
        return None, None
    return w.get("bytes_per_launch"), w.get("commit")


def _device_clock_ghz(eng):
    """Peak engine clock of the device (hipDeviceProp_t::clockRate): what the issue-bound floor is priced at."""
    try:
        return eng.device_attr("clock_khz") / 1e6
    except Exception:
        return None


def _pct(ms, q):
    return float(np.percentile(ms, q)) if len(ms) else None


def other_workloads(eng, cs, forests_rf, budget_s=40.0):
    """Bounded measurements of the OTHER workloads inside the default run (the driver runs only `bench.py --gpus 1`,
    so C2 / C5 / pileup / SEC-apply had no driver-timed line): each a few launches behind the headline's timed region, HIP events
    on the context stream, the same kernels and algorithmic bytes as `--workload <w>` reports; nothing here touches the headline
    keys.  A measurement that fails or would overrun the budget is reported as such, never raised."""
    from variantcalling_amd import model_io, synth
    from variantcalling_amd.engine import configure
    t_begin = time.perf_counter()
    out = {}

    def left():
        return budget_s - (time.perf_counter() - t_begin)

    def guarded(name, f):
        if left() <= 0:
            out[name] = dict(skipped=f"budget of {budget_s:.0f} s spent")
            return
        t0 = time.perf_counter()
        try:
            out[name] = f()
            out[name]["wall_s"] = round(time.perf_counter() - t0, 2)
        except Exception as e:                                   # noqa: BLE001 - the headline line must still be printed
            out[name] = dict(error=repr(e)[:300])

    # ---- SEC apply on the resident calls (the 5 M calls of the headline pass are resident and scored)
    def sec():
        rng = np.random.default_rng(0)
        vk = cs.variants.keys()
        loci = np.unique(vk[rng.random(vk.size) < 0.4])
        keys = loci[rng.integers(0, loci.size, 4_000_000)]
        counts = rng.integers(0, 60, size=(keys.size, 3)).astype(np.int32)
        db_k, db_e = eng.sec_db_build(keys, counts)
        eng.set_sec_db(db_k, db_e)
        eng.timed_sec_apply(2)
        ms = eng.timed_sec_apply(10) / 10
        gbps = ALG_BYTES_SEC * cs.variants.n / (ms * 1e-3) / 1e9
        return dict(ms=ms, frac=gbps / HBM_PEAK_GBPS, achieved_gbps=gbps, bound="hbm", units=cs.variants.n, database_loci=int(db_k.size),
                    traffic=_traffic("sec_apply", cs.variants.n)[0], kernel="sec_apply_tiles_kernel<false, 3, 512>")
    guarded("sec_apply", sec)

    # ---- config C5 on the first 2 M calls of the same callset: feature build, leaf-matrix GEMM (one launch), row traversal
    def c5():
        xgb = model_io.load_models(os.path.join(ROOT, "tests", "golden", "synth_rf_v1.npz"))["xgb_model_ignore_gt_incl_hpol_runs"]
        sub = cs.variants.slice(0, min(2_000_000, cs.variants.n))
        eng.set_models(xgb)
        eng.upload_variants(sub)
        X, group = eng.feature_matrix()
        N, F = X.shape
        eng.timed_feature_matrix(2)
        fm_ms = eng.timed_feature_matrix(5)
        fm_bytes = ALG_BYTES_C3 + 4.0 * F
        fm_gbps = fm_bytes * N / (fm_ms * 1e-3) / 1e9
        rows_g = [np.flatnonzero(group == g).astype(np.int32) for g in range(3)]
        m3, gemm_ms = eng.forest_gemm3(rows_g, iters=5)
        trav_ms, same = 0.0, True
        for g in range(3):
            b, ms_b = eng.forest_gemm(g, rows_g[g], use_mfma=0, iters=3)
            trav_ms += ms_b
            same &= bool(np.array_equal(m3[rows_g[g]], b))
        tops = 2.0 * 64 * 64 * 100 * N / (gemm_ms * 1e-3) / 1e12
        return dict(units=N, feature_build=dict(ms=fm_ms, frac=fm_gbps / HBM_PEAK_GBPS, achieved_gbps=fm_gbps, bound="hbm",
                                               traffic=_traffic("c5_feature_build", N)[0]),
                    gemm3=dict(ms=gemm_ms, achieved_tops=tops, frac=tops / MFMA_I8_PEAK_TOPS, frac_of_spec_5000=tops / MFMA_I8_SPEC_TOPS, bound="mfma"),
                    traversal_ms=trav_ms, gemm_equals_traversal=same)
    guarded("c5", c5)

    # ---- config C2: 1 M SNV-only callset, the scoring pass (its own genome-sized tables)
    def c2():
        from oracle import oracle as O
        cs2 = synth.make_callset(1_000_000, snv_only=True)
        configure(eng, cs2.ref, cs2.runs, cs2.tracks, cs2.blacklist, forests_rf, "TGCA", 10, 10, True)
        eng.upload_variants(cs2.variants)
        eng.timed_steps(60, 0, False)
        tot, _ = eng.timed_steps(20, 0, False, per_step_events=False)
        ms = tot / 20
        gbps = ALG_BYTES_C2 * cs2.variants.n / (ms * 1e-3) / 1e9
        eng.filter_resident()
        res = eng.download_results()
        k = 20_000
        exp = O.filter_variants(cs2.variants.slice(0, k), cs2.ref, cs2.runs, cs2.tracks, cs2.blacklist, forests_rf)
        ok = bool(np.array_equal(res.filter[:k], exp.filter) and np.array_equal(res.flags[:k], exp.flags) and np.array_equal(res.tree_score[:k], exp.tree_score))
        return dict(ms=ms, frac=gbps / HBM_PEAK_GBPS, achieved_gbps=gbps, bound="hbm", units=cs2.variants.n,
                    traffic=_traffic("c2", cs2.variants.n)[0], oracle_rows_bit_exact=ok, oracle_rows_checked=k)
    guarded("c2", c2)

    # ---- a11 pileup tally: 2 M loci x ~30 observations
    def pileup():
        from oracle import oracle as O
        n_loci = 2_000_000
        off, obs = synth.make_pileup(n_loci, seed=5)
        eng.upload_pileup(off, obs)
        eng.timed_pileup(2)
        ms = eng.timed_pileup(10) / 10
        gbps = ALG_BYTES_PILEUP * n_loci / (ms * 1e-3) / 1e9
        exp = O.pileup_tally(off[:2001], obs[:off[2000]])
        got = eng.pileup_tally(off[:2001], obs[:off[2000]])
        ok = all(np.array_equal(got[k], exp[k]) for k in ("ref_fwd", "ref_rev", "alt_fwd", "alt_rev", "other", "dp", "bq_ref", "bq_alt"))
        return dict(ms=ms, frac=gbps / HBM_PEAK_GBPS, achieved_gbps=gbps, bound="hbm", units=n_loci, observations=int(obs.size),
                    traffic=None, oracle_slice_bit_exact=bool(ok))
    guarded("pileup", pileup)
    out["seconds"] = round(time.perf_counter() - t_begin, 1)
    out["what"] = ("bounded measurements of the other workloads behind the headline's timed region (same kernels, same algorithmic bytes as "
                   "`bench.py --workload <w>`; fewer launches, smaller pileup); frac = of the 8 TB/s HBM roofline, gemm3 of the 3 944 TOP/s "
                   "the guide measures for dense int8 MFMA")
    return out


def run_filter(args):
    from variantcalling_amd import dist, model_io, shard, synth
    from variantcalling_amd.engine import Engine, configure

    grp = dist.Group()
    if grp.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={grp.world}: launch with "
                         f"python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus}")
    snv_only = args.workload == "c2"
    n_var = 1_000_000 if (snv_only and args.variants == 5_000_000) else args.variants
    t_setup = time.perf_counter()
    n_req = n_var * (grp.world if args.scaling == "weak" else 1)
    cs = synth.make_callset(n_req, snv_only=snv_only)
    n_total = cs.variants.n
    forests = model_io.load_models(os.path.join(ROOT, "tests", "golden", "synth_rf_v1.npz"))[MODEL]
    cpu = None
    if grp.rank == 0 and grp.world == 1 and args.cpu_sample > 0:
        cpu = cpu_baseline(cs, forests, min(args.cpu_sample, n_total))       # before the GPU context: forks
        cpu["host_cores_available"] = os.cpu_count()
    # (cuts snapped to a contig change where that costs < 1 % imbalance: shard.shard_bounds - every rank holds the whole callset here)
    bounds = shard.shard_bounds(n_total, grp.world, cs.variants.contig)
    mine = cs.variants.slice(int(bounds[grp.rank]), int(bounds[grp.rank + 1]))
    cap = shard.shard_cap(n_total, grp.world, bounds)

    eng = Engine(grp.local_rank)
    info = eng.device_info()
    # every rank keeps only the part of the genome and of the side tables its shard can touch
    ctx_tables = shard.slice_context(cs.ref, cs.runs, cs.tracks, cs.blacklist, mine, hpol_dist=10) if grp.world > 1 else \
        (cs.ref, cs.runs, cs.tracks, cs.blacklist, mine)
    ref_r, runs_r, tracks_r, bl_r, mine_r = ctx_tables
    configure(eng, ref_r, runs_r, tracks_r, bl_r, forests, "TGCA", 10, 10, True)
    eng.set_kernel_variant(args.variant)
    eng.upload_variants(mine_r)
    gather = grp.world > 1
    if gather:
        uid = grp.broadcast_bytes(eng.comm_unique_id() if grp.rank == 0 else None, 0)
        # (RCCL prints a version banner on STDOUT when its first communicator comes up: this process's stdout is for ONE JSON
        # line, so file descriptor 1 points at stderr for the duration of the call)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            eng.comm_init(uid, grp.rank, grp.world)
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
        rccl = eng.comm_info()                                   # what RCCL itself reports: a silent one-rank run cannot pass for N
        if rccl["nranks"] != grp.world or rccl["rank"] != grp.rank:
            raise SystemExit(f"RCCL communicator reports {rccl}, launcher says rank {grp.rank} of {grp.world}")
    else:
        rccl = None
    t_setup = time.perf_counter() - t_setup

    # ---- spin-up (untimed, part of the setup): the chip's clock and power state take ~40 dispatches to settle - the
    # per-dispatch trace of a 200-step run (profiles/r03_jitter_hist.txt) shows the first fifth 3 % slower than the rest,
    # lag-1 autocorrelation 0.74: a ramp, not noise.  A production stream scores callset after callset; the timed region
    # below measures that state.  The ramp itself is reported (first / last 20 passes of the spin-up) in `spinup`.
    ramp = None
    probe = []                                                   # per-pass kernel times, each bracketed by its own event pair (untimed part)
    if args.spinup > 0:
        eng.timed_steps(args.spinup, cap, gather)
        sp = np.asarray(eng.last_step_ms(args.spinup), np.float64)
        k = min(20, sp.size)
        ramp = dict(passes=int(args.spinup), first_ms=float(sp[:k].mean()), last_ms=float(sp[-k:].mean()))
        probe += list(sp[-min(40, sp.size):])
    # ---- warm-up (untimed)
    if args.warmup > 0:
        eng.timed_steps(args.warmup, cap, gather)
        probe += list(np.asarray(eng.last_step_ms(args.warmup), np.float64))
    # ---- timed region: exactly K steps between barrier + device sync on both sides.  ONE event pair on the launch stream
    # brackets the K steps (per-step pairs are markers the stream drains to: ~9 us of every step - round 5; the per-pass
    # percentiles below come from the untimed passes just before, which keep their pairs)
    eng.device_sync()
    grp.barrier()
    t0 = time.perf_counter()
    ms_total, ms_kernel = eng.timed_steps(args.steps, cap, gather, per_step_events=args.step_events)
    eng.device_sync()
    grp.barrier()
    wall = grp.max_float(time.perf_counter() - t0)
    ms_kernel_max = grp.max_float(ms_kernel)
    step_ms = eng.last_step_ms(args.steps) if args.step_events else np.asarray(probe, np.float64)
    if gather and not args.step_events:
        # with a collective the single pair also spans the wait for the last gather: the kernel's own duration is the mean of
        # the bracketed passes just before the timed region (same kernels, same buffers)
        ms_kernel_max = grp.max_float(float(np.mean(probe)) * args.steps if len(probe) else ms_kernel)

    # ---- post-run correctness spot check against the oracle (rank 0, small slice; untimed).  With a
    # collective the timed passes wrote straight into the gather buffers: one more pass fills the
    # resident result columns this check reads.
    if gather:
        eng.filter_resident()
    res = eng.download_results()
    check = None
    gathered_ok = None
    checked_rows = 0
    if grp.rank == 0:
        # --check-rows K: the first K and the last K rows of this rank's shard (all of it if that covers it; K < 0: all),
        # in chunks the vectorised oracle digests in seconds
        from oracle import oracle as O
        k = mine.n if args.check_rows < 0 else min(args.check_rows, mine.n)
        spans = [(0, mine.n)] if 2 * k >= mine.n else [(0, k), (mine.n - k, mine.n)]
        check = True
        for lo, hi in spans:
            for a in range(lo, hi, 250_000):
                b_ = min(a + 250_000, hi)
                exp = O.filter_variants(mine.slice(a, b_), cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
                check = check and bool(np.array_equal(res.filter[a:b_], exp.filter) and np.array_equal(res.flags[a:b_], exp.flags)
                                       and np.array_equal(res.tree_score[a:b_], exp.tree_score))
                checked_rows += b_ - a
    if gather:
        b = bounds
        allr = eng.gathered_download(cap, grp.world, [int(b[r + 1] - b[r]) for r in range(grp.world)])
        lo = int(b[grp.rank])
        ok = bool(allr.filter.size == n_total and np.array_equal(allr.filter[lo:lo + mine.n], res.filter)
                  and np.array_equal(allr.tree_score[lo:lo + mine.n], res.tree_score))
        ok_all = grp.sum_float(1.0 if ok else 0.0) == grp.world
        # --check-rows -1 under a collective: rank 0 also compares EVERY row of the gathered callset with the oracle on the
        # unsharded tables (tools/scale_selfcheck.sh) - the shards, the per-rank table slices and the rank order of the gather
        if grp.rank == 0 and args.check_rows < 0:
            from oracle import oracle as O
            gathered_ok = True
            for a in range(0, n_total, 250_000):
                b_ = min(a + 250_000, n_total)
                exp = O.filter_variants(cs.variants.slice(a, b_), cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
                gathered_ok = gathered_ok and bool(np.array_equal(allr.filter[a:b_], exp.filter) and np.array_equal(allr.flags[a:b_], exp.flags)
                                                   and np.array_equal(allr.tree_score[a:b_], exp.tree_score))
    else:
        ok_all = True
    # ---- PCIe-inclusive rate of the host-buffer boundary (never `value`): upload + one pass + download
    e2e = None
    if grp.world == 1 and not args.no_e2e:
        from variantcalling_amd import schema as S
        keep = S.FilterResult(np.zeros(mine.n, np.float32), np.zeros(mine.n, np.uint8), np.zeros(mine.n, np.uint8))
        eng.filter_variants(mine, out=keep)
        ts, ts_fresh = [], []
        for _ in range(5):
            t1 = time.perf_counter()
            eng.filter_variants(mine, out=keep)                 # the caller's result arrays, reused from call to call
            ts.append(time.perf_counter() - t1)
        for _ in range(3):
            t1 = time.perf_counter()
            eng.filter_variants(mine)                           # three fresh numpy arrays per call (first-touch faults + munmap)
            ts_fresh.append(time.perf_counter() - t1)
        same = bool(np.array_equal(keep.filter, res.filter) and np.array_equal(keep.tree_score, res.tree_score))
        e2e = dict(value=mine.n / min(ts), unit="variants/s", ms=min(ts) * 1e3, ms_median=float(np.median(ts)) * 1e3,
                   ms_fresh_result_arrays=min(ts_fresh) * 1e3, equals_resident_pass=same,
                   what="ugvc_filter_variants: H2D of the variant columns from pageable numpy buffers + one pass + D2H of the "
                        "results into the caller's arrays (chunk pipeline, csrc/pipeline.hip)")

    if grp.rank == 0:
        alg = ALG_BYTES_C2 if snv_only else ALG_BYTES_C3
        kern_ms = ms_kernel_max / args.steps
        achieved = alg * mine.n / (kern_ms * 1e-3) / 1e9
        traffic, traffic_commit = _traffic("c2" if snv_only else "filter", n_total, grp.world)
        # The second roofline: the pass is bound by instruction ISSUE, not by HBM.  A tree-node visit is 4 vector
        # instructions (address, code address, compare, index update) + 2 LDS reads; a wave-64 vector instruction occupies
        # its 16-lane SIMD for 4 cycles, so 64 variants x one visit cost >= 16 cycles of one SIMD whatever else overlaps.
        # floor = visits / 64 x 16 cycles / (CUs x 4 SIMDs) / peak clock; visits = sum over variants of trees x depth.
        n_sub = int((mine.ref_len == mine.alt_len).sum())
        td = [int(f.n_trees) * int(f.max_depth) if f is not None else 0 for f in forests]
        td_indel = max(td[1:]) if len(td) > 1 else 0
        visits = float(n_sub) * td[0] + float(mine.n - n_sub) * td_indel
        clk_peak = _device_clock_ghz(eng)
        # the clock the pass SUSTAINS, read by the kernel itself (s_memtime against the constant 100 MHz s_memrealtime across one
        # wave of the last of 40 back-to-back passes: ugvc_pass_clock) - the floor is priced at it, the peak clock stays beside it
        clk_meas = None
        try:
            clk_meas = eng.pass_clock_ghz(40)[0] if args.variant == 0 and not gather else None
        except Exception:
            clk_meas = None
        clk = clk_meas or clk_peak
        issue = None
        if clk:
            floor_ms = visits / 64.0 * 16.0 / (info["n_cus"] * 4) / (clk * 1e9) * 1e3
            issue = dict(bound="valu_issue", visits_per_launch=visits, cycles_per_visit_floor=16,
                         what="4 vector instructions per tree-node visit x 4 issue cycles per wave-64 instruction on a 16-lane SIMD",
                         simds=info["n_cus"] * 4, clock_ghz=clk, clock_source=("measured inside the pass (s_memtime / s_memrealtime)" if clk_meas
                                                                                else "hipDeviceProp_t::clockRate (peak)"),
                         clock_ghz_peak=clk_peak, floor_ms=floor_ms, kernel_ms=kern_ms, frac=floor_ms / kern_ms)
        out = dict(
            metric="variants/sec filtered (whole node), 5M-call WGS", value=n_total * args.steps / wall,
            unit="variants/s", n_gpus=grp.world, steps=args.steps, warmup=args.warmup,
            ms_per_step=wall / args.steps * 1e3, higher_is_better=True, scaling=args.scaling,
            vs_baseline=None, dtype="u8/i32 featurize + f32 compare + f64 accumulate", data="synthetic",
            config=dict(workload=("C2 " if snv_only else "C3 ") + f"{n_total} variants "
                        f"({'SNV-only' if snv_only else '82% SNV / 18% indel'}), {cs.ref.codes.size / 1e9:.1f} Gb 24-contig genome tiled from "
                        f"real hg38 chr1 blocks, runs + 3 annotation tracks ({(cs.runs.starts.size + sum(t.starts.size for t in cs.tracks)) / 1e6:.1f}M "
                        f"intervals), {cs.blacklist.size / 1e6:.1f}M-locus blacklist, "
                        "RF 40 trees depth 8 x 3 groups, F=20; fused featurize+lookup+score+FILTER, inputs resident in HBM",
                        variants_per_gpu=mine.n, model=MODEL, sharding=f"equal-count x{grp.world}"
                        + (", cuts snapped to a contig change within 1 %, per-rank genome / side-table slices" if grp.world > 1 else ""),
                        collective="RCCL all-gather (score f32, filter u8, flags u8)" if gather else "none",
                        rccl_nranks=rccl["nranks"] if rccl else 1,
                        device=info["name"], kernel_variant=args.variant, commit=_git_head()),
            roofline=dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBPS, unit="GB/s",
                          frac=achieved / HBM_PEAK_GBPS, traffic=traffic, traffic_measured_at_commit=traffic_commit,
                          kernel=PASS_KERNELS, kernel_ms=kern_ms, kernel_ms_p5=_pct(step_ms, 5), kernel_ms_p50=_pct(step_ms, 50),
                          kernel_ms_p95=_pct(step_ms, 95),
                          kernel_ms_from=("one HIP event pair on the launch stream around the K timed steps / K" if not args.step_events and not gather
                                          else "mean of the per-pass event pairs of the untimed passes just before the timed region" if not args.step_events
                                          else "sum of the per-step event pairs of the timed region / K"),
                          percentiles_from=("the per-step event pairs of the timed region" if args.step_events else
                                            "per-pass event pairs of the warm-up steps and the last 40 spin-up passes (untimed)"),
                          alg_bytes_per_variant=alg, variants_per_launch=mine.n,
                          issue_bound=issue),
            e2e_incl_pcie=e2e, spinup=ramp, other_workloads=None,
            parity=dict(oracle_slice_bit_exact=check, oracle_rows_checked=checked_rows, gather_consistent=ok_all,
                        gathered_all_rows_bit_exact=gathered_ok),
            setup_s=round(t_setup, 1), cpu_baseline=cpu)
        # the other workloads (the driver's default run only): behind everything the headline keys are made of
        if grp.world == 1 and not args.no_other and not snv_only and args.variant == 0:
            out["other_workloads"] = other_workloads(eng, cs, forests)
        print(json.dumps(out), flush=True)
    eng.close()
    grp.close()


def _single_gpu_only(args):
    if args.gpus != 1 or int(os.environ.get("WORLD_SIZE", "1")) != 1:
        raise SystemExit(f"--workload {args.workload} is a single-GPU measurement (the path it times does not shard further)")


def _line(metric, value, unit, args, ms, workload, roofline, extra=None, dtype="i32"):
    out = dict(metric=metric, value=value, unit=unit, n_gpus=1, steps=args.steps, warmup=args.warmup, ms_per_step=ms,
               higher_is_better=True, scaling="weak", vs_baseline=None, dtype=dtype, data="synthetic",
               config=dict(workload=workload, commit=_git_head()), roofline=roofline, cpu_baseline=None)
    out.update(extra or {})
    print(json.dumps(out), flush=True)


def run_pileup(args):
    _single_gpu_only(args)
    from oracle import oracle as O
    from variantcalling_amd import synth
    from variantcalling_amd.engine import Engine
    n_loci = args.variants
    off, obs = synth.make_pileup(n_loci, seed=5)
    cpu = None
    if args.cpu_sample > 0:
        k = min(n_loci, max(args.cpu_sample, 200_000))
        t0 = time.perf_counter()
        O.pileup_tally(off[:k + 1], obs[:off[k]])
        dt = time.perf_counter() - t0
        cpu = dict(value=k / dt, unit="loci/s", cores=1, kind="port", sample=f"first {k} loci, oracle.pileup_tally (numpy), {dt:.1f} s")
    eng = Engine(0)
    eng.upload_pileup(off, obs)
    if args.warmup:
        eng.timed_pileup(args.warmup)
    eng.device_sync()
    t0 = time.perf_counter()
    ms = eng.timed_pileup(args.steps)
    eng.device_sync()
    wall = time.perf_counter() - t0
    kern_ms = ms / args.steps
    achieved = ALG_BYTES_PILEUP * n_loci / (kern_ms * 1e-3) / 1e9
    exp = O.pileup_tally(off[:2001], obs[:off[2000]])
    got = eng.pileup_tally(off[:2001], obs[:off[2000]])
    ok = all(np.array_equal(got[k], exp[k]) for k in ("ref_fwd", "ref_rev", "alt_fwd", "alt_rev", "other", "dp", "bq_ref", "bq_alt"))
    _line("pileup loci/sec tallied", n_loci * args.steps / wall, "loci/s", args, wall / args.steps * 1e3,
          f"a11 pileup tally: {n_loci} loci, {obs.size} observations (Poisson(30) deep), CSR resident in HBM",
          dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBPS, unit="GB/s", frac=achieved / HBM_PEAK_GBPS,
               traffic=_traffic("pileup", n_loci)[0], traffic_measured_at_commit=_traffic("pileup", n_loci)[1],
               kernel="pileup_kernel (csrc/kernels_aux.hip)", kernel_ms=kern_ms, alg_bytes_per_locus=ALG_BYTES_PILEUP),
          dict(parity=dict(oracle_slice_bit_exact=bool(ok)), cpu_baseline=cpu), dtype="u16 observations, i32 tallies, f64 SOR")
    eng.close()


def run_sec_apply(args):
    _single_gpu_only(args)
    from variantcalling_amd import model_io, synth
    from variantcalling_amd.engine import Engine, configure
    cs = synth.make_callset(args.variants)
    forests = model_io.load_models(os.path.join(ROOT, "tests", "golden", "synth_rf_v1.npz"))[MODEL]
    eng = Engine(0)
    configure(eng, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
    eng.upload_variants(cs.variants)
    eng.filter_resident()
    rng = np.random.default_rng(0)
    vk = cs.variants.keys()
    loci = np.unique(vk[rng.random(vk.size) < 0.4])
    keys = loci[rng.integers(0, loci.size, 8_000_000)]
    counts = rng.integers(0, 60, size=(keys.size, 3)).astype(np.int32)
    db_k, db_e = eng.sec_db_build(keys, counts)
    eng.set_sec_db(db_k, db_e)
    # CPU baseline (checker code, bounded sample): oracle/stats.py sec_apply - the scalar restatement of the reference's statistic
    # (ugvc/utils/stats_utils.py:12-70) looped over the calls, as the reference's per-record tools do
    cpu = None
    if args.cpu_sample > 0:
        from oracle import stats as OS
        k = min(cs.variants.n, max(20_000, args.cpu_sample // 5))
        v = cs.variants
        t0 = time.perf_counter()
        OS.sec_apply(vk[:k], v.dp[:k], v.ad_ref[:k], v.ad_alt[:k], db_k, db_e)
        dt = time.perf_counter() - t0
        cpu = dict(value=k / dt, unit="variants/s", cores=1, kind="port",
                   sample=f"first {k} calls of the same callset against the same database, oracle/stats.py sec_apply (scalar statistic per call), {dt:.1f} s")
    eng.timed_sec_apply(max(args.warmup, 1))
    eng.device_sync()
    t0 = time.perf_counter()
    ms_dev = eng.timed_sec_apply(args.steps)
    eng.device_sync()
    wall = time.perf_counter() - t0
    ms = wall / args.steps * 1e3
    kern_ms = ms_dev / args.steps
    achieved = ALG_BYTES_SEC * cs.variants.n / (kern_ms * 1e-3) / 1e9
    traffic, traffic_commit = _traffic("sec_apply", cs.variants.n)
    _line("SEC calls/sec tested against the cohort database", cs.variants.n * args.steps / wall, "variants/s", args, ms,
          f"SEC apply: {cs.variants.n} resident calls against {db_k.size} database loci (k = 3), verdict into the resident flags",
          dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBPS, unit="GB/s", frac=achieved / HBM_PEAK_GBPS, traffic=traffic,
               traffic_measured_at_commit=traffic_commit,
               kernel="sec_apply_tiles_kernel<false, 3, 512> (csrc/kernels_sec.hip); HIP events around the launches on the context stream",
               kernel_ms=kern_ms, alg_bytes_per_variant=ALG_BYTES_SEC), dict(cpu_baseline=cpu), dtype="i32 counts, f64 log-likelihoods")
    eng.close()


def run_c5(args):
    _single_gpu_only(args)
    from variantcalling_amd import model_io, synth
    from variantcalling_amd.engine import Engine, configure
    n = 2_000_000 if args.variants == 5_000_000 else args.variants
    cs = synth.make_callset(n)
    forests = model_io.load_models(os.path.join(ROOT, "tests", "golden", "synth_rf_v1.npz"))["xgb_model_ignore_gt_incl_hpol_runs"]
    eng = Engine(0)
    configure(eng, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
    eng.upload_variants(cs.variants)
    X, group = eng.feature_matrix()
    N = X.shape[0]
    eng.set_kernel_variant(args.variant)                        # (profiling bits of the feature-matrix launch; 0 in every reported run)
    # BASELINE.md C5 "feature-build GB/s": the resident N x F matrix built `steps` times back to back (device events, no
    # download); algorithmic bytes = the scoring pass's 121.6 B read per variant + 4 F written
    eng.timed_feature_matrix(2)
    fm_ms = eng.timed_feature_matrix(max(args.steps, 5))
    fm_bytes = ALG_BYTES_C3 + 4.0 * X.shape[1]
    fm_gbps = fm_bytes * N / (fm_ms * 1e-3) / 1e9
    tot_gemm2 = tot_trav = tot_gemm_r1 = 0.0
    same = True
    t0 = time.perf_counter()
    rows_g = [np.flatnonzero(group == g).astype(np.int32) for g in range(3)]
    # round 5: ONE launch for the three variant-type groups (forest_gemm3_kernel: exit leaf by a maximum, margins by row)
    m3, tot_gemm = eng.forest_gemm3(rows_g, iters=args.steps)
    for g in range(3):
        rows = rows_g[g]
        a, ms_a = eng.forest_gemm(g, rows, use_mfma=1, iters=max(args.steps // 2, 1))   # round-4 kernel, one launch per group: for the record
        b, ms_b = eng.forest_gemm(g, rows, use_mfma=0, iters=args.steps)
        c, ms_c = eng.forest_gemm(g, rows, use_mfma=2, iters=max(args.steps // 4, 1))   # round-1 kernel (predicates by LDS gathers), for the record
        same &= bool(np.array_equal(a, b) and np.array_equal(a, c) and np.array_equal(m3[rows], b))
        tot_gemm2 += ms_a
        tot_trav += ms_b
        tot_gemm_r1 += ms_c
    wall = time.perf_counter() - t0
    T, I, L = 100, 64, 64
    # CPU baseline (checker code, bounded sample): the oracle's feature rows + its tree traversal of the same ensemble
    cpu = None
    if args.cpu_sample > 0:
        from oracle import oracle as O
        k = min(N, max(50_000, args.cpu_sample))
        t0 = time.perf_counter()
        ft = O.featurize(cs.variants.slice(0, k), cs.ref, cs.runs, cs.tracks, "TGCA", 10, 10)
        Xo, go = ft["X"], ft["group"]
        for g in range(3):
            if forests[g] is not None and (go == g).any():
                O.forest_predict(forests[g], Xo[go == g])
        dt = time.perf_counter() - t0
        cpu = dict(value=k / dt, unit="variants/s", cores=1, kind="port",
                   sample=f"first {k} variants: oracle.feature_matrix + oracle.forest_predict (numpy) of the same ensemble, {dt:.1f} s")
    tops = 2.0 * I * L * T * N / (tot_gemm * 1e-3) / 1e12
    _line("variants/sec scored, leaf-matrix GEMM on MFMA (config C5)", N / (tot_gemm * 1e-3), "variants/s", args, tot_gemm,
          f"C5: {N} variants, on-GPU N x 20 feature matrix, XGBoost-shaped T=100 depth-6 ensemble x 3 groups as path-matrix GEMM (i8 MFMA) "
          "next to the row traversal",
          dict(bound="mfma", achieved=tops, peak=MFMA_I8_PEAK_TOPS, unit="TOP/s (int8)", frac=tops / MFMA_I8_PEAK_TOPS, traffic=None,
               kernel="forest_gemm3_kernel<false> (csrc/kernels_gemm.hip), v_mfma_i32_16x16x64_i8: ONE launch for the three variant-type "
                      "groups; HIP events around the launches", kernel_ms=tot_gemm, round4_kernel_ms_three_launches=tot_gemm2,
               round1_kernel_ms=tot_gemm_r1,
               alg_ops_per_variant=2.0 * I * L * T, traversal_ms=tot_trav, traversal_variants_per_s=N / (tot_trav * 1e-3),
               frac_of_spec_5000=tops / MFMA_I8_SPEC_TOPS,
               feature_build=dict(bound="hbm", ms=fm_ms, achieved=fm_gbps, peak=HBM_PEAK_GBPS, unit="GB/s", frac=fm_gbps / HBM_PEAK_GBPS,
                                  alg_bytes_per_variant=fm_bytes, traffic=_traffic("c5_feature_build", N)[0],
                                  kernel="fused5_kernel<3, 16, true> via ugvc_feature_matrix (resident N x F f32, no download)")),
          dict(parity=dict(gemm_equals_traversal=bool(same)), wall_s=round(wall, 2), cpu_baseline=cpu),
          dtype="f32 compares, int8 MFMA path matrix, f32 margins")
    eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=["filter", "c2", "pileup", "sec_apply", "c5_gemm"], default="filter")
    ap.add_argument("--variants", type=int, default=5_000_000, help="variants (loci for --workload pileup)")
    ap.add_argument("--snv-only", action="store_true", help="C2 shape (same as --workload c2)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--cpu-sample", type=int, default=100_000, help="variants timed on the single-process CPU baseline (0 = skip)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive measurement")
    ap.add_argument("--no-other", action="store_true", help="skip the bounded measurements of the other workloads (`other_workloads`)")
    ap.add_argument("--spinup", type=int, default=150,
                    help="untimed passes before the warm-up steps: brings the GPU to its sustained clock (reported in `spinup`; 0 = none)")
    ap.add_argument("--check-rows", type=int, default=5000,
                    help="rows at either end of the shard compared with the CPU oracle after the timed region (-1: every row)")
    ap.add_argument("--variant", type=int, default=0, help="kernel variant (debug)")
    ap.add_argument("--step-events", action="store_true",
                    help="bracket every TIMED step with its own HIP event pair (round 4's measurement; costs ~9 us per step)")
    args = ap.parse_args()
    if args.snv_only:
        args.workload = "c2"
    if args.workload in ("filter", "c2"):
        run_filter(args)
    elif args.workload == "pileup":
        run_pileup(args)
    elif args.workload == "sec_apply":
        run_sec_apply(args)
    else:
        run_c5(args)


if __name__ == "__main__":
    main()
