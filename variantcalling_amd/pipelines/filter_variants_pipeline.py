"""filter_variants_pipeline: apply a trained filtering model to a raw callset VCF on an MI355X.

Drop-in for `ugbio_filtering.filter_variants_pipeline.run(argv)` (registered at
/root/reference/ugvc/__main__.py:18,47; script setup.py:42), flags exactly as documented in
docs/filter_variants_pipeline.md:9-46.  VCF -> SoA columns -> ONE call into libugvc_mi355x.so
(featurize, interval / blacklist lookup, per-variant-type model score, FILTER) -> VCF with
PASS|LOW_SCORE, TREE_SCORE, HPOL_RUN, COHORT_FP (docs/howto-callset-filter.md:61-65)."""
from __future__ import annotations

import argparse
import logging
import os
import sys

from .. import model_io
from ..io import multiallelic
from ..io import vcf_native as vcfio      # native codec (libugvc_vcf.so); io.vcf is its pure-Python reference
from . import common

logger = logging.getLogger("ugvc")


def get_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="filter_variants_pipeline.py", description="Filter VCF")
    ap.add_argument("--input_file", help="Name of the input VCF file", type=str, required=True)
    ap.add_argument("--model_file", help="Pickle model file", type=str, required=True)
    ap.add_argument("--model_name", help="Model file", type=str, required=True)
    ap.add_argument("--hpol_filter_length_dist", nargs=2, type=int, help="Length and distance to the hpol run to mark",
                    default=[10, 10])
    ap.add_argument("--runs_file", help="Homopolymer runs file", type=str, required=True)
    ap.add_argument("--blacklist", help="Blacklist file", type=str, required=False)
    ap.add_argument("--blacklist_cg_insertions", help="Should CCG/GGC insertions be filtered out?", action="store_true")
    ap.add_argument("--reference_file", help="Indexed reference FASTA file", type=str, required=True)
    ap.add_argument("--output_file", help="Output VCF file", type=str, required=True)
    ap.add_argument("--is_mutect", help="Is the input a result of mutect", action="store_true")
    ap.add_argument("--flow_order", help="Sequencing flow order (4 cycle)", type=str, default="TGCA")
    ap.add_argument("--annotate_intervals", help="interval files for annotation (multiple possible)", type=str,
                    action="append", default=[])
    ap.add_argument("--device", help="GPU index (MI355X); under a multi-process launcher every rank takes its LOCAL_RANK", type=int, default=0)
    return ap


def run(argv: list[str]):
    """Filter VCF"""
    args = get_parser().parse_args(argv[1:])
    from ..engine import Engine, configure     # fails loudly if the library or the GPU is missing

    import time
    t0 = time.perf_counter()
    stages = {}

    def lap(name):
        nonlocal t0
        t1 = time.perf_counter()
        stages[name] = t1 - t0
        t0 = t1

    # One process per GPU when launched under `python -m torch.distributed.run` / with RANK, WORLD_SIZE, MASTER_* in the
    # environment (the reference parallelises the neighbouring tools per contig, docs/run_comparison_pipeline.md:81;
    # here: equal-count slices of the sorted callset).  Every rank parses the inputs with its share of
    # the host threads (rank 0 needs every record's text anyway to write the output), scores its slice against the part
    # of the genome and of the side tables that slice can touch, and one RCCL all-gather of (score, filter, flags) puts
    # the whole verdict on every rank; rank 0 writes.  No torch anywhere: the rendezvous is dist.Group (plain TCP).
    from .. import dist, shard
    grp = dist.Group()
    device = args.device if grp.world == 1 else grp.local_rank
    n_threads = 0 if grp.world == 1 else max(1, (os.cpu_count() or 1) // max(grp.local_world, 1))
    # the GPU context (HIP runtime start-up: 0.1-0.3 s) comes up on a thread of its own while the inputs are read; reference,
    # side tables and the callset VCF are read concurrently (common.load_side_tables)
    from concurrent.futures import ThreadPoolExecutor
    ctx_pool = ThreadPoolExecutor(max_workers=1)
    f_eng = ctx_pool.submit(Engine, device)
    logger.info("reading side tables and %s", args.input_file)
    # Rank 0 reads the whole callset (it writes the output: it needs every record's text); every OTHER rank tokenises only its
    # equal-count slice of the records (vcf_native.read_vcf(part=...): the file is still inflated and cut into lines, the
    # tokeniser / ordering / column stages - two thirds of the reader's time - run on 1 / world of them).  That slice is the
    # rank's shard of the SORTED callset only if the file is sorted: established below across ranks, else every rank reads all.
    part = (grp.rank, grp.world) if grp.world > 1 and grp.rank > 0 else None
    # Single process: as soon as the reader has counted the record lines (a third of the way through the read) the context thread
    # prepares what the boundary call allocates once per callset size - resident columns, pinned staging, worker pool - and
    # loads the kernels (Engine.reserve: 44 of the first 5 M-row call's 49 ms); rows and allele bytes with an eighth of margin
    # for multi-allelic records (a short reservation is topped up by the call itself).
    reserve = {}

    def on_count(n_records, _text_bytes):
        if grp.world == 1:
            reserve["f"] = ctx_pool.submit(lambda: f_eng.result().reserve(n_records + n_records // 8, 4 * n_records))

    def wait_for_reserve():
        f = reserve.pop("f", None)
        return f.exception() if f is not None else None      # (waits: the context must not be used for a pass, or closed, under it)

    # Single process (round 6): a table goes to the GPU the moment its reader has it - on the context's thread, in the order the
    # engine needs (the reference first: the interval tables are checked against its contigs) - so the uploads run UNDER the
    # slowest reader (the callset VCF) instead of behind all of them: the "context + uploads" stage was 0.10-0.12 s of a 1.2 s run.
    # With several ranks every rank uploads the slice of the tables its shard can touch, known only when everything is read.
    import threading
    hp_len, hp_dist = args.hpol_filter_length_dist
    early = {"lock": threading.Lock(), "ref_sent": False, "waiting": [], "futs": [], "done": set()}

    def _upload(kind, index, table):
        eng = f_eng.result()
        if kind == "reference":
            eng.set_reference(table)
        elif kind == "runs":
            eng.set_runs(table, hp_len, hp_dist, True)
        elif kind == "track":
            eng.set_track(index, table)
        else:
            eng.set_blacklist(table)
        early["done"].add((kind, index))

    def on_ready(kind, index, table):
        if grp.world != 1:
            return
        with early["lock"]:
            if kind == "reference":
                early["futs"].append(ctx_pool.submit(_upload, kind, index, table))
                early["ref_sent"] = True
                for w in early["waiting"]:
                    early["futs"].append(ctx_pool.submit(_upload, *w))
                early["waiting"].clear()
            elif early["ref_sent"]:
                early["futs"].append(ctx_pool.submit(_upload, kind, index, table))
            else:
                early["waiting"].append((kind, index, table))

    try:
        ref, runs, tracks, bl, extra = common.load_side_tables(
            args.reference_file, args.runs_file, args.annotate_intervals, args.blacklist,
            also={"vcf": lambda names: vcfio.read_vcf(args.input_file, names, is_mutect=args.is_mutect, n_threads=n_threads, part=part,
                                                      on_count=on_count)}, on_ready=on_ready)
        vcf = extra["vcf"]
        if grp.world > 1:
            import json
            t = vcf.table
            mine_ok = bool(getattr(vcf, "sorted_in_file", True))
            first = [int(t.contig[0]), int(t.pos[0])] if t.n else None
            last = [int(t.contig[-1]), int(t.pos[-1])] if t.n else None
            infos = [json.loads(b.decode()) for b in grp.allgather_bytes(json.dumps(
                dict(ok=mine_ok, n_total=int(getattr(vcf, "n_total", t.n)), first=first, last=last)).encode())]
            seams = all(infos[r]["last"] is None or infos[r + 1]["first"] is None or infos[r]["last"] <= infos[r + 1]["first"]
                        for r in range(1, grp.world - 1))
            parts_ok = all(i["ok"] for i in infos) and len({i["n_total"] for i in infos}) == 1 and seams
            if not parts_ok and part is not None:           # an unsorted file: the shards are slices of the SORTED callset - read all
                vcf.close()
                vcf = vcfio.read_vcf(args.input_file, ref.names, is_mutect=args.is_mutect, n_threads=n_threads)
            part = part if parts_ok else None
        # an estimator fitted on a named frame finds its interval columns by BED stem (`LCR-hs38`, `exome.twist`, ...)
        forests = model_io.load_model_file(args.model_file, args.model_name, track_names=[t.name for t in tracks])
        common.check_model_tracks(forests, len(tracks), "filter_variants_pipeline")
    except BaseException:
        try:
            for f in early["futs"]:                          # (uploads queued on the context's thread: not under a closing context)
                f.exception()
            wait_for_reserve()
            f_eng.result().close()
        except Exception:                                   # (no GPU / no library: the input error is the one to report)
            pass
        ctx_pool.shutdown()
        raise
    lap("reference + side tables + model + VCF -> columns (concurrent, native codec; uploads under the readers)")
    # one row per ALT allele (multi-allelic records, spanning deletions: io/multiallelic.py), one verdict per record
    table, base_row = multiallelic.expand(vcf)
    if grp.world > 1:
        # shards = equal-count slices of the RECORDS (a record's allele rows stay together): rows [rb[r], rb[r + 1]) of the
        # expanded table on a rank that holds all of it, the whole expanded part on a rank that read its slice only
        import json
        import numpy as np
        n_rec = int(getattr(vcf, "n_total", vcf.table.n))
        b = shard.shard_bounds(n_rec, grp.world)
        if part is None:
            rb = np.searchsorted(base_row, b, side="left")
            mine = table.slice(int(rb[grp.rank]), int(rb[grp.rank + 1]))
        else:
            if int(vcf.part_lo) != int(b[grp.rank]) or vcf.table.n != int(b[grp.rank + 1] - b[grp.rank]):
                raise RuntimeError("internal: the codec's part bounds differ from shard.shard_bounds")
            mine = table
        counts = [int(x.decode()) for x in grp.allgather_bytes(str(mine.n).encode())]
    # (the context is CLOSED on a thread of its own once the verdict is on the host: freeing 3+ GB of device buffers is tens of
    # milliseconds the write-back need not wait for - round 6; joined before the tool returns)
    class _CloseBehind:
        def __init__(self, eng):
            self.eng, self.t = eng, None

        def __enter__(self):
            return self.eng

        def __exit__(self, *exc):
            self.t = threading.Thread(target=self.eng.close, daemon=False)
            self.t.start()
            return False

        def join(self):
            if self.t is not None:
                self.t.join()
    closing = _CloseBehind(f_eng.result())
    with closing as eng:
        if grp.world == 1:
            try:
                for f in early["futs"]:                          # the uploads that ran under the readers: wait, surface their errors
                    f.result()
                # whatever has not gone up yet (no file given: an empty table; the model; the flow order)
                if ("reference", 0) not in early["done"]:
                    eng.set_reference(ref)
                if ("runs", 0) not in early["done"]:
                    import numpy as np
                    from .. import schema as S
                    eng.set_runs(runs if runs is not None else S.IntervalTrack(np.zeros(0, np.int32), np.zeros(0, np.int32),
                                                                                np.zeros(ref.n_contigs + 1, np.int32), "runs"), hp_len, hp_dist, True)
                for k, t in enumerate(tracks):
                    if ("track", k) not in early["done"]:
                        eng.set_track(k, t)
                eng.set_n_tracks(len(tracks))
                if ("blacklist", 0) not in early["done"]:
                    eng.set_blacklist(bl)
                eng.set_flow_order(args.flow_order)
                eng.set_models(forests)
            finally:
                res_err = wait_for_reserve()
                ctx_pool.shutdown()
            if res_err is not None:
                raise res_err
            lap("context + uploads (what the readers' threads had not sent yet: model, flow order)")
            res_rows = eng.filter_variants(table)
            lap("upload variants + scoring pass + download")
        else:
            ctx_pool.shutdown()
            ref_r, runs_r, tracks_r, bl_r, mine_r = shard.slice_context(ref, runs, tracks, bl, mine, hpol_dist=hp_dist)
            configure(eng, ref_r, runs_r, tracks_r, bl_r, forests, args.flow_order, hp_len, hp_dist, True)
            uid = grp.broadcast_bytes(eng.comm_unique_id() if grp.rank == 0 else None, 0)
            eng.comm_init(uid, grp.rank, grp.world)
            info = eng.comm_info()
            if info["nranks"] != grp.world or info["rank"] != grp.rank:
                raise RuntimeError(f"RCCL communicator reports {info}, launcher says rank {grp.rank} of {grp.world}")
            lap("context + uploads (reference slice, table slices, model) + RCCL communicator")
            cap = -(-max(max(counts), 1) // 256) * 256          # (every rank's slot of the gather buffers 256-byte aligned: shard.shard_cap)
            eng.upload_variants(mine_r)
            eng.filter_resident()
            eng.allgather_resident(cap)
            res_rows = eng.gathered_download(cap, grp.world, counts)
            lap(f"upload shard + scoring pass + RCCL all-gather x{info['nranks']} + download")
        # (only a rank that holds the whole callset folds the allele rows back: rank 0 - the others are done)
        res = multiallelic.collapse(res_rows, base_row, vcf.table.n) if (grp.world == 1 or part is None) else None
    grp.barrier()
    if grp.rank != 0:
        closing.join()
        grp.close()
        return 0
    cg = common.cg_insertion_mask(vcf.table) if args.blacklist_cg_insertions else None
    logger.info("writing %s", args.output_file)
    try:
        vcfio.write_filtered_vcf(args.output_file, vcf, res, cg)
    finally:
        closing.join()
    lap("FILTER/INFO write-back (native codec)")
    run.last_stage_seconds = stages                        # read by tools/bench_pipeline.py
    logger.info("stage seconds: %s", ", ".join(f"{k} {v:.3f}" for k, v in stages.items()))
    n_pass = int(((res.filter == 0) & (res.flags & 3 == 0)).sum())
    logger.info("%d variants, %d PASS", vcf.table.n, n_pass)
    grp.close()
    return 0


if __name__ == "__main__":
    run(sys.argv)
