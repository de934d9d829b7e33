"""sec_validation: how a cohort's systematic-error (SEC) database behaves on held-out samples, on an MI355X.

Stands in for `ugbio_filtering.sec.sec_validation.run(argv)` (registered at /root/reference/ugvc/__main__.py:19,56 beside
sec_training / correct_systematic_errors; "SEC ... still undocumented", README.md:12).  Body and flags live in the absent
submodule: the FLAGS AND THE REPORT BELOW ARE BUILDER-DEFINED.  The reference's own parts are the statistic
(`multinomial_likelihood_ratio` after `scale_contingency_table`, /root/reference/ugvc/utils/stats_utils.py:12-70) and what
a hit means downstream (filter "SEC", /root/reference/ugvc/reports/report_utils.py:71-75,408-413).

Every validation VCF goes through ONE `ugvc_sec_apply` launch against the database (as correct_systematic_errors does);
the tool is the consumer of the likelihood ratios: per sample it reports how many calls sit on database loci, how many of
those the database explains at `--min_ratio`, the quantiles of the ratio, and - over a grid of thresholds - the fraction of
calls a threshold would tag, so that the threshold of correct_systematic_errors can be chosen on data.  Output:
`<prefix>.sec_validation.csv` (one row per sample and one for all), `<prefix>.sec_validation.thresholds.csv`."""
from __future__ import annotations

import argparse
import csv
import logging
import sys

import numpy as np

logger = logging.getLogger("ugvc")

THRESHOLDS = (1e-6, 1e-4, 1e-3, 0.01, 0.02, 0.05, 0.1, 0.2, 0.5, 1.0)
QUANTILES = (0.01, 0.05, 0.25, 0.5, 0.75, 0.95, 0.99)


def get_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="sec_validation.py", description=run.__doc__)
    ap.add_argument("--inputs", help="VCF of one held-out sample (repeatable)", type=str, action="append", default=[])
    ap.add_argument("--input_list", help="text file with one VCF path per line", type=str)
    ap.add_argument("--sec_db", help="SEC database written by sec_training (.npz)", type=str, required=True)
    ap.add_argument("--reference_file", help="Indexed reference FASTA file (contig names and order)", type=str, required=True)
    ap.add_argument("--output_prefix", help="Prefix of the two report files", type=str, required=True)
    ap.add_argument("--min_ratio", help="the threshold of correct_systematic_errors to report at", type=float, default=0.05)
    ap.add_argument("--no_scaling", help="compare with the cohort's raw counts instead of scaling them to the call's depth", action="store_true")
    ap.add_argument("--device", help="GPU index (MI355X)", type=int, default=0)
    return ap


def summarise(name: str, ratio: np.ndarray, hit: np.ndarray) -> dict:
    """One report row from a sample's likelihood ratios (NaN off the database) and verdicts."""
    on = ~np.isnan(ratio)
    r = ratio[on]
    row = dict(sample=name, n_calls=int(ratio.size), n_on_database=int(on.sum()), n_sec=int(hit.sum()),
               frac_on_database=float(on.mean()) if ratio.size else 0.0, frac_sec_of_calls=float(hit.mean()) if ratio.size else 0.0,
               frac_sec_of_database_calls=float(hit.sum() / on.sum()) if on.sum() else 0.0)
    for q in QUANTILES:
        row[f"ratio_q{int(q * 100):02d}"] = float(np.quantile(r, q)) if r.size else float("nan")
    return row


def run(argv: list[str]):
    """Validate a SEC database on held-out samples: how many calls it covers and explains, at which thresholds"""
    args = get_parser().parse_args(argv[1:])
    from ..engine import Engine            # fails loudly if the library or the GPU is missing
    from ..io import vcf_native
    from .sec_training import load_db
    paths = list(args.inputs)
    if args.input_list:
        with open(args.input_list) as fh:
            paths += [ln.strip() for ln in fh if ln.strip()]
    if not paths:
        raise ValueError("sec_validation: no VCFs given (--inputs / --input_list)")
    names = vcf_native.read_fasta_names(args.reference_file)
    db_keys, expected = load_db(args.sec_db, names)
    rows, all_ratio, all_hit = [], [], []
    with Engine(args.device) as eng:
        eng.set_contigs(names)                         # the join is on (contig, pos): no bases needed
        eng.set_sec_db(db_keys, expected)
        for p in paths:
            vt = vcf_native.read_vcf(p, names).table
            eng.upload_variants(vt)
            ratio, hit = eng.sec_apply(args.min_ratio, not args.no_scaling, mark=False)
            rows.append(summarise(p, ratio, hit))
            all_ratio.append(ratio)
            all_hit.append(hit)
            logger.info("%s: %d calls, %d on the database, %d explained at %g", p, vt.n, rows[-1]["n_on_database"], rows[-1]["n_sec"], args.min_ratio)
    ratio, hit = np.concatenate(all_ratio), np.concatenate(all_hit)
    rows.append(summarise("ALL", ratio, hit))
    with open(args.output_prefix + ".sec_validation.csv", "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    on = ~np.isnan(ratio)
    with open(args.output_prefix + ".sec_validation.thresholds.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["min_ratio", "n_sec", "frac_of_calls", "frac_of_database_calls"])
        for t in THRESHOLDS:
            k = int((ratio[on] >= t).sum())                        # is_sec = ratio >= min_ratio (kernels_sec.hip)
            w.writerow([t, k, k / max(ratio.size, 1), k / max(int(on.sum()), 1)])
    return 0


if __name__ == "__main__":
    run(sys.argv)
