"""CPU ORACLE (SEC statistic, quality transforms) - TEST INFRASTRUCTURE ONLY.

Restates /root/reference/ugvc/utils/stats_utils.py:12-70 (contingency scaling, add-one
corrected multinomial likelihood and likelihood ratio) and ugvc/utils/math_utils.py:31-101
(phred / unphred).  PINNED: tests/test_oracle_golden.py replays every known-answer value of
the reference's own tests (test/unit/utils/test_stats_utils.py:18-110,
test/unit/utils/test_math_utils.py:10-23) against these functions.
"""
from __future__ import annotations

import numpy as np
from scipy.special import gammaln, xlogy


def scale_contingency_table(table, n):
    """stats_utils.py:12-28: rescale counts so they sum to ~n (round half to even, as numpy)."""
    s = sum(table)
    if s > 0:
        return list(np.round(np.array(table) * (n / s)).astype(int))
    return table


def correct_multinomial_frequencies(counts):
    """stats_utils.py:31-44: add-one corrected category frequencies."""
    c = np.array(counts) + 1
    return c / np.sum(c)


def multinomial_likelihood(actual, expected):
    """stats_utils.py:47-63: multinomial.pmf(x=actual, n=sum(actual), p=corrected(expected)),
    written out as scipy evaluates it: exp(gammaln(n+1) + sum(xlogy(x, p) - gammaln(x+1)))."""
    p = correct_multinomial_frequencies(expected)
    x = np.asarray(actual, dtype=np.float64)
    n = x.sum()
    return float(np.exp(gammaln(n + 1) + np.sum(xlogy(x, p) - gammaln(x + 1))))


def multinomial_likelihood_ratio(actual, expected):
    """stats_utils.py:66-70."""
    lik = multinomial_likelihood(actual, expected)
    mx = multinomial_likelihood(actual, actual)
    return lik, lik / mx


def sec_batch(actual: np.ndarray, expected: np.ndarray):
    """Vectorised [n_loci, k] form used to check the GPU kernel."""
    a = np.asarray(actual, dtype=np.float64)
    e = np.asarray(expected, dtype=np.float64)

    def logpmf(x, cnt):
        p = (cnt + 1) / (cnt + 1).sum(axis=1, keepdims=True)
        return gammaln(x.sum(axis=1) + 1) + np.sum(xlogy(x, p) - gammaln(x + 1), axis=1)

    lik = np.exp(logpmf(a, e))
    return lik, lik / np.exp(logpmf(a, a))


def sec_db_build(keys: np.ndarray, counts: np.ndarray):
    """SEC database from cohort observations (BUILDER-DEFINED, include/ugvc_mi355x.h "SEC database"): per locus key the
    summed counts.  Checker of ugvc_sec_db_build."""
    keys = np.asarray(keys, dtype=np.uint64)
    counts = np.asarray(counts, dtype=np.int64)
    uk, inv = np.unique(keys, return_inverse=True)
    out = np.zeros((uk.size, counts.shape[1]), np.int64)
    np.add.at(out, inv, counts)
    return uk, out


def sec_apply(keys, dp, ad_ref, ad_alt, db_keys, db_expected, min_ratio=0.05, scale_expected=True):
    """Checker of ugvc_sec_apply: per variant on a database locus, observed = (ad_ref, ad_alt[, max(dp - ad_ref - ad_alt,
    0)]), expected optionally passed through scale_contingency_table(expected, sum(observed)) (stats_utils.py:12-28),
    ratio = multinomial_likelihood_ratio(observed, expected)[1] (:66-70) - each call the scalar restatement above."""
    keys = np.asarray(keys, dtype=np.uint64)
    db_keys = np.asarray(db_keys, dtype=np.uint64)
    k = np.asarray(db_expected).shape[1]
    ratio = np.full(keys.size, np.nan)
    hit = np.zeros(keys.size, bool)
    j = np.searchsorted(db_keys, keys)
    for i in range(keys.size):
        if j[i] >= db_keys.size or db_keys[j[i]] != keys[i]:
            continue
        r, a = max(int(ad_ref[i]), 0), max(int(ad_alt[i]), 0)
        obs = [r, a] + ([max(int(dp[i]) - r - a, 0)] if k > 2 else []) + [0] * max(0, k - 3)
        exp = [int(x) for x in db_expected[j[i]]]
        if scale_expected:
            exp = [int(x) for x in scale_contingency_table(exp, sum(obs))]
        ratio[i] = multinomial_likelihood_ratio(obs, exp)[1]
        hit[i] = ratio[i] >= min_ratio
    return ratio, hit


def phred(p):
    """math_utils.py:31-47."""
    return -10 * np.log10(np.array(p, dtype=float))


def phred_str(p):
    """math_utils.py:50-65."""
    return "".join(chr(int(x) + 33) for x in phred(p))


def unphred(q):
    """math_utils.py:67-84."""
    if isinstance(q, float):
        return 10 ** (-q / 10)
    return np.power(10, -np.array(q, dtype=float) / 10)


def unphred_str(s):
    """math_utils.py:86-101."""
    return unphred([ord(x) - 33 for x in s])
