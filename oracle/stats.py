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
