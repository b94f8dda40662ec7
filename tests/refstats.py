"""Shared by tests/test_reference_stats.py (CPU) and tests/test_gpu_reference_stats.py (GPU): the statistics of the
workload the driver times -- performance_benchmark.py:106-133's loop under uniform float32 actions -- as the REFERENCE
produces them (tests/golden/reference_stats.npz, written by `oracle/gen_golden.py stats`: thousands of episodes per env
run by the reference's own code with its own np.random draws), and the comparison of a build-side sample against them.

A sample is a dict of per-episode arrays (any subset of): length, ret, viol, crit, cbits [n, 3], cause (bit 0
terminated, bit 1 truncated, bit 2 critical shutdown) -- or, where only sums are at hand (the device path reduces on
the GPU), a dict of the same statistics' VALUES plus `n`.  Every statistic is a mean over episodes, so its sampling
error is std / sqrt(n); a build-side value passes when it lies within NSIGMA combined standard errors of the
reference's (the build-side error is taken from the reference's spread scaled to the build's episode count)."""
import os

import numpy as np

from conftest import GOLDEN

NSIGMA = 4.0
QUANTILES = (0.05, 0.10, 0.25, 0.50, 0.75, 0.90, 0.95)


def load_reference(key):
    d = np.load(os.path.join(GOLDEN, "reference_stats.npz"))
    return {k[len(key) + 1:]: d[k] for k in d.files if k.startswith(key + "_")}


def length_thresholds(ref_length, max_steps):
    """Episode lengths at which the length CDF is compared: the reference's own quantiles (heavy tails -- RobotAssembly's
    median episode is ONE step, its mean 44 -- are compared by distribution, not only by mean), 1, and max_steps - 1."""
    t = sorted({int(np.quantile(ref_length, q, method="lower")) for q in QUANTILES} | {1, 2, max_steps - 1})
    return [x for x in t if 1 <= x < max_steps]


def statistics(sample, thresholds):
    """name -> per-episode value array (its mean is the statistic)."""
    L = np.asarray(sample["length"], dtype=np.float64)
    out = {"length_mean": L}
    if "viol" in sample:
        out["violations_per_episode"] = np.asarray(sample["viol"], dtype=np.float64)
    if "crit" in sample:
        out["critical_per_episode"] = np.asarray(sample["crit"], dtype=np.float64)
    if "ret" in sample:
        out["return_mean"] = np.asarray(sample["ret"], dtype=np.float64)
    if "cbits" in sample:
        for k in range(3):
            out[f"constraint{k}_violated_steps_per_episode"] = np.asarray(sample["cbits"][:, k], dtype=np.float64)
    if "cause" in sample:
        c = np.asarray(sample["cause"])
        out["p_terminated"] = ((c & 1) != 0).astype(np.float64)
        out["p_truncated"] = ((c & 2) != 0).astype(np.float64)
        out["p_critical_shutdown"] = ((c & 4) != 0).astype(np.float64)
    for t in thresholds:
        out[f"p_length_le_{t}"] = (L <= t).astype(np.float64)
    return out


def reference_table(key, max_steps):
    """name -> (mean, std, n) of the reference's sample, and the length thresholds used."""
    ref = load_reference(key)
    th = length_thresholds(ref["length"], max_steps)
    st = statistics(ref, th)
    return {k: (float(v.mean()), float(v.std()), int(v.size)) for k, v in st.items()}, th


def compare(ref_table, got, n_got, nsigma=NSIGMA):
    """got: name -> value (means over n_got episodes).  Returns (rows, failures); a row is
    (name, reference, build, combined standard error, deviation in standard errors)."""
    rows, bad = [], []
    for name, val in sorted(got.items()):
        if name not in ref_table:
            continue
        m, s, n = ref_table[name]
        if s == 0.0:                              # an event the reference never (or always) saw in n episodes: the rule of three
            s = np.sqrt(3.0 / n)
        se = float(np.sqrt(s * s / n + s * s / max(n_got, 1)))
        dev = (float(val) - m) / se
        rows.append((name, m, float(val), se, dev))
        if not abs(dev) <= nsigma:
            bad.append(rows[-1])
    return rows, bad


def format_rows(rows):
    return "\n".join("%-44s reference %12.5f  build %12.5f  se %10.5f  %+6.2f sigma" % r for r in rows)
