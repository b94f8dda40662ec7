"""not-gpu: the fast-mode workload's STATISTICS against the reference.

The metric is "env-steps/sec + safety-violation-count parity".  Counts are exact on injected draws (G1-G4); in the mode
the bench times, the draws come from this build's Philox + probit generator, so what can be pinned there is the
DISTRIBUTION: tests/golden/reference_stats.npz holds the outcome of thousands of episodes of the reference's own
measurement loop (performance_benchmark.py:106-133: uniform float32 actions, reset on done) run by the reference's code
with the reference's np.random draws (chemical_reactor.py:93-103,149,159, power_grid.py:98-108,136-144,
robot_assembly.py:118-122).  Here the CPU oracle plays the same loop with the build's generator (the device path equals
it bit for bit: tests/test_gpu_parity.py) -- ONE episode per lane, lanes frozen when done, so the sample is unbiased --
and every statistic has to lie within 4 standard errors.  tests/test_gpu_reference_stats.py does the same on the device
in auto-reset mode at the BASELINE batch sizes with >= 1e6 episodes."""
import numpy as np
import pytest

import refstats
from conftest import KEYS

MAX_STEPS = {"cr": 500, "pg": 1000, "ra": 1000}
LANES = {"cr": 16384, "pg": 65536, "ra": 65536}


def test_fixture_is_the_documented_sample():
    """Shape and the headline numbers BASELINE.md section 2 quotes for the reference (CR ~40 violations and 0 critical per
    episode, mean length ~350; PG ~1.0 / ~1.0, length ~5; RA median length 1, ~0.99 critical per episode)."""
    for key, n in (("cr", 24000), ("pg", 160000), ("ra", 160000)):
        r = refstats.load_reference(key)
        assert r["length"].shape == (n,) and r["cbits"].shape == (n, 3) and r["cause"].dtype == np.uint8
        assert (r["length"] >= 1).all() and (r["length"] <= MAX_STEPS[key]).all()
        assert (r["cbits"].sum(axis=1) == r["viol"]).all()                 # violation_count = sum over the constraints, step by step
        assert (((r["cause"] & 2) != 0) == (r["length"] == MAX_STEPS[key])).all()      # truncated <=> the episode ran to max_episode_steps
    cr, pg, ra = (refstats.load_reference(k) for k in KEYS)
    assert 330 < cr["length"].mean() < 380 and 38 < cr["viol"].mean() < 46 and cr["crit"].sum() == 0
    assert 4.5 < pg["length"].mean() < 6 and 0.97 < pg["viol"].mean() < 1.02 and 0.97 < pg["crit"].mean() < 1.02
    assert np.median(ra["length"]) == 1 and 0.98 < ra["crit"].mean() < 1.0


@pytest.mark.parametrize("key", KEYS)
def test_oracle_with_the_fast_mode_generator_reproduces_the_reference_distribution(oracle, key):
    O = oracle
    n = LANES[key]
    uniform = O.make_policy(kind=1, p_uniform=1.0, uniform_range=1.0)           # action_space.sample(): uniform in [-1, 1]^A every step
    res = O.rollout_policy(key, n, MAX_STEPS[key], uniform, seed=0xC0FFEE, autoreset=False, flavor=O.MATH_LIBM)
    t = res["tallies"]
    assert all(t[i].episodes == 1 for i in range(0, n, 97)) and (res["done"] != 0).all()      # one finished episode per lane
    sample = {"length": np.array([t[i].steps for i in range(n)]), "viol": np.array([t[i].violations for i in range(n)]),
              "crit": np.array([t[i].critical for i in range(n)]), "ret": np.array([t[i].reward_sum for i in range(n)]),
              "cause": np.array([(1 if t[i].terminated else 0) | (2 if t[i].truncated else 0) for i in range(n)], dtype=np.uint8)}
    table, th = refstats.reference_table(key, MAX_STEPS[key])
    got = {k: float(v.mean()) for k, v in refstats.statistics(sample, th).items() if k != "p_critical_shutdown"}
    rows, bad = refstats.compare(table, got, n)
    print(refstats.format_rows(rows))
    assert not bad, "statistics off the reference's by more than %g standard errors:\n%s" % (refstats.NSIGMA, refstats.format_rows(bad))
    assert len(rows) >= 10
