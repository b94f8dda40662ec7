"""-m gpu: the statistics of the FAST MODE the driver times -- fused rollout, in-kernel generator, auto-reset -- against
the reference's: "safety-violation-count parity" in the mode where no recorded draw exists.

tests/golden/reference_stats.npz: 24 000 / 160 000 / 160 000 episodes of performance_benchmark.py:106-133's loop (uniform
float32 actions, reset on done) run by the REFERENCE with its own np.random draws (chemical_reactor.py:93-103,149,159,
power_grid.py:98-108,136-144, robot_assembly.py:118-122).  Here nig_rollout -- the kernel form bench.py times for the
batch: three-wave ChemicalReactor at 65 536 lanes, wide-512 PowerGrid at 262 144, three-wave RobotAssembly at 65 536 --
runs launch after launch with fresh uniform actions until every lane has finished K episodes (>= 1e6 episodes per env),
and each statistic of the FIRST K episodes of every lane (a fixed count per lane: no length bias, unlike "episodes
finished inside a window") must lie within 4 combined standard errors of the reference's: mean episode length, the
length CDF at the reference's quantiles (RobotAssembly's heavy tail: median 1, mean 44), violations / critical violations
per episode, per-constraint violated steps per episode, termination / truncation / critical-shutdown split, mean return.
All reductions run on the device from the per-step flag words and rewards the kernel writes."""
import numpy as np
import pytest
import torch

import refstats
from conftest import ENV_NAME

pytestmark = pytest.mark.gpu

MAX_STEPS = {"cr": 500, "pg": 1000, "ra": 1000}
CASES = {"cr": (65536, 16, "split_rollout_kernel<ChemicalReactor,1,4>"),
         "pg": (262144, 4, "rollout_wide_kernel<PowerGrid,1,512>"),
         "ra": (65536, 16, "split_rollout_kernel<RobotAssembly,1,4>")}
P = 250


@pytest.fixture(scope="module")
def ni():
    import neorl_industrial_gym_amd as ni
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    ni.tune(split_blocks=-1, wide_min_blocks=-1)
    yield ni


@pytest.mark.parametrize("key", ["cr", "pg", "ra"])
def test_fast_mode_statistics_match_the_reference(ni, key):
    import bench
    import types
    B, K, kernel = CASES[key]
    wl = types.SimpleNamespace(ni=ni, key=key, B=B, outputs="min")
    assert bench.rollout_kernel_name(wl) == kernel                     # the form the driver's line times for this env and batch
    s = ni.uniform_action_statistics(ENV_NAME[key], B, K)          # fused launches + device-side reductions (utils.py)
    n = s["episodes"]
    assert n == B * K >= 1_000_000 and int(s["hist"].sum()) == n and int((s["hist"] * np.arange(s["hist"].size)).sum()) == s["steps"]
    table, th = refstats.reference_table(key, MAX_STEPS[key])
    cdf = np.cumsum(s["hist"]) / n
    got = {"length_mean": s["steps"] / n, "violations_per_episode": s["viol"] / n, "critical_per_episode": s["crit"] / n,
           "return_mean": s["ret"] / n, "p_terminated": s["term"] / n, "p_truncated": s["trunc"] / n,
           "p_critical_shutdown": s["shut"] / n}
    for k in range(3):
        got[f"constraint{k}_violated_steps_per_episode"] = s[f"c{k}"] / n
    for t in th:
        got[f"p_length_le_{t}"] = float(cdf[t])
    rows, bad = refstats.compare(table, got, n)
    print(f"\n{ENV_NAME[key]}: {n} episodes ({B} lanes x first {K}), {s['launches']} launches of {P} steps")
    print(refstats.format_rows(rows))
    assert len(rows) == len(got) >= 14
    assert not bad, "fast-mode statistics off the reference's by more than %g standard errors:\n%s" % (refstats.NSIGMA, refstats.format_rows(bad))
