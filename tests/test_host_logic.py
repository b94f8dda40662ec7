"""not-gpu: host-side logic -- sharding, the partial-tally combine and its world_size-2 gloo
exchange, the 13-key aggregation against the reference's golden evaluate_with_safety dicts,
and the boundary types."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import KEYS, ROOT, load_golden, result_of


def _tally_from_episodes(ret, length, viol, crit, shut):
    import neorl_industrial_gym_amd as ni
    T = ni._lib
    p = np.zeros(T.T_ROWS)
    p[T.T_EPISODES] = len(ret); p[T.T_RET_SUM] = ret.sum(); p[T.T_RET_SQ] = (ret ** 2).sum()
    p[T.T_RET_MIN] = ret.min() if len(ret) else np.inf; p[T.T_RET_MAX] = ret.max() if len(ret) else -np.inf
    p[T.T_LEN_SUM] = length.sum(); p[T.T_LEN_SQ] = (length.astype(np.float64) ** 2).sum()
    p[T.T_VIOL] = viol.sum(); p[T.T_CRIT] = crit.sum(); p[T.T_SHUTDOWN] = shut.sum()
    p[T.T_SUCCESS] = (ret > 0).sum()
    return p


def test_shard_range_partitions_exactly():
    from neorl_industrial_gym_amd.parallel import shard_range
    for total in (1, 7, 8, 65536, 2097152, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


@pytest.mark.parametrize("key", KEYS)
def test_metrics_from_partial_matches_reference_dict(key):
    """Aggregating the golden per-episode arrays through the build's tally formulas reproduces the
    reference's evaluate_with_safety() dict (utils.py:128-152)."""
    from neorl_industrial_gym_amd.parallel import combine_partials, metrics_from_partial
    d = load_golden(key, "g4")
    want = result_of(d)
    n = len(d["ep_length"])
    cut = n // 3      # two uneven shards
    parts = []
    for sl in (slice(0, cut), slice(cut, n)):
        parts.append(_tally_from_episodes(d["ep_return"][sl], d["ep_length"][sl], d["ep_viol"][sl],
                                          d["ep_crit"][sl], d["ep_shutdown"][sl]))
    got = metrics_from_partial(combine_partials(torch.tensor(np.stack(parts))), n)
    assert set(got) == set(want) and len(got) == 13
    for k in ("safety_violations", "critical_violations", "emergency_shutdowns", "successful_episodes"):
        assert got[k] == want[k], k
    for k in ("return_mean", "return_std", "return_min", "return_max", "length_mean", "length_std",
              "safety_violations_per_episode", "constraint_satisfaction_rate", "success_rate"):
        assert got[k] == pytest.approx(want[k], rel=1e-5, abs=1e-9), k


def _gloo_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from conftest import load_golden
    from neorl_industrial_gym_amd.parallel import all_reduce_partial, metrics_from_partial, shard_range
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = load_golden("pg", "g4")
    n = len(d["ep_length"])
    s, c = shard_range(n, rank, world)
    sl = slice(s, s + c)
    part = torch.tensor(_tally_from_episodes(d["ep_return"][sl], d["ep_length"][sl], d["ep_viol"][sl],
                                             d["ep_crit"][sl], d["ep_shutdown"][sl]))
    total = all_reduce_partial(part)
    res = metrics_from_partial(total)
    np.save(os.path.join(tmp, f"r{rank}.npy"), total.numpy())
    assert res["safety_violations"] == int(d["ep_viol"].sum())
    dist.destroy_process_group()


def test_all_gather_combine_world2_gloo(tmp_path):
    """N>1 path on CPU: two ranks, contiguous shards, gloo all-gather, identical combine on both."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(_gloo_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    assert np.array_equal(a, b)                      # bit-identical on every rank
    d = load_golden("pg", "g4")
    whole = _tally_from_episodes(d["ep_return"], d["ep_length"], d["ep_viol"], d["ep_crit"], d["ep_shutdown"])
    import neorl_industrial_gym_amd as ni
    T = ni._lib
    for r in (T.T_EPISODES, T.T_LEN_SUM, T.T_LEN_SQ, T.T_VIOL, T.T_CRIT, T.T_SHUTDOWN, T.T_SUCCESS, T.T_RET_MIN, T.T_RET_MAX):
        assert a[r] == whole[r]                      # integer rows / extrema exact
    assert a[T.T_RET_SUM] == pytest.approx(whole[T.T_RET_SUM], rel=1e-14)


def test_boundary_types():
    import neorl_industrial_gym_amd as ni
    sm = ni.SafetyMetrics(constraints_satisfied=2, total_constraints=3, violation_count=1, critical_violations=1,
                          safety_score=2 / 3)
    assert sm.satisfaction_rate == 2 / 3 and sm.violation_severity == {} and sm.adaptive_threshold == 0.95
    assert ni.SafetyMetrics(0, 0, 0, 0, 1.0).satisfaction_rate == 1.0
    c = ni.SafetyConstraint(name="x", check_fn=lambda s, a: True, penalty=-1.0)
    assert c.critical is False and c.description == ""
    assert [q.value for q in ni.DatasetQuality] == ["expert", "medium", "mixed", "random"]
    from neorl_industrial_gym_amd.core import Box
    b = Box(-1.0, 1.0, (3,), np.float32)
    x = b.sample()
    assert x.dtype == np.float32 and b.contains(x) and b.low.dtype == np.float32


def test_evaluate_requires_trained_agent():
    import neorl_industrial_gym_amd as ni

    class A:
        is_trained = False
    with pytest.raises(RuntimeError, match="Agent must be trained before evaluation"):
        ni.evaluate_with_safety(A(), object(), 1)
