"""-m gpu: round-3 additions pinned on the device: bench.py's own N-rank launch with the real workload, ..."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ENV_NAME, KEYS, ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ni():
    import neorl_industrial_gym_amd as ni
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return ni


def test_bench_two_ranks_on_one_gpu():
    """`bench.py --gpus 2` WITHOUT torchrun on the one-GPU box: it starts its own two ranks (both on cuda:0, gloo for
    the exchange: NIG_BENCH_REHEARSE=1 -- a 1-GPU box cannot host two RCCL ranks), runs the real device workload on
    each, and the line reports n_gpus == 2, two ranks with episodes in the tally exchange, lanes keyed by rank."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["NIG_BENCH_REHEARSE"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "8192", "--steps", "2",
                        "--warmup", "1", "--plan-steps", "250", "--settle", "0", "--no-step-api", "--no-cpu-baseline",
                        "--no-parity", "--no-mixed", "--no-brackets"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    rec = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["ranks"] == 2 and len(rec["episodes_per_rank"]) == 2
    assert all(e > 0 for e in rec["episodes_per_rank"]) and rec["tally"]["episodes"] == sum(rec["episodes_per_rank"])
    assert rec["config"]["global_batch"] == 2 * 8192 and rec["value"] > 0
    pg = rec["powergrid"]
    assert pg["tally_check"]["ranks"] == 2 and pg["tally"]["episodes"] == sum(pg["tally_check"]["episodes_per_rank"])


# ---------------------------------------------------------------------------------------------------------------
# PowerGrid's LDS-resident rollout (csrc/nig_pg_lds.hpp, rollout_wide_kernel<PowerGrid, OUT, 512>) against the
# register-resident rollout_kernel on the same inputs, and against the oracle.
# ---------------------------------------------------------------------------------------------------------------
PG = "PowerGrid-v0"
NEVER = 1 << 30


def _pg_ring(env, R, t0=70):
    ring = torch.empty(R, env.action_dim, env.ld, dtype=torch.float32, device=env.device)
    for s in range(R):
        env.fill_actions(t0 + s, ring[s])
    return ring


def _pg_run(ni, wide, B, chunks, outputs, R, max_steps=1000, seed=11, tally=True, env_index0=0, cmask=None, t0=70):
    ni.tune(wide_min_blocks=1 if wide else NEVER)
    env = ni.make_batched(PG, B, seed=seed, autoreset=True, tally=tally, max_episode_steps=max_steps, env_index0=env_index0)
    if cmask is not None:
        env.set_constraint_mask(cmask)
    ring = _pg_ring(env, R, t0)
    env.reset()
    got = []
    for T in chunks:
        rew = fl = obs = None
        if outputs != "none":
            rows = () if outputs == "last" else (T,)
            rew = torch.full(rows + (env.ld,), float("nan"), dtype=torch.float32, device=env.device)
            fl = torch.zeros(rows + (env.ld,), dtype=torch.int32, device=env.device)
        if outputs == "aos":
            obs = torch.full((T, B, env.state_dim), float("nan"), dtype=torch.float32, device=env.device)
        elif outputs == "soa":
            obs = torch.full((T, env.state_dim, env.ld), float("nan"), dtype=torch.float32, device=env.device)
        env.rollout(T, ring, rew, fl, obs)
        torch.cuda.synchronize()
        for t in (rew, fl, obs):
            if t is not None:
                got.append(t[..., :B].cpu() if t is not obs or outputs == "soa" else t.cpu())
    got += [env.state_soa[:, :B].cpu(), env.ctr[:B].cpu(), env.life_viol[:B].cpu()]
    if tally:
        got += [env.ep_return[:B].cpu(), env.tally[:, :B].cpu()]
    env.close()
    return got


def _same(a, b):
    assert len(a) == len(b)
    for i, (x, y) in enumerate(zip(a, b)):
        if x.dtype.is_floating_point:
            xi = x.view(torch.int32 if x.dtype == torch.float32 else torch.int64)
            yi = y.view(torch.int32 if y.dtype == torch.float32 else torch.int64)
            assert torch.equal(xi, yi), f"output {i}: {int((xi != yi).sum())} words differ"
        else:
            assert torch.equal(x, y), f"output {i} differs"


@pytest.fixture()
def wide_knob(ni):
    yield
    ni.tune(wide_min_blocks=256)


@pytest.mark.parametrize("outputs", ["none", "min", "last", "soa", "aos"])
def test_pg_lds_rollout_equals_register_rollout(ni, wide_knob, outputs):
    """Three wide blocks + one whole 256-lane block + a ragged tail: the wide kernel runs the first 1536 lanes, the
    register kernel the rest; the same batch entirely on the register kernel must agree in every observable, over
    several launches (ring wrap-around, odd step counts)."""
    B = 3 * 512 + 256 + 37
    a = _pg_run(ni, True, B, [7, 1, 12], outputs, R=5)
    b = _pg_run(ni, False, B, [7, 1, 12], outputs, R=5)
    _same(a, b)


def test_pg_lds_rollout_short_episodes_and_masks(ni, wide_knob):
    """Truncation every 3 steps (every lane resets again and again), a constraint mask, a lane offset, no tally."""
    B = 4 * 512
    for kw in (dict(max_steps=3), dict(cmask=0b101), dict(env_index0=(1 << 33) + 12345), dict(tally=False)):
        a = _pg_run(ni, True, B, [9, 4], "aos", R=4, **kw)
        b = _pg_run(ni, False, B, [9, 4], "aos", R=4, **kw)
        _same(a, b)


def test_pg_lds_rollout_bit_identical_to_oracle_at_baseline_size(ni, wide_knob, oracle):
    """BASELINE configs[2]: 262 144 PowerGrid lanes, the wide kernel over the whole batch (its default), 60 fused steps
    with the row-major trajectory: final state words, step counters, violation / critical / episode counts vs the
    CPU oracle, bit for bit; every trajectory row of the last step equals the final state unless the lane reset."""
    B, T = 262144, 60
    ni.tune(wide_min_blocks=256)
    env = ni.make_batched(PG, B, autoreset=True, tally=True)
    ring = torch.empty(T, env.action_dim, env.ld, dtype=torch.float32, device=env.device)
    for s in range(T):
        env.fill_actions(1 + s, ring[s])           # slot k == the generator's action stream at t = k + 1
    fl = torch.zeros(T, env.ld, dtype=torch.int32, device=env.device)
    rw = torch.zeros(T, env.ld, dtype=torch.float32, device=env.device)
    obs = torch.zeros(T, B, env.state_dim, dtype=torch.float32, device=env.device)
    env.reset()
    env.rollout(T, ring, rw, fl, obs)
    torch.cuda.synchronize()
    st, sc, total, _ = oracle.rollout("pg", B, T, flavor=oracle.MATH_POLY, nthreads=16)
    got = env.get_state().cpu().numpy()
    assert np.array_equal(got.view(np.uint32), st.view(np.uint32))
    assert np.array_equal(env.current_step.cpu().numpy(), sc)
    L = ni._lib
    nv = int(((fl[:, :B] >> L.FLAG_NVIOL_SHIFT) & 3).sum().item())
    nc = int(((fl[:, :B] >> L.FLAG_NCRIT_SHIFT) & 3).sum().item())
    assert (nv, nc) == (total.violations, total.critical)
    assert int(env.tally[L.T_EPISODES].sum().item()) == total.episodes
    keep = ((fl[T - 1, :B] & L.FLAG_DID_RESET) == 0).cpu().numpy()
    last = obs[T - 1].cpu().numpy()
    assert keep.sum() > B // 2 and np.array_equal(last[keep].view(np.uint32), got[keep].view(np.uint32))
    env.close()
