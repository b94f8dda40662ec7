"""-m gpu: round-3 additions pinned on the device: bench.py's own N-rank launch with the real workload; PowerGrid's
LDS-resident rollout (csrc/nig_pg_lds.hpp) against the register-resident kernel and the oracle, up to the maximum batch;
the reference-side pins of the get_dataset behaviour laws (device policy kernel), ChemicalReactor's info dicts and the
three-wave rollout directly against oracle trajectories."""
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
                        "--no-parity", "--no-mixed", "--no-robotassembly", "--no-brackets"], env=env, capture_output=True, text=True, timeout=900)
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
    """wide: True = the LDS-resident body, whole 512-lane blocks in the wide form; an int = the knob value (256: the
    default -- a small batch then runs the same body in 256-lane blocks); "pair" = default knobs: a batch of at most one
    256-lane block per compute unit runs the paired form (producer waves, rollout_pg_pair_kernel); False = rollout_kernel
    only."""
    ni.tune(wide_min_blocks=NEVER if wide is False else (1 if wide is True else 256 if wide == "pair" else int(wide)),
            split_blocks=256 if wide == "pair" else 0)
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
    ni.tune(wide_min_blocks=256, split_blocks=256)


@pytest.mark.parametrize("outputs", ["none", "min", "last", "soa", "aos"])
@pytest.mark.parametrize("knob", [True, 256, "pair"])
def test_pg_lds_rollout_equals_register_rollout(ni, wide_knob, outputs, knob):
    """Three wide blocks + one whole 256-lane block + a ragged tail.  knob True: the wide (512-lane) form runs the first
    1536 lanes, the 256-lane form of the same LDS-resident body the next block, rollout_kernel the ragged tail; knob 256:
    the batch is below the wide threshold, so all seven whole blocks run the 256-lane form; "pair" (the defaults): those
    seven blocks run the paired form, a producer wave drawing the normals of every stepping wave.  Either way
    every observable equals the same batch entirely on the register-resident kernel, over several launches (ring
    wrap-around, odd step counts)."""
    B = 3 * 512 + 256 + 37
    a = _pg_run(ni, knob, B, [7, 1, 12], outputs, R=5)
    b = _pg_run(ni, False, B, [7, 1, 12], outputs, R=5)
    _same(a, b)


def test_pg_lds_rollout_short_episodes_and_masks(ni, wide_knob):
    """Truncation every 3 steps (every lane resets again and again), a constraint mask, a lane offset, no tally."""
    B = 4 * 512
    for kw in (dict(max_steps=3), dict(cmask=0b101), dict(env_index0=(1 << 33) + 12345), dict(tally=False)):
        b = _pg_run(ni, False, B, [9, 4], "aos", R=4, **kw)
        for form in (True, "pair"):
            a = _pg_run(ni, form, B, [9, 4], "aos", R=4, **kw)
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


def test_pg_paired_form_at_one_block_per_compute_unit_against_oracle(ni, wide_knob, oracle):
    """The paired form at the size it is for: 65 536 PowerGrid lanes = one 256-lane block with its four producer waves on
    every compute unit, 2 x 150 + 1 fused steps (the two-slot noise rings wrap 150 times under real contention; a protocol
    slip would show as a mismatch or a hang of this call).  bench.py's naming rule says which kernel ran; final state,
    step counters, violation / critical / episode totals equal the CPU oracle's, bit for bit."""
    import types
    import bench
    B, T = 65536, 301
    ni.tune(wide_min_blocks=256, split_blocks=256)
    assert bench.rollout_kernel_name(types.SimpleNamespace(key="pg", B=B, outputs="min", ni=ni)) == "rollout_pg_pair_kernel<1>"
    env = ni.make_batched(PG, B, autoreset=True, tally=True)
    ring = torch.empty(T, env.action_dim, env.ld, dtype=torch.float32, device=env.device)
    for s in range(T):
        env.fill_actions(1 + s, ring[s])
    fl = torch.zeros(T, env.ld, dtype=torch.int32, device=env.device)
    rw = torch.zeros(T, env.ld, dtype=torch.float32, device=env.device)
    env.reset()
    env.rollout(150, ring[:150], rw[:150], fl[:150])
    env.rollout(150, ring[150:300], rw[150:300], fl[150:300])
    env.rollout(1, ring[300:], rw[300:], fl[300:])
    torch.cuda.synchronize()
    st, sc, total, _ = oracle.rollout("pg", B, T, flavor=oracle.MATH_POLY, nthreads=16)
    assert np.array_equal(env.get_state().cpu().numpy().view(np.uint32), st.view(np.uint32))
    assert np.array_equal(env.current_step.cpu().numpy(), sc)
    L = ni._lib
    nv = int(((fl[:, :B] >> L.FLAG_NVIOL_SHIFT) & 3).sum().item())
    nc = int(((fl[:, :B] >> L.FLAG_NCRIT_SHIFT) & 3).sum().item())
    assert (nv, nc) == (total.violations, total.critical)
    assert int(env.tally[L.T_EPISODES].sum().item()) == total.episodes
    env.close()


# ---------------------------------------------------------------------------------------------------------------
# Reference-side pins of what round 2 held only statistically or transitively
# ---------------------------------------------------------------------------------------------------------------
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.mark.parametrize("key", KEYS)
@pytest.mark.parametrize("quality", ["expert", "medium", "mixed", "random"])
def test_device_policy_kernel_law_against_reference(ni, key, quality):
    """The policy kernel's action (rollout_policy_kernel / split_policy_kernel, noise amplitudes set to zero) on the
    observations of tests/golden/behaviour_laws.npz == the reference's get_dataset law with its draws patched to zero
    (chemical_reactor.py:364-393, power_grid.py:216-233, robot_assembly.py:266-290): 1e-5 with an absolute floor at
    float32 resolution of O(1) terms (see tests/test_oracle_golden.py law_error); and == the host predict() bit for bit."""
    from neorl_industrial_gym_amd.policies import behaviour_policy
    g = dict(np.load(os.path.join(GOLDEN, "behaviour_laws.npz")))
    obs, want = g[f"{key}_{quality}_obs"], g[f"{key}_{quality}_act"]
    n = obs.shape[0]
    pol = behaviour_policy(ENV_NAME[key], quality)
    pol.sigma[:] = 0
    pol.half_range[:] = 0
    pol.p_uniform = np.float32(0)
    env = ni.make_batched(ENV_NAME[key], n, autoreset=False)
    env.reset()
    env.set_state(obs, current_step=0)
    env.set_policy(pol)
    act = torch.full((1, env.action_dim, env.ld), float("nan"), dtype=torch.float32, device=env.device)
    seen = torch.zeros(1, n, env.state_dim, dtype=torch.float32, device=env.device)
    env.rollout_policy(1, obs_out=seen, act_out=act)
    torch.cuda.synchronize()
    got = act[0, :, :n].t().cpu().numpy()
    assert np.array_equal(seen[0].cpu().numpy().view(np.uint32), obs.view(np.uint32))          # the policy acted on these rows
    host = behaviour_policy(ENV_NAME[key], quality).predict(obs)
    assert np.array_equal(got.view(np.uint32), host.view(np.uint32))
    err = np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want.astype(np.float64)), 0.2)
    assert err.max() <= 1e-5, (key, quality, float(err.max()))
    env.close()


def test_chemical_reactor_info_dicts_against_reference(ni):
    """ChemicalReactorEnv._get_safety_info (chemical_reactor.py:307-322): reset-time safety_metrics values, the
    constraint_values margins and the step / violations / total_violations / critical_shutdown entries of every info
    dict along the reference's own seeded run (tests/golden/cr_info.npz, np.random.seed(4321), two episodes)."""
    g = dict(np.load(os.path.join(GOLDEN, "cr_info.npz")))
    assert str(g["margin_type"]) == "float32" and str(g["bounds_type"]) == "bool"
    env = ni.make("ChemicalReactor-v0")
    np.random.seed(4321)
    i, n = 0, len(g["step"])
    while i < n:
        assert g["is_reset"][i] == 1
        obs, info = env.reset()
        assert np.array_equal(obs.view(np.uint32), g["obs"][i].view(np.uint32))
        sm, cv = info["safety_metrics"], info["constraint_values"]
        assert [np.float32(sm[k]) for k in ("temperature", "pressure", "level", "emergency_stop", "alarm_status")] == list(g["sm_reset"][i])
        assert isinstance(cv["temp_margin"], np.float32) and isinstance(cv["level_in_bounds"], (bool, np.bool_))
        assert (np.float32(cv["temp_margin"]), np.float32(cv["pressure_margin"]), int(cv["level_in_bounds"])) == \
               (np.float32(g["temp_margin"][i]), np.float32(g["pressure_margin"][i]), int(g["level_in_bounds"][i]))
        assert (info["step"], info["violations"], info["total_violations"]) == (g["step"][i], g["violations"][i], g["total_violations"][i])
        i += 1
        while i < n and g["is_reset"][i] == 0:
            obs, reward, term, trunc, info = env.step(g["action"][i])
            cv = info["constraint_values"]
            # states follow the reference within the parity bar (np.exp in the concentration row), margins with them
            assert rel_err_np(obs, g["obs"][i]) <= 1e-5, i
            assert abs(float(cv["temp_margin"]) - g["temp_margin"][i]) <= 1e-5 * max(1.0, abs(g["temp_margin"][i]))
            assert abs(float(cv["pressure_margin"]) - g["pressure_margin"][i]) <= 1e-5 * max(1.0, abs(g["pressure_margin"][i]))
            assert int(cv["level_in_bounds"]) == g["level_in_bounds"][i]
            assert (info["step"], info["violations"], info["total_violations"], int(info["critical_shutdown"])) == \
                   (g["step"][i], g["violations"][i], g["total_violations"][i], g["critical_shutdown"][i]), i
            i += 1
    assert i == n


def rel_err_np(a, b, floor=1e-6):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float((np.abs(a - b) / np.maximum(np.abs(b), floor)).max())


@pytest.mark.parametrize("name,key,kernel", [("ChemicalReactor-v0", "cr", "split_rollout_kernel<ChemicalReactor,3,4>"),
                                             ("RobotAssembly-v0", "ra", "split_rollout_kernel<RobotAssembly,3,4>")])
def test_three_wave_form_rows_against_oracle_trajectories(ni, oracle, name, key, kernel):
    """The three-wave rollout (csrc/nig_split.hpp) checked DIRECTLY against the oracle, not through the one-wave form:
    the knob is forced, the host rule is asserted to select the three-wave kernel, and every row-major trajectory row,
    reward and termination flag of the first and last wave of the batch is compared step by step with the oracle's
    teacher-forced step fed by the generator's own draws (terminal observations included), then the final state."""
    import types
    import bench
    B, T, seed = 1024, 40, 0x5EED
    before = ni.tune()["split_blocks"]
    ni.tune(split_blocks=256)
    try:
        assert bench.rollout_kernel_name(types.SimpleNamespace(key=key, B=B, outputs="full", ni=ni)) == kernel
        env = ni.make_batched(name, B, seed=seed, autoreset=True, tally=True, max_episode_steps=23)    # truncation: resets inside the window
        ring = torch.empty(T, env.action_dim, env.ld, dtype=torch.float32, device=env.device)
        for s in range(T):
            env.fill_actions(1 + s, ring[s])
        rew = torch.zeros(T, env.ld, dtype=torch.float32, device=env.device)
        fl = torch.zeros(T, env.ld, dtype=torch.int32, device=env.device)
        obs = torch.zeros(T, B, env.state_dim, dtype=torch.float32, device=env.device)
        env.reset()
        assert env.counter % 2 == 0                    # the first step is an odd launch counter: the PAIRED (three-wave) form applies
        env.rollout(T, ring, rew, fl, obs)
        torch.cuda.synchronize()
        final = env.get_state().cpu().numpy()
        L = ni._lib
        sp = oracle.spec(name)
        for lo in (0, B - 64):
            acts = ring[:, :, lo:lo + 64].permute(0, 2, 1).contiguous().cpu().numpy()
            state = np.stack([oracle.reset(name, oracle.gen_reset_noise(name, seed, lo + i, 0), flavor=oracle.MATH_POLY)[0] for i in range(64)])
            step = np.zeros(64, dtype=np.int32)
            for k in range(T):
                t = k + 1
                noise = np.stack([oracle.gen_step_noise(name, seed, lo + i, t) for i in range(64)])
                r = oracle.step(name, state, acts[k], noise, step, max_steps=23, flavor=oracle.MATH_POLY)
                row = obs[k, lo:lo + 64].cpu().numpy()
                assert np.array_equal(row.view(np.uint32), r["state_next"].view(np.uint32)), (lo, k)
                assert np.array_equal(rew[k, lo:lo + 64].cpu().numpy(), r["reward"].astype(np.float32)), (lo, k)
                f = fl[k, lo:lo + 64].cpu().numpy()
                assert np.array_equal((f & L.FLAG_TERMINATED) != 0, r["terminated"] != 0) and np.array_equal((f & L.FLAG_TRUNCATED) != 0, r["truncated"] != 0)
                state, step = r["state_next"].copy(), step + 1
                for i in np.nonzero((r["terminated"] | r["truncated"]) != 0)[0]:
                    state[i] = oracle.reset(name, oracle.gen_reset_noise(name, seed, lo + i, t), flavor=oracle.MATH_POLY)[0]
                    step[i] = 0
            assert np.array_equal(final[lo:lo + 64].view(np.uint32), state.view(np.uint32)), lo
        env.close()
    finally:
        ni.tune(split_blocks=before)


def test_pg_lds_rollout_at_the_maximum_batch(ni, wide_knob, oracle):
    """NIG_MAX_BATCH = 2^24 PowerGrid lanes (32 768 wide blocks; a 2.1 GB state matrix and 2.1 GB of row-major rows per
    step: every row offset of the LDS-resident kernel at the far end of its range): the first and the last 4 096 lanes
    equal the oracle bit for bit after three fused steps, the last step's rows equal the final state where no reset fell."""
    B, T = 1 << 24, 3
    ni.tune(wide_min_blocks=256)
    env = ni.make_batched(PG, B, autoreset=True, tally=False)
    ring = torch.empty(T, env.action_dim, env.ld, dtype=torch.float32, device=env.device)
    for s in range(T):
        env.fill_actions(s + 1, ring[s])
    rew = torch.zeros(env.ld, dtype=torch.float32, device=env.device)
    fl = torch.zeros(env.ld, dtype=torch.int32, device=env.device)
    obs = torch.zeros(1, B, env.state_dim, dtype=torch.float32, device=env.device)
    env.reset()
    env.rollout(2, ring[0:2], rew, fl)
    env.rollout(1, ring[2:3], rew, fl, obs)
    torch.cuda.synchronize()
    assert env.counter == T
    for lo in (0, B - 4096):
        st, sc, _, _ = oracle.rollout("pg", 4096, T, env0=lo, flavor=oracle.MATH_POLY)
        got = env.state_soa[:, lo:lo + 4096].t().contiguous().cpu().numpy()
        assert np.array_equal(got.view(np.uint32), st.view(np.uint32)), lo
        assert np.array_equal(env.current_step[lo:lo + 4096].cpu().numpy(), sc), lo
        keep = (fl[lo:lo + 4096] & ni._lib.FLAG_DID_RESET) == 0
        o = obs[0, lo:lo + 4096]
        assert int(keep.sum()) > 2000
        assert torch.equal(o[keep].view(torch.int32), env.state_soa[:, lo:lo + 4096].t()[keep].contiguous().view(torch.int32))
    env.close()


@pytest.mark.parametrize("dt", [None, 0.02, 1e-40, 3e35])
def test_robot_assembly_clip_division_and_special_states(ni, oracle, dt):
    """RobotAssembly's rarely taken paths against the oracle, bit for bit (NaN == NaN): joints beyond +-pi (the fp64
    clip that the kernel skips unless a lane needs it, robot_assembly.py:149-153), -0.0 / NaN joints, infinite and NaN
    previous positions (the velocity division's special cases, :159-160), actions at and beyond the limits; with the
    default dt, another ordinary one (division through the reciprocal, csrc/nig_detmath.hpp ddiv_y) and two far outside
    the range that path is used for (IEEE division), one of them subnormal in float32."""
    name, B, seed = "RobotAssembly-v0", 512, 0x5EED
    kw = {} if dt is None else {"dt": dt}
    env = ni.make_batched(name, B, seed=seed, autoreset=False, tally=False, **kw)
    env.reset()
    st = env.get_state().cpu().numpy().copy()
    rng = np.random.default_rng(11)
    pi32 = np.float32(np.pi)
    for i in range(B):
        k = i % 8
        if k == 0:   st[i, 7 + rng.integers(7)] = pi32                       # the float above the double pi
        elif k == 1: st[i, 7 + rng.integers(7)] = -pi32 - np.float32(0.05)
        elif k == 2: st[i, 7 + rng.integers(7)] = np.float32(3.1415925)     # the largest float inside
        elif k == 3: st[i, 7 + rng.integers(7)] = np.float32([np.nan, -0.0, 40.0, -1e6][rng.integers(4)])
        elif k == 4: st[i, rng.integers(3)] = np.float32([np.inf, -np.inf, np.nan, -0.0, 1e30][rng.integers(5)])
        # k = 5, 6, 7: ordinary rows in the same waves
    env.set_state(st)
    act = rng.uniform(-1.0, 1.0, size=(B, 7)).astype(np.float32)
    act[::3] = np.sign(act[::3])                                             # +-1: pushes joints at the limit across it
    act[5::16, 2] = 7.5
    env.step(act)
    got = env.get_state().cpu().numpy()
    o = oracle.step(name, st, act, None, np.zeros(B, dtype=np.int32), dt=env.dt, flavor=oracle.MATH_POLY)
    same = (got.view(np.uint32) == o["state_next"].view(np.uint32)) | (np.isnan(got) & np.isnan(o["state_next"]))
    assert same.all(), np.argwhere(~same)[:10]
    fl = env.flags.cpu().numpy()
    assert np.array_equal((fl & ni._lib.FLAG_TERMINATED) != 0, o["terminated"] != 0)
    with np.errstate(over="ignore"):
        want = o["reward"].astype(np.float32)
    assert np.array_equal(env.reward.cpu().numpy(), want, equal_nan=True)
    env.close()
