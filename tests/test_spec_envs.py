"""The four README-only environments (HVACControl, WaterTreatment, SteelAnnealing, SupplyChain).

The reference names them in its README (README.md:28-32) and ships no implementation, so there is
NO reference parity to test: these tests pin the build-specified plants (spec_plants.py) as such --
generated tables current, plants behave like controlled processes, and the HIP kernels equal the
independent CPU statement bit for bit in every launch shape.
"""
import os

import numpy as np
import pytest

from conftest import ROOT

SPEC = {"hvac": ("HVACControl-v0", 18, 5), "water": ("WaterTreatment-v0", 15, 4),
        "steel": ("SteelAnnealing-v0", 20, 6), "supply": ("SupplyChain-v0", 28, 10)}


@pytest.fixture(scope="module")
def oracle():
    import oracle.oracle as o
    o.lib()
    return o


def _plants():
    import importlib.util
    spec = importlib.util.spec_from_file_location("spec_plants", os.path.join(ROOT, "neorl-industrial-gym_amd", "spec_plants.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_generated_tables_are_current():
    m = _plants()
    text = m.emit_inc()
    for path in (os.path.join(ROOT, "neorl-industrial-gym_amd", "csrc", "nig_spec_plants.inc"),
                 os.path.join(ROOT, "oracle", "nig_spec_plants.inc")):
        assert open(path).read() == text, path
    for p, (name, S, A) in zip(m.PLANTS, SPEC.values()):
        assert p["name"] == name and m.dims(p) == (S - A - 3, A, S)       # README dims


@pytest.mark.parametrize("key", list(SPEC))
def test_plants_hold_their_setpoints_and_can_be_driven_out(oracle, key):
    """Mid-range actuators and zero action keep every constrained variable inside its box (the
    tables are designed around that operating point); saturated actions drive the plant into its
    constraints and to termination."""
    name, S, A = SPEC[key]
    sp = oracle.spec(name)
    m = _plants()
    P = [p for p in m.PLANTS if p["name"] == name][0]
    st = oracle.reset(name, np.zeros((1, sp.k_reset)))[0]
    assert st.shape == (S,) and np.all(st[S - A - 3:S - 3] == 0.5)
    viol = 0
    for k in range(300):
        r = oracle.step(name, st, np.zeros(A, dtype=np.float32), np.zeros(2), step_pre=k)
        viol += int(r["viol"][0])
        assert np.all(np.isfinite(r["state_next"])) and not r["terminated"][0]
        st = r["state_next"][0]
    assert viol == 0
    assert st[S - 1] == pytest.approx(30.0, rel=1e-4) and st[S - 2] > 0            # elapsed time, effort integral
    # drive every actuator to one end: constraints fire within a few hundred steps
    total = 0
    for sign in (1.0, -1.0):
        st = oracle.reset(name, np.zeros((1, sp.k_reset)))[0]
        for k in range(600):
            r = oracle.step(name, st, np.full(A, sign, dtype=np.float32), np.zeros(2), step_pre=k)
            total += int(r["viol"][0])
            st = r["state_next"][0]
            if r["terminated"][0]:
                break
    assert total > 0, P["name"]


@pytest.mark.parametrize("key", list(SPEC))
def test_random_rollouts_are_nontrivial(oracle, key):
    name, S, A = SPEC[key]
    st, sc, total, _ = oracle.rollout(name, 512, 400, nthreads=4)
    assert np.all(np.isfinite(st)) and total.steps == 512 * 400
    assert total.episodes >= 0 and total.violations >= 0


# ------------------------------------------------------------------------------------------
gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def ni():
    import neorl_industrial_gym_amd as ni
    return ni


@gpu
@pytest.mark.parametrize("key,B,T,max_steps", [("hvac", 4096, 120, None), ("water", 4099, 120, None), ("steel", 2048, 150, 40),
                                                ("supply", 3001, 100, 25)])
def test_step_kernel_bit_identical_to_cpu_statement(ni, oracle, key, B, T, max_steps):
    import torch
    name = SPEC[key][0]
    env = ni.make_batched(name, B, autoreset=True, tally=True, max_episode_steps=max_steps)
    env.reset()
    act = torch.empty(env.action_dim, env.ld, dtype=torch.float32, device=env.device)
    viol = torch.zeros(B, dtype=torch.int64, device=env.device)
    for t in range(1, T + 1):
        env.fill_actions(t, act)
        _, _, te, tr, info = env.step(act[:, :B], layout="soa")
        viol += info.violation_count
    st, sc, total, tallies = oracle.rollout(name, B, T, nthreads=8, per_env=True, max_steps=max_steps)
    assert np.array_equal(env.get_state().cpu().numpy().view(np.uint32), st.view(np.uint32))
    assert np.array_equal(env.current_step.cpu().numpy(), sc)
    assert np.array_equal(viol.cpu().numpy(), np.array([t.violations for t in tallies]))
    L = ni._lib
    assert np.array_equal(env.tally.cpu().numpy()[L.T_EPISODES], np.array([t.episodes for t in tallies]))
    env.close()


@gpu
@pytest.mark.parametrize("key", list(SPEC))
def test_clip_special_values_match_the_cpu_statement(ni, oracle, key):
    """The plant model's clip is one v_med3_f32 on the device and two compares in the CPU statement: signed zeros at
    zero limits, values far outside the limits, infinities and NaN in process variables and actuator positions (and
    -0.0 / huge actions) come out bit-identical, in next state, termination and violation count."""
    import torch
    name, S, A = SPEC[key]
    B, seed = 256, 0x5EED
    env = ni.make_batched(name, B, seed=seed, autoreset=False, tally=False)
    env.reset()
    st = env.get_state().cpu().numpy().copy()
    rng = np.random.default_rng(7)
    special = np.array([-0.0, 0.0, np.nan, np.inf, -np.inf, 1e30, -1e30, 1.0, -1.0, 1e-40, -1e-40], dtype=np.float32)
    for i in range(B):
        for c in rng.choice(S - 3, size=1 + i % 4, replace=False):       # process variables and actuator positions
            st[i, c] = special[rng.integers(len(special))]
    env.set_state(st)
    act = rng.uniform(-1.0, 1.0, size=(B, A)).astype(np.float32)
    act[::5, 0] = -0.0
    act[1::7, A - 1] = 1e30
    a_dev = torch.as_tensor(act.T.copy(), device=env.device)
    obs, rew, te, tr, info = env.step(a_dev, layout="soa")
    got = env.get_state().cpu().numpy()
    noise = np.stack([oracle.gen_step_noise(name, seed, i, 1) for i in range(B)])
    act_clipped = np.clip(act, -1.0, 1.0)                                  # base.py:167 (finite actions here)
    r = oracle.step(name, st, act_clipped, noise, np.zeros(B, dtype=np.int32), flavor=oracle.MATH_POLY)
    assert np.array_equal(got.view(np.uint32), r["state_next"].view(np.uint32))
    assert np.array_equal(te.cpu().numpy().astype(bool), r["terminated"] != 0)
    assert np.array_equal(info.violation_count.cpu().numpy(), r["viol"])
    env.close()


@gpu
@pytest.mark.parametrize("key", list(SPEC))
def test_fused_rollout_all_output_shapes(ni, oracle, key):
    """Fused rollout == step calls == CPU statement; row-major and [S][ld] trajectories for state dims
    that are not multiples of 4 (18, 15), partial last wave, chunked launches."""
    import torch
    name, S, A = SPEC[key]
    B, T, R = 1004, 30, 7          # B multiple of 4: row-major rows of 15 floats stay 16-byte aligned per step
    a = ni.make_batched(name, B, autoreset=True, tally=True, max_episode_steps=12)
    b = ni.make_batched(name, B, autoreset=True, tally=True, max_episode_steps=12)
    c = ni.make_batched(name, B, autoreset=True, tally=True, max_episode_steps=12)
    dev, ld = a.device, a.ld
    ring = torch.empty(R, A, ld, dtype=torch.float32, device=dev)
    for s in range(R):
        a.fill_actions(s + 1, ring[s])
    rw = torch.zeros(T, ld, dtype=torch.float32, device=dev); fl = torch.zeros(T, ld, dtype=torch.int32, device=dev)
    rm = torch.zeros(T, B, S, dtype=torch.float32, device=dev); so = torch.zeros(T, S, ld, dtype=torch.float32, device=dev)
    rw2 = torch.zeros_like(rw); fl2 = torch.zeros_like(fl)
    a.reset(); b.reset(); c.reset()
    a.rollout(T, ring, rw, fl, rm)
    k = 0
    for n in (7, 7, 7, 7, 2):                                  # chunked, [S][ld] rows; ring restarts per call
        rot = torch.roll(ring, shifts=-(k % R), dims=0).contiguous()
        c.rollout(n, rot, rw2[k:k + n], fl2[k:k + n], so[k:k + n])
        k += n
    assert torch.equal(rw, rw2) and torch.equal(fl, fl2)
    assert torch.equal(rm.permute(0, 2, 1).contiguous().view(torch.int32), so[:, :, :B].contiguous().view(torch.int32))
    for t in range(T):
        _, r, te, tr, info = b.step(ring[t % R][:, :B], layout="soa")
        assert torch.equal(rw[t, :B], r) and torch.equal(fl[t, :B], info.flags), t
    assert torch.equal(a.state_soa.view(torch.int32), b.state_soa.view(torch.int32))
    assert torch.equal(a.state_soa.view(torch.int32), c.state_soa.view(torch.int32))
    assert torch.equal(a.ctr, b.ctr) and torch.equal(a.life_viol, b.life_viol)
    for e in (a, b, c):
        e.close()
    # and against the CPU statement (ring slot k == generator action stream at t = k + 1)
    d = ni.make_batched(name, B, autoreset=True, tally=True, max_episode_steps=12)
    ring2 = torch.empty(T, A, ld, dtype=torch.float32, device=dev)
    for s in range(T):
        d.fill_actions(s + 1, ring2[s])
    d.reset(); d.rollout(T, ring2)
    st, sc, total, _ = oracle.rollout(name, B, T, nthreads=4, max_steps=12)
    assert np.array_equal(d.get_state().cpu().numpy().view(np.uint32), st.view(np.uint32))
    assert np.array_equal(d.current_step.cpu().numpy(), sc)
    d.close()


@gpu
@pytest.mark.parametrize("key", list(SPEC))
def test_policy_rollout_bit_identical(ni, oracle, key):
    """Closed loop with an on-device affine policy (10 actions for SupplyChain)."""
    import torch
    name, S, A = SPEC[key]
    B, T = 700, 40
    rng = np.random.default_rng(3)
    W = np.zeros((A, S), dtype=np.float32)
    W[:, :5] = rng.normal(0, 0.01, (A, 5))
    pol = ni.DevicePolicy(S, A, W=W, b=rng.normal(0, 0.2, A), sigma=np.full(A, 0.1), half_range=np.linspace(0, 0.2, A),
                          p_uniform=0.1, uniform_range=0.9, clip=(-1.0, 1.0))
    for autoreset in (False, True):
        env = ni.make_batched(name, B, autoreset=autoreset, tally=True, max_episode_steps=25)
        env.set_policy(pol)
        obs = torch.zeros(T, B, S, dtype=torch.float32, device=env.device)
        act = torch.zeros(T, A, env.ld, dtype=torch.float32, device=env.device)
        env.reset()
        env.rollout_policy(T, None, None, obs, act)
        P = oracle.Policy.from_buffer_copy(bytes(pol.to_struct()))
        r = oracle.rollout_policy(name, B, T, P, max_steps=25, autoreset=autoreset, trajectories=True)
        assert np.array_equal(env.get_state().cpu().numpy().view(np.uint32), r["state"].view(np.uint32))
        live = r["live"].astype(bool)
        assert np.array_equal(obs.cpu().numpy().view(np.uint32)[live], r["obs"].view(np.uint32)[live])
        assert np.array_equal(act[:, :, :B].permute(0, 2, 1).cpu().numpy().view(np.uint32)[live], r["act"].view(np.uint32)[live])
        env.close()


@gpu
@pytest.mark.parametrize("key", list(SPEC))
def test_single_env_surface(ni, oracle, key):
    """ni.make(...) returns the base-class surface; the host-side check functions of the constraint objects
    agree with the violation bits the device reports."""
    name, S, A = SPEC[key]
    np.random.seed(11)
    env = ni.make(name)
    assert env.observation_space.shape == (S,) and env.action_space.shape == (A,)
    assert [c.name for c in env.safety_constraints] == [c[0] for c in env._PLANT["constraints"]]
    obs, info = env.reset()
    assert obs.shape == (S,) and obs.dtype == np.float32
    seen = 0
    for k in range(400):
        prev = obs
        a = np.full(A, 1.0 if (k // 100) % 2 == 0 else -1.0, dtype=np.float32)
        obs, rew, term, trunc, info = env.step(a)
        want = [not c.check_fn(prev, a) for c in env.safety_constraints]
        m = env.get_safety_metrics()
        assert m.violation_count == sum(want), k
        seen += sum(want)
        if term or trunc:
            obs, info = env.reset()
    assert seen > 0
    env.close()
    if A > 16 or S % 2:                                 # the MFMA actor exists for <= 16 actions and even state dims
        b = ni.make_batched(name, 64)
        z = np.zeros
        with pytest.raises(ni._lib.NigError):
            b.set_mlp_policy([(z((S, 256)), z(256)), (z((256, 256)), z(256)), (z((256, A)), z(A))])
        b.close()


@gpu
def test_seven_env_mixed_batch(ni):
    """BASELINE config 'all 7 envs mixed-batch, padded SoA': the README's seven in one padded matrix."""
    import torch
    names = ["ChemicalReactor-v0", "RobotAssembly-v0", "HVACControl-v0", "WaterTreatment-v0", "SteelAnnealing-v0",
             "PowerGrid-v0", "SupplyChain-v0"]
    mix = ni.MixedBatchedEnv([(n, 640) for n in names])
    assert mix.state_soa.shape[0] == 32 and mix.state_soa.shape[1] >= 7 * 640
    mix.reset()
    for e, off in zip(mix.envs, mix.offsets):
        alone = ni.make_batched(e.env_id, e.batch, env_index0=off, autoreset=True)
        alone.reset()
        ring = torch.empty(3, e.action_dim, e.ld, dtype=torch.float32, device=e.device)
        for s in range(3):
            e.fill_actions(s + 1, ring[s])
        e.rollout(9, ring); alone.rollout(9, ring)
        torch.cuda.synchronize()
        assert torch.equal(e.get_state().view(torch.int32), alone.get_state().view(torch.int32)), e.env_id
        alone.close()
    mix.close()


@gpu
@pytest.mark.parametrize("key", list(SPEC))
def test_get_dataset_for_spec_envs(ni, key):
    """get_dataset works for the build-specified envs too (behaviour policy = proportional control read off the
    plant table); the expert set is better than the random one."""
    name, S, A = SPEC[key]
    env = ni.make_batched(name, 256, autoreset=False, tally=True)
    d = {q: env.get_dataset(q) for q in ("expert", "random")}
    for q, ds in d.items():
        n = ds["observations"].shape[0]
        assert ds["observations"].shape == (n, S) and ds["actions"].shape == (n, A)
        assert ds["rewards"].shape == (n,) and ds["terminals"].shape == (n,) and n > 1000
        assert float(ds["actions"].abs().max()) <= 1.0
    assert float(d["expert"]["rewards"].mean()) > float(d["random"]["rewards"].mean())
    env.close()


@gpu
@pytest.mark.parametrize("key", list(SPEC))
def test_evaluate_with_safety_on_spec_envs(ni, key):
    """evaluate_with_safety (utils.py:42-154 surface) on the build-specified envs: batched on the device, and the
    single-env host loop; a do-nothing agent beats a random one (the plants rest at their operating point)."""
    name, S, A = SPEC[key]
    keys13 = None
    res = {}
    for tag, agent in (("hold", ni.constant_agent(S, A)), ("random", ni.random_agent(S, A))):
        env = ni.make_batched(name, 128, tally=True, autoreset=False, max_episode_steps=200)
        res[tag] = ni.evaluate_with_safety(agent, env, n_episodes=128)
        keys13 = keys13 or set(res[tag])
        assert set(res[tag]) == keys13 and len(keys13) == 13
        assert all(np.isfinite(v) for v in res[tag].values() if isinstance(v, (int, float)))
        env.close()
    assert res["hold"]["return_mean"] > res["random"]["return_mean"]
    assert res["hold"]["safety_violations"] <= res["random"]["safety_violations"]
    np.random.seed(3)
    single = ni.make(name, max_episode_steps=30)
    out = ni.evaluate_with_safety(ni.constant_agent(S, A), single, n_episodes=2)
    assert set(out) == keys13 and out["length_mean"] == 30.0
    single.close()
