"""-m gpu: the reference's RECORDED rollouts (tests/golden/<env>_g3.npz: 64 episodes per env, every np.random draw
of _get_initial_state and _dynamics recorded by oracle/gen_golden.py) stepped by the FUSED rollout kernels the bench
times -- nig_rollout_noise runs the same kernel form nig_rollout selects, with the generator's draws replaced by the
recorded ones.  One hop from `environments/base.py:157-213` + `chemical_reactor.py:149,159` / `power_grid.py:136-144` to
split_rollout_kernel / rollout_wide_kernel / rollout_pg_pair_kernel / rollout_kernel (round 3 pinned those kernels
through the oracle only: fused == oracle(generator) bit for bit, oracle == fixtures).

Layout of a test batch: lane l plays the recorded episodes (l mod E), (l mod E) + 1, ... back to back -- the handle
auto-resets, and the reset row set of a step holds the NEXT episode's recorded initial-state draws -- so every lane's
expected per-step observation / reward / termination / violation stream is the concatenation of fixture episodes.
Tolerance: float32 outputs within 1e-5 relative (abs floor 1e-6); integer outputs exact."""
import types

import numpy as np
import pytest
import torch

from conftest import ENV_NAME, RTOL, load_golden, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ni():
    import neorl_industrial_gym_amd as ni
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    yield ni
    ni.tune(split_blocks=-1, wide_min_blocks=-1)


def _chain(d, B, T):
    """Per lane and step: action, step draws, reset draws (of the episode that starts if this step ends one) and the
    fixture's outputs, for lanes that play episodes (l % E), (l % E) + 1, ... back to back."""
    off, E = d["ep_offsets"], len(d["ep_length"])
    A, K, S, KR = d["action"].shape[1], d["noise"].shape[1], d["obs"].shape[1], d["ep_init_noise"].shape[1]
    act = np.zeros((T, A, B), np.float32)
    nz = np.zeros((T, max(K, 1), B))
    rz = np.zeros((T, KR, B))
    idx = np.zeros((T, B), np.int64)            # fixture row of (step, lane)
    first = np.zeros(B, np.int64)
    for l in range(min(B, E)):                  # lane l and lane l + E play the same episodes: build E lanes, tile below
        e, t = l % E, 0
        first[l] = e
        while t < T:
            n = min(int(d["ep_length"][e]), T - t)
            j = off[e] + np.arange(n)
            idx[t:t + n, l] = j
            act[t:t + n, :, l] = d["action"][j]
            if K:
                nz[t:t + n, :, l] = d["noise"][j]
            nxt = (e + 1) % E
            rz[t:t + n, :, l] = d["ep_init_noise"][nxt]      # only the episode's last step consumes it
            t += n
            e = nxt
    for l0 in range(E, B, E):
        w = min(E, B - l0)
        act[:, :, l0:l0 + w] = act[:, :, :w]; nz[:, :, l0:l0 + w] = nz[:, :, :w]; rz[:, :, l0:l0 + w] = rz[:, :, :w]
        idx[:, l0:l0 + w] = idx[:, :w]; first[l0:l0 + w] = first[:w]
    return act, nz, rz, idx, first


FORMS = [
    # (id, env key, lanes, steps, tune(split_blocks, wide_min_blocks), autoreset, kernel bench.rollout_kernel_name must report)
    ("three-wave-cr", "cr", 256, 1300, (256, 256), True, "split_rollout_kernel<ChemicalReactor,3,4>"),
    ("three-wave-ra", "ra", 256, 240, (256, 256), True, "split_rollout_kernel<RobotAssembly,3,4>"),
    # the BASELINE headline shape: one 256-lane block per compute unit, every ring of the chip under contention
    ("three-wave-cr-65536", "cr", 65536, 120, (256, 256), True, "split_rollout_kernel<ChemicalReactor,3,4>"),
    ("wide-512-pg-262144", "pg", 262144, 24, (256, 256), True, "rollout_wide_kernel<PowerGrid,3,512>"),
    ("wide-512-pg", "pg", 1024, 96, (0, 1), True, "rollout_wide_kernel<PowerGrid,3,512>"),
    ("wide-256-pg", "pg", 512, 96, (0, 256), True, "rollout_wide_kernel<PowerGrid,3,256>"),
    ("pair-pg", "pg", 512, 96, (256, 256), True, "rollout_pg_pair_kernel<3>"),
    ("one-wave-cr", "cr", 256, 1300, (0, 1 << 30), True, "rollout_kernel<ChemicalReactor,3>"),
    ("one-wave-pg", "pg", 320, 96, (0, 1 << 30), True, "rollout_kernel<PowerGrid,3>"),
    ("one-wave-ra", "ra", 200, 240, (0, 1 << 30), True, "rollout_kernel<RobotAssembly,3>"),
    # no auto-reset: one recorded episode per lane, frozen after its end (the handle cannot take the other forms)
    ("one-wave-cr-frozen", "cr", 64, 520, (256, 256), False, None),
    ("one-wave-pg-frozen", "pg", 64, 40, (256, 1), False, None),
]


@pytest.mark.parametrize("form", FORMS, ids=[f[0] for f in FORMS])
def test_recorded_reference_rollouts_through_the_fused_kernels(ni, form):
    import bench
    _, key, B, T, (split_blocks, wide_min), autoreset, kernel = form
    d = load_golden(key, "g3")
    E = len(d["ep_length"])
    S, K = d["obs"].shape[1], d["noise"].shape[1]
    ni.tune(split_blocks=split_blocks, wide_min_blocks=wide_min)
    if kernel is not None:
        assert bench.rollout_kernel_name(types.SimpleNamespace(key=key, B=B, outputs="full", ni=ni)) == kernel
    if not autoreset:
        T = min(T, int(d["ep_length"].max()))
    act, nz, rz, idx, first = _chain(d, B, T)
    dev = "cuda"
    env = ni.make_batched(ENV_NAME[key], B, autoreset=autoreset, tally=True)
    env.reset(init_noise=d["ep_init_noise"][first].T)
    ld = env.ld
    pad = lambda x: torch.from_numpy(np.concatenate([x, np.zeros(x.shape[:-1] + (ld - B,), x.dtype)], -1)).to(dev)
    ring = pad(act)
    nzt = pad(nz) if K else None
    rzt = pad(rz) if autoreset else None
    rew = torch.zeros(T, ld, dtype=torch.float32, device=dev)
    fl = torch.zeros(T, ld, dtype=torch.int32, device=dev)
    obs = torch.full((T, B, S), float("nan"), dtype=torch.float32, device=dev)
    # two launches, the second from an odd and the first from an even launch counter in turn: the chunk boundary and the
    # pairing of launch counters must not matter to the injected-draw kernels
    cut = T // 3 + 1
    for t0, t1 in ((0, cut), (cut, T)):
        env.rollout_noise(t1 - t0, ring[t0:t1], None if nzt is None else nzt[t0:t1], None if rzt is None else rzt[t0:t1],
                          rew[t0:t1], fl[t0:t1], obs[t0:t1])
    torch.cuda.synchronize()
    rew, fl, obs = rew.cpu().numpy()[:, :B], fl.cpu().numpy()[:, :B].view(np.uint32), obs.cpu().numpy()
    L = ni._lib
    if autoreset:
        live = np.ones((T, B), bool)
    else:
        live = np.arange(T)[:, None] < d["ep_length"][first][None, :]
        assert np.all((fl[~live] & L.FLAG_INACTIVE) != 0)
    want_obs, want_rew = d["obs"][idx], d["reward"][idx]
    assert rel_err(obs[live], want_obs[live]).max() <= RTOL
    assert rel_err(rew[live], want_rew[live]).max() <= RTOL
    # float32 state words: PowerGrid bit-exact against the reference along the whole free-running chain; RobotAssembly
    # except the velocity rows (nothing reads them back: tests/test_gpu_parity.py G1).  ChemicalReactor's concentration
    # is one np.exp ulp off now and then and feeds the next step's pressure and temperature, so in a FREE-RUNNING
    # chain the last-bit differences spread (teacher-forced single steps, G1, keep them to column 4): 1e-5 above is the bar.
    same = (obs.view(np.uint32) == want_obs.view(np.uint32)) | ~live[:, :, None]
    cols = set(np.unique(np.where(~same)[2]).tolist())
    if key != "cr":
        assert cols <= ({14, 15, 16} if key == "ra" else set()), cols
    f = fl[live]
    assert np.array_equal((f & L.FLAG_TERMINATED) != 0, d["terminated"][idx][live] != 0)
    assert np.array_equal((f & L.FLAG_TRUNCATED) != 0, d["truncated"][idx][live] != 0)
    assert np.array_equal((f >> L.FLAG_NVIOL_SHIFT) & 3, d["viol"][idx][live])
    assert np.array_equal((f >> L.FLAG_NCRIT_SHIFT) & 3, d["crit"][idx][live])
    done = (d["terminated"][idx] | d["truncated"][idx]) != 0
    assert np.array_equal((f & L.FLAG_DID_RESET) != 0, done[live] & autoreset)
    # the restart states: a lane that finished in step t starts step t + 1 from the reference's recorded initial state
    # (checked through the state the handle holds after the launch, and implicitly by every later observation)
    state = env.get_state().cpu().numpy()
    ends = done[T - 1]
    if autoreset and ends.any():
        nxt = np.searchsorted(d["ep_offsets"], idx[T - 1], side="right") % E    # episode after the one row idx belongs to
        assert np.array_equal(state[ends].view(np.uint32), d["ep_init_state"][nxt[ends]].view(np.uint32))
    assert rel_err(state[~ends & live[T - 1]], want_obs[T - 1][~ends & live[T - 1]]).max(initial=0) <= RTOL
    # episode bookkeeping of the finished episodes (utils.py:99-125 as tallied on the device)
    tally = env.tally.cpu().numpy()[:, :B]
    n_done = (done & live).sum(0)
    assert np.array_equal(tally[L.T_EPISODES], n_done)
    viol_done = np.zeros(B); len_done = np.zeros(B); ret_done = np.zeros(B)
    for l in range(min(B, E)):
        e = first[l]
        for _ in range(int(n_done[l])):
            viol_done[l] += d["ep_viol"][e]; len_done[l] += d["ep_length"][e]; ret_done[l] += d["ep_return"][e]
            e = (e + 1) % E
    for l0 in range(E, B, E):
        w = min(E, B - l0)
        viol_done[l0:l0 + w] = viol_done[:w]; len_done[l0:l0 + w] = len_done[:w]; ret_done[l0:l0 + w] = ret_done[:w]
    assert np.array_equal(tally[L.T_VIOL], viol_done) and np.array_equal(tally[L.T_LEN_SUM], len_done)
    assert rel_err(tally[L.T_RET_SUM], ret_done, floor=1e-3).max() <= RTOL
    # base.py:183 total_violations: every violation so far, the running episode's included
    assert np.array_equal(env.total_violations.cpu().numpy(), (d["viol"][idx] * live).sum(0))
    env.close()


def test_noise_rollout_equals_the_parity_step_kernel(ni):
    """The injected-draw rollout is nig_step's parity mode fused: same row sets, bit-identical state / reward / flags."""
    key, B, T = "pg", 512, 24
    d = load_golden(key, "g3")
    act, nz, rz, idx, first = _chain(d, B, T)
    outs = []
    for fused in (True, False):
        ni.tune(split_blocks=0, wide_min_blocks=1)
        env = ni.make_batched(ENV_NAME[key], B, autoreset=True)
        env.reset(init_noise=d["ep_init_noise"][first].T)
        rew = torch.zeros(T, env.ld, dtype=torch.float32, device="cuda")
        fl = torch.zeros(T, env.ld, dtype=torch.int32, device="cuda")
        obs = torch.zeros(T, B, 32, dtype=torch.float32, device="cuda")
        if fused:
            env.rollout_noise(T, torch.from_numpy(act).cuda(), torch.from_numpy(nz).cuda(), torch.from_numpy(rz).cuda(), rew, fl, obs)
        else:
            for t in range(T):
                o, r, te, tr, info = env.step(act[t].T.copy(), step_noise=nz[t], reset_noise=rz[t], layout="aos")
                rew[t, :B] = r; fl[t, :B] = info.flags
        torch.cuda.synchronize()
        outs.append((env.get_state().cpu().numpy(), rew.cpu().numpy(), fl.cpu().numpy()))
        env.close()
    assert np.array_equal(outs[0][0].view(np.uint32), outs[1][0].view(np.uint32))
    assert np.array_equal(outs[0][1].view(np.uint32), outs[1][1].view(np.uint32))
    assert np.array_equal(outs[0][2], outs[1][2])


def test_noise_rollout_argument_checks(ni):
    env = ni.make_batched("HVACControl-v0", 256, autoreset=True)
    T, ld = 4, env.ld
    ring = torch.zeros(T, env.action_dim, ld, device="cuda")
    nz = torch.zeros(T, 2, ld, dtype=torch.float64, device="cuda")
    rew = torch.zeros(T, ld, device="cuda"); fl = torch.zeros(T, ld, dtype=torch.int32, device="cuda")
    obs = torch.zeros(T, 256, env.state_dim, device="cuda")
    with pytest.raises(Exception, match="only the envs the reference can record draws for"):
        env.rollout_noise(T, ring, nz, nz, rew, fl, obs)
    env.close()
    env = ni.make_batched("ChemicalReactor-v0", 256, autoreset=True)
    ring = torch.zeros(T, 3, env.ld, device="cuda")
    obs = torch.zeros(T, 256, 12, device="cuda")
    with pytest.raises(Exception, match="needs reset_noise"):
        env.rollout_noise(T, ring, nz, None, rew, fl, obs)
    env.close()
