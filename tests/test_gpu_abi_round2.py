"""-m gpu: edge cases of the C ABI pinned through ctypes (round-2 fixes): frozen lanes of an
auto-reset handle in the fused rollout, SafetyMetrics for handles whose constraint count is not 3,
the two satisfaction-rate tally rows, PID controller memory across launches."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import ENV_NAME, KEYS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ni():
    import neorl_industrial_gym_amd as ni
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return ni


def _ring(env, R, t0=70):
    ring = torch.empty(R, env.action_dim, env.ld, dtype=torch.float32, device=env.device)
    for s in range(R):
        env.fill_actions(t0 + s, ring[s])
    return ring


@pytest.mark.parametrize("key", KEYS)
@pytest.mark.parametrize("how", ["never_reset", "masked_reset", "set_state_done"])
def test_rollout_freezes_done_lanes_of_an_autoreset_handle(ni, key, how):
    """base.py:159-160: a lane that is done waits for reset() -- also on a handle that auto-resets
    (lanes never reset, left out by reset(mask), or marked done by set_state).  Full 256-lane
    blocks (B % 256 == 0): the fused rollout must leave those lanes exactly as n step launches do."""
    B, T, R = 1024, 9, 4
    a = ni.make_batched(ENV_NAME[key], B, autoreset=True, tally=True, max_episode_steps=6)
    b = ni.make_batched(ENV_NAME[key], B, autoreset=True, tally=True, max_episode_steps=6)
    ring = _ring(a, R)
    mask = (torch.arange(B, device=a.device) % 3 != 0).to(torch.uint8)
    for e in (a, b):
        if how == "never_reset":
            e.reset(mask=mask)                       # lanes with mask == 0 were never reset: DONE since nig_create
        elif how == "masked_reset":
            e.reset()
            e.step(ring[0][:, :B], layout="soa")
            e.ctr[::5] |= ni._lib.CTR_DONE           # out-of-band: what a caller's own bookkeeping may do
            e.set_state(current_step=e.current_step, violation_count=e.violation_count, done=e.done)
            e.reset(mask=mask)
        else:
            e.reset()
            done = torch.zeros(B, dtype=torch.bool, device=e.device)
            done[100:400] = True
            e.set_state(current_step=e.current_step, violation_count=e.violation_count, done=done)
    frozen_before = b.done.clone()
    assert bool(frozen_before.any())
    state_before = b.state_soa.clone()
    rew = torch.zeros(T, a.ld, dtype=torch.float32, device=a.device)
    fl = torch.zeros(T, a.ld, dtype=torch.int32, device=a.device)
    a.rollout(T, ring, rew, fl)
    for k in range(T):
        _, r, _, _, info = b.step(ring[k % R][:, :B], layout="soa")
        assert torch.equal(rew[k, :B], r), k
        assert torch.equal(fl[k, :B], info.flags), k
    torch.cuda.synchronize()
    assert torch.equal(a.state_soa.view(torch.int32), b.state_soa.view(torch.int32))
    assert torch.equal(a.ctr, b.ctr) and torch.equal(a.life_viol, b.life_viol)
    # frozen lanes: untouched, still done, flagged inactive in every step
    assert torch.equal(a.done, frozen_before)
    assert torch.equal(a.state_soa[:, frozen_before].view(torch.int32), state_before[:, frozen_before].view(torch.int32))
    assert bool(((fl[:, :B][:, frozen_before] & ni._lib.FLAG_INACTIVE) != 0).all())
    assert not bool(torch.isnan(a.state_soa).any())
    # a full reset clears the condition: the fast path and the step kernel agree again
    a.reset(); b.reset()
    a.rollout(T, ring, rew, fl)
    for k in range(T):
        b.step(ring[k % R][:, :B], layout="soa")
    assert torch.equal(a.state_soa.view(torch.int32), b.state_soa.view(torch.int32)) and torch.equal(a.ctr, b.ctr)
    assert not bool(a.done.any())
    a.close(); b.close()


def _metrics_through_ctypes(ni, env):
    """nig_get_safety_metrics called the way a C host would: raw pointers, own output array."""
    L = ni._lib.lib()
    out = torch.full((5, env.ld), -99, dtype=torch.int32, device=env.device)
    rc = L.nig_get_safety_metrics(env._h, C.c_void_p(env.flags.data_ptr()), C.c_void_p(out.data_ptr()), env.ld,
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, L.nig_last_error()
    torch.cuda.synchronize()
    return out[:, :env.batch].cpu().numpy()


def test_safety_metrics_with_four_conditions(ni):
    """AdvancedChemicalReactor carries 4 safety-metric conditions: total_constraints = 4 and a count of
    4 (NIG_FLAG_NVIOL_HI) must come out of nig_get_safety_metrics."""
    B = 512
    env = ni.make_batched("AdvancedChemicalReactor-v0", B, autoreset=False)
    env.reset()
    st = env.get_state().clone()
    st[: B // 2, 0] = 700.0             # T above the limit -> new T stays above: conditions 0 and 2
    st[: B // 4, 2] = 6.0e6
    env.set_state(st)
    act = torch.zeros(env.action_dim, B, device=env.device)
    act[3] = 700.0                      # feed temperature setpoint keeps the reactor hot
    _, _, _, _, info = env.step(act, layout="soa")
    m = _metrics_through_ctypes(ni, env)
    nv = info.violation_count.cpu().numpy()
    assert (m[1] == 4).all()
    assert np.array_equal(m[2], nv) and np.array_equal(m[0], 4 - nv) and np.array_equal(m[4], 4 - nv)
    assert nv.max() >= 2 and nv.min() == 0
    bits = info.constraint_violated.cpu().numpy()
    assert bits.shape[0] == 4 and np.array_equal(bits.sum(0), nv)
    assert np.array_equal(m, env.get_safety_metrics().cpu().numpy())
    env.close()


@pytest.mark.parametrize("key", KEYS)
def test_safety_metrics_follow_the_constraint_mask(ni, key):
    """remove_safety_constraint (base.py:224-228) on a batched handle: total_constraints drops,
    satisfied = enabled - violated, the removed constraint is not counted."""
    B = 2048
    full = ni.make_batched(ENV_NAME[key], B, autoreset=False)
    part = ni.make_batched(ENV_NAME[key], B, autoreset=False)
    part.set_constraint_mask(0b101)
    ring = _ring(full, 1)
    for e in (full, part):
        e.reset()
        st = e.get_state().clone()
        if key == "cr":
            st[::2, 0] = 360.0; st[::3, 1] = 6.0e5; st[::5, 10] = 95.0
        elif key == "pg":
            st[::2, 0] = 0.7; st[::3, 1] = 1.2; st[::5, 9] = 100.0
        else:
            st[::2, 18] = 60.0; st[::3, 0] = 0.55; st[::5, 7] = 2.5
        e.set_state(st)
        e.step(ring[0][:, :B], layout="soa")
    mf, mp = _metrics_through_ctypes(ni, full), _metrics_through_ctypes(ni, part)
    vb = StepBits(ni, full.flags)
    assert (mf[1] == 3).all() and (mp[1] == 2).all()
    assert np.array_equal(mf[2], vb.sum(0)) and np.array_equal(mp[2], vb[[0, 2]].sum(0))
    assert np.array_equal(mp[0], 2 - mp[2]) and np.array_equal(mf[0], 3 - mf[2])
    assert vb[1].any() and mp[2].max() == 2
    full.close(); part.close()


def StepBits(ni, flags):
    f = flags.cpu().numpy()
    return np.stack([(f >> (ni._lib.FLAG_VIOL_SHIFT + k)) & 1 for k in range(3)])


@pytest.mark.parametrize("env_id,n_total,mask", [("ChemicalReactor-v0", 3, None), ("PowerGrid-v0", 2, 0b110),
                                                 ("AdvancedChemicalReactor-v0", 4, None), ("RobotAssembly-v0", 0, 0)])
def test_tally_satisfaction_rows(ni, env_id, n_total, mask):
    """NIG_T_SATISFIED / NIG_T_CONSTRAINTS = sums over the steps of FINISHED episodes of
    constraints_satisfied / total_constraints (utils.py:109), from the step kernel and from the
    fused rollout; evaluate's constraint_satisfaction_rate follows from them."""
    from neorl_industrial_gym_amd.parallel import metrics_from_partial
    B, T, R = 1536, 40, 8
    L = ni._lib
    a = ni.make_batched(env_id, B, autoreset=True, tally=True, max_episode_steps=12)
    b = ni.make_batched(env_id, B, autoreset=True, tally=True, max_episode_steps=12)
    if mask is not None:
        a.set_constraint_mask(mask); b.set_constraint_mask(mask)
    ring = _ring(a, R)
    if env_id.startswith("Advanced"):
        ring.mul_(0.001)                 # physical actuator ranges of that env
    fl = torch.zeros(T, a.ld, dtype=torch.int32, device=a.device)
    rew = torch.zeros(T, a.ld, dtype=torch.float32, device=a.device)
    a.reset(); b.reset()
    a.rollout(T, ring, rew, fl)
    for k in range(T):
        b.step(ring[k % R][:, :B], layout="soa")
    assert torch.equal(a.tally[[L.T_SATISFIED, L.T_CONSTRAINTS]], b.tally[[L.T_SATISFIED, L.T_CONSTRAINTS]])
    t = a.tally.cpu().numpy()
    assert t[L.T_EPISODES].sum() > 0
    assert np.array_equal(t[L.T_CONSTRAINTS], n_total * t[L.T_LEN_SUM])
    assert np.array_equal(t[L.T_SATISFIED], n_total * t[L.T_LEN_SUM] - t[L.T_VIOL])
    got = metrics_from_partial(a.reduce_tally())
    steps, viol = t[L.T_LEN_SUM].sum(), t[L.T_VIOL].sum()
    want = 1.0 if n_total == 0 else (n_total * steps - viol) / (n_total * steps)
    assert got["constraint_satisfaction_rate"] == pytest.approx(want, rel=1e-12)
    a.close(); b.close()


def test_pid_memory_persists_across_launches(ni, oracle):
    """PIDControllerAgent keeps its integral and previous error for life (baseline_agents.py:55-80):
    a closed-loop rollout cut into launches of odd lengths == one launch == the oracle, and a
    second installation of the policy starts from zero again."""
    B, T = 1000, 60
    pol = ni.pid_agent(12, 3, kp=0.02, ki=0.001, kd=0.01, setpoint=[320.0, 253312.5, 50.0])

    def run(cuts):
        env = ni.make_batched("ChemicalReactor-v0", B, autoreset=True, tally=True, seed=11)
        env.set_policy(pol)
        env.reset()
        acts = torch.zeros(T, 3, env.ld, dtype=torch.float32, device=env.device)
        k = 0
        for n in cuts:
            env.rollout_policy(n, act_out=acts[k:k + n])
            k += n
        assert k == T
        torch.cuda.synchronize()
        return env, acts[:, :, :B].clone()

    one, a1 = run([T])
    many, a2 = run([7, 1, 20, 13, 19])
    assert torch.equal(a1.view(torch.int32), a2.view(torch.int32))
    assert torch.equal(one.state_soa.view(torch.int32), many.state_soa.view(torch.int32))
    assert torch.equal(one.ctr, many.ctr) and torch.equal(one.tally, many.tally)
    assert float(a1[-1].abs().max()) > 0 and not torch.equal(a1[5], a1[6])
    P = oracle.Policy.from_buffer_copy(bytes(pol.to_struct()))
    r = oracle.rollout_policy("cr", B, T, P, seed=11, autoreset=True, trajectories=True)
    assert np.array_equal(one.get_state().cpu().numpy().view(np.uint32), r["state"].view(np.uint32))
    assert np.array_equal(a1.permute(0, 2, 1).cpu().numpy().view(np.uint32), r["act"].view(np.uint32))
    # re-installing the policy is the agent's constructor: memory back to zero
    many.set_policy(pol)
    many.counter = 0
    many.reset()
    acts = torch.zeros(T, 3, many.ld, dtype=torch.float32, device=many.device)
    many.rollout_policy(T, act_out=acts)
    assert torch.equal(acts[:, :, :B].view(torch.int32), a1.view(torch.int32))
    one.close(); many.close()


def test_evaluate_with_safety_on_four_condition_env(ni):
    """Batched evaluate_with_safety on AdvancedChemicalReactor (4 conditions): the satisfaction rate
    is taken over 4 constraints per step, and equals the single-env host loop's."""
    B = 64
    agent = ni.constant_agent(20, 6, [0.002, 0.004, 800.0, 330.0, 20.0, 0.0])
    benv = ni.make_batched("AdvancedChemicalReactor-v0", B, autoreset=False, tally=True, max_episode_steps=30)
    got = ni.evaluate_with_safety(agent, benv, n_episodes=B)
    env = ni.make("AdvancedChemicalReactor-v0", max_episode_steps=30)
    want = ni.evaluate_with_safety(agent, env, n_episodes=2)        # deterministic env: every episode is the same
    for k in ("return_mean", "length_mean", "constraint_satisfaction_rate", "safety_violations_per_episode"):
        assert got[k] == pytest.approx(want[k], rel=1e-6), k
    assert 0.0 < got["constraint_satisfaction_rate"] <= 1.0
    benv.close()


def test_reduce_metrics_through_rccl(ni):
    """nig_reduce_metrics: the path's one collective as a C host would call it -- the tallies of several handles
    (here the segments of a mixed batch) reduced, all-gathered over an ncclComm_t (RCCL; a single-rank
    communicator on this one-GPU box, created with ctypes on the RCCL copy torch already loaded) and combined in
    rank order.  Equals the host-side combine of the per-handle partial vectors; comm = NULL skips the collective."""
    import os
    from neorl_industrial_gym_amd.parallel import combine_partials, metrics_from_partial
    L, lib = ni._lib.lib(), ni._lib
    segs = [("PowerGrid-v0", 2000), ("ChemicalReactor-v0", 1500), ("RobotAssembly-v0", 1000)]
    mix = ni.MixedBatchedEnv(segs, seed=4, autoreset=True, tally=True)
    ring = torch.zeros(4, mix.A_max, mix.ld, dtype=torch.float32, device=mix.device)
    for s in range(4):
        mix.fill_actions(20 + s, ring[s])
    mix.reset()
    mix.rollout(80, ring)
    want = combine_partials(torch.stack([e.reduce_tally() for e in mix.envs])).cpu().numpy()
    assert want[lib.T_EPISODES] > 0

    rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0

    n = len(mix.envs)
    hs = (C.c_void_p * n)(*[e._h for e in mix.envs])
    scratch = torch.zeros((n + 2) * lib.T_ROWS, dtype=torch.float64, device=mix.device)
    out = torch.zeros(lib.T_ROWS, dtype=torch.float64, device=mix.device)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for c in (comm, None):
        out.zero_()
        lib.check(L.nig_reduce_metrics(hs, n, c, C.c_void_p(scratch.data_ptr()), scratch.numel(), C.c_void_p(out.data_ptr()), st))
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want)
    m = metrics_from_partial(out)
    assert len(m) == 13 and m["safety_violations"] == int(want[lib.T_VIOL])
    assert L.nig_reduce_metrics(hs, n, comm, C.c_void_p(scratch.data_ptr()), 5, C.c_void_p(out.data_ptr()), st) == 1   # scratch too small
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    rccl.ncclCommDestroy(comm)
    mix.close()


def test_workspace_kinds_nig_create_accepts_and_rejects(ni):
    """nig_create's caller workspace (include/nig.h): memory positively identified as host-pinned is refused (the step
    kernel's float64 hardware atomics execute at the memory side and would miscount there); a torch CUDA tensor is taken;
    a failing or unknown attribute query is NOT a refusal (device memory mapped through the virtual-memory API)."""
    L = ni._lib.lib()
    lay = ni._lib.layout_query(0, 1024, ni._lib.F_AUTORESET | ni._lib.F_TALLY)
    pinned = torch.empty(int(lay.bytes) + 256, dtype=torch.uint8).pin_memory()
    p = (pinned.data_ptr() + 255) // 256 * 256
    h = C.c_void_p()
    rc = L.nig_create(0, 1024, 0, C.c_uint64(1), C.c_uint64(0), 0, C.c_double(0.0), ni._lib.F_AUTORESET | ni._lib.F_TALLY, C.c_void_p(p), C.byref(h))
    assert rc != 0 and b"device-local" in L.nig_last_error()
    dev = torch.empty(int(lay.bytes), dtype=torch.uint8, device="cuda")
    rc = L.nig_create(0, 1024, 0, C.c_uint64(1), C.c_uint64(0), 0, C.c_double(0.0), ni._lib.F_AUTORESET | ni._lib.F_TALLY,
                      C.c_void_p(dev.data_ptr()), C.byref(h))
    assert rc == 0, L.nig_last_error()
    assert L.nig_destroy(h) == 0


EXPANDABLE_CHILD = r'''
import torch
import neorl_industrial_gym_amd as ni
env = ni.make_batched("ChemicalReactor-v0", 4096, autoreset=True, tally=True)      # workspace = a tensor of the expandable-segments allocator
env.reset()
ring = torch.empty(4, 3, env.ld, device="cuda")
for s in range(4):
    env.fill_actions(10 + s, ring[s])
env.rollout(600, ring)
part = env.reduce_tally()
assert int(part[ni._lib.T_EPISODES]) > 0
print("ok", torch.cuda.memory_stats().get("segment.all.current", -1))
'''


def test_workspace_from_the_expandable_segments_allocator_is_accepted():
    """The same handle on a workspace that torch mapped through hipMemCreate / hipMemMap (PYTORCH_CUDA_ALLOC_CONF=
    expandable_segments:True; a release that does not support the option ignores it with a warning): accepted and usable."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, PYTORCH_CUDA_ALLOC_CONF="expandable_segments:True", PYTORCH_HIP_ALLOC_CONF="expandable_segments:True",
               NIG_NO_AUTOBUILD="1")
    p = subprocess.run([sys.executable, "-c", EXPANDABLE_CHILD], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.startswith("ok"), (p.stdout[-1000:], p.stderr[-3000:])


def test_clock_stamp_gives_the_shader_clock_per_compute_unit(ni):
    """nig_clock_stamp (include/nig.h; bench.py's rank_times.clock): two stamps around a run of launches give, for every compute
    unit stamped both times, (s_memtime1 - s_memtime0) / (s_memrealtime1 - s_memrealtime0) x 100 MHz -- a plausible shader clock,
    nearly the same on every compute unit (the first, per-XCD version of the stamp read 2 030-5 570 "MHz": s_memtime is not one
    counter per chip)."""
    import bench
    probe = bench.ClockProbe(ni, torch, torch.device("cuda", 0))
    env = ni.make_batched("ChemicalReactor-v0", 65536, autoreset=True)
    env.reset()
    ring = _ring(env, 8)
    probe.stamp(0)
    for _ in range(40):
        env.rollout(250, ring)
    probe.stamp(1)
    torch.cuda.synchronize()
    c = probe.read()
    env.close()
    assert "error" not in c, c
    assert c["compute_units"] >= 64 and 500.0 < c["shader_clock_mhz"] < 3000.0, c
    assert c["shader_clock_mhz_p95"] / c["shader_clock_mhz_p05"] < 1.2, c          # one clock domain: a few percent across the chip
    assert 3.0 < c["span_ms"] < 100.0, c                                             # ~40 launches of ~0.15 ms
    assert ni._lib.lib().nig_clock_stamp(None, None) != 0                            # NULL buffer: an error code, not a fault
