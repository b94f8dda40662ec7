"""-m gpu: the three-wave form of the ChemicalReactor rollout (csrc/nig_split.hpp: producer / integrator /
recorder wave per 64 lanes, used for batches of up to one 256-lane block per CU) against the one-wave
rollout_kernel on the same inputs.  Everything a rollout leaves behind must be bit-identical: per-step reward /
flag rows, both trajectory layouts, final state, counter words, lifetime violations, running returns, tallies."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

NAME = "ChemicalReactor-v0"


@pytest.fixture(scope="module")
def ni():
    import neorl_industrial_gym_amd as ni
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    yield ni
    ni.tune(split_blocks=256)


def _ring(env, R, t0=70):
    ring = torch.empty(R, env.action_dim, env.ld, dtype=torch.float32, device=env.device)
    for s in range(R):
        env.fill_actions(t0 + s, ring[s])
    return ring


def _run(ni, split, B, chunks, outputs, R, max_steps, first_counter=0, seed=11, tally=True, env_index0=0, cmask=None, name=NAME):
    """Roll `chunks` (list of step counts) through one handle; returns every observable as CPU tensors."""
    ni.tune(split_blocks=256 if split else 0)
    env = ni.make_batched(name, B, seed=seed, autoreset=True, tally=tally, max_episode_steps=max_steps, env_index0=env_index0)
    if cmask is not None:
        env.set_constraint_mask(cmask)
    ring = _ring(env, R)
    env.reset()
    env.counter = first_counter
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
    got += [env.state_soa.cpu(), env.ctr.cpu(), env.life_viol.cpu()]
    if tally:
        got += [env.ep_return.cpu(), env.tally.cpu()]
    counter = env.counter
    env.close()
    return got, counter


def _same(a, b):
    assert len(a) == len(b)
    for i, (x, y) in enumerate(zip(a, b)):
        assert x.shape == y.shape and x.dtype == y.dtype, i
        xv = x.contiguous().view(torch.int32) if x.dtype == torch.float32 else (x.contiguous().view(torch.int64) if x.dtype == torch.float64 else x)
        yv = y.contiguous().view(torch.int32) if y.dtype == torch.float32 else (y.contiguous().view(torch.int64) if y.dtype == torch.float64 else y)
        assert torch.equal(xv, yv), f"observable {i} differs"


@pytest.mark.parametrize("outputs", ["none", "last", "rows", "soa", "aos"])
@pytest.mark.parametrize("B", [1024, 2500])
def test_split_form_equals_one_wave_form(ni, outputs, B):
    """Whole blocks (1024) and whole blocks + a ragged tail launch (2500); 41 steps over a 7-slot ring with
    20-step episodes: ring wrap, truncations, terminations and in-kernel resets all occur."""
    kw = dict(B=B, chunks=[41], outputs=outputs, R=7, max_steps=20)
    a, ca = _run(ni, True, **kw)
    b, cb = _run(ni, False, **kw)
    assert ca == cb == 41
    _same(a, b)
    assert float(a[-1][0].sum()) > 0          # episodes finished and were tallied


@pytest.mark.parametrize("n_steps", [1, 2, 3, 4, 5, 6, 7, 9, 13])
def test_split_form_short_and_odd_lengths(ni, n_steps):
    """Launches shorter than the producer's look-ahead and every remainder of its unrolled loop."""
    kw = dict(B=512, chunks=[n_steps], outputs="aos", R=3, max_steps=4)
    a, _ = _run(ni, True, **kw)
    b, _ = _run(ni, False, **kw)
    _same(a, b)


@pytest.mark.parametrize("first_counter", [0, 1, 6, 7])
def test_split_form_chunked_and_misaligned_starts(ni, first_counter):
    """Consecutive launches on one handle (state, counters and running returns carried over) starting on odd
    and even launch counters: an even start peels one step into the one-wave form first."""
    kw = dict(B=768, chunks=[5, 1, 8, 2, 11], outputs="rows", R=5, max_steps=9, first_counter=first_counter)
    a, ca = _run(ni, True, **kw)
    b, cb = _run(ni, False, **kw)
    assert ca == cb == first_counter + 27
    _same(a, b)


@pytest.mark.parametrize("kw", [dict(tally=False), dict(env_index0=3 * 65536 + 512, seed=0xABCDEF), dict(cmask=0b101),
                                dict(cmask=0)], ids=["no_tally", "shard_offset", "mask_101", "mask_none"])
def test_split_form_handle_variants(ni, kw):
    """Handles without the episode tally, lanes keyed at a shard offset, handles with constraints removed
    (base.py:224-228): the three-wave form reads the same handle state as the one-wave form."""
    args = dict(B=1024, chunks=[23, 10], outputs="aos", R=6, max_steps=12, **kw)
    a, _ = _run(ni, True, **args)
    b, _ = _run(ni, False, **args)
    _same(a, b)


RA = "RobotAssembly-v0"


@pytest.mark.parametrize("outputs", ["none", "last", "rows", "soa", "aos"])
@pytest.mark.parametrize("B", [1024, 2500])
def test_split_form_robot_assembly_equals_one_wave_form(ni, outputs, B):
    """RobotAssembly (S = 24, A = 7, no step noise: the producer only loads and clips actions; three ring slots) in
    the three-wave form against rollout_kernel: 41 steps over a 7-slot ring, 20-step episodes plus the env's own
    terminations (workspace / velocity limits), in-kernel cooperative resets in most steps."""
    kw = dict(B=B, chunks=[41], outputs=outputs, R=7, max_steps=20, name=RA)
    a, ca = _run(ni, True, **kw)
    b, cb = _run(ni, False, **kw)
    assert ca == cb == 41
    _same(a, b)
    assert float(a[-1][0].sum()) > 0


@pytest.mark.parametrize("first_counter", [0, 1])
def test_split_form_robot_assembly_chunks_lengths_and_variants(ni, first_counter):
    """Consecutive launches of every short length (the producer's look-ahead tail cases) on one handle, from even and
    odd launch counters (no pairing for an env without step noise: both start in the three-wave form), then handles
    without the tally, at a shard offset and with constraints masked."""
    kw = dict(B=768, chunks=[5, 1, 8, 2, 3, 4, 6, 7, 11], outputs="rows", R=5, max_steps=9, first_counter=first_counter, name=RA)
    a, ca = _run(ni, True, **kw)
    b, cb = _run(ni, False, **kw)
    assert ca == cb == first_counter + 47
    _same(a, b)
    for v in (dict(tally=False), dict(env_index0=3 * 65536 + 512, seed=0xABCDEF), dict(cmask=0b101), dict(cmask=0)):
        args = dict(B=512, chunks=[23, 10], outputs="aos", R=6, max_steps=12, name=RA, **v)
        a, _ = _run(ni, True, **args)
        b, _ = _run(ni, False, **args)
        _same(a, b)


def test_split_form_robot_assembly_long_run_at_full_size(ni):
    """One block per compute unit (65 536 lanes, the size the form is for), 2 x 2 500 steps with the env's own episode
    length: the three-slot rings wrap ~1 700 times under real contention; final state, counters, returns and tallies equal
    the one-wave kernel's bit for bit (a protocol slip would show as a mismatch or as a hang of this call)."""
    kw = dict(B=65536, chunks=[2500, 2500], outputs="none", R=64, max_steps=1000, name=RA)
    a, ca = _run(ni, True, **kw)
    b, cb = _run(ni, False, **kw)
    assert ca == cb == 5000
    _same(a, b)
    assert float(a[-1][0].sum()) > 1e6         # millions of episodes finished and were tallied


@pytest.mark.parametrize("name", [NAME, "PowerGrid-v0", RA])
@pytest.mark.parametrize("B", [1024, 1000])
def test_step_api_helper_waves_equal_the_plain_step_kernel(ni, name, B):
    """The step API at one wave per SIMD launches step_kernel with HELPER waves (restart states, the generator's table and,
    for PowerGrid, the tally flush prepared beside the step: csrc/nig_kernels.hpp step_kernel, HELP); the knob at 0 keeps
    the plain kernel with its cooperative reset.  60 steps of 9-step episodes through both, whole blocks and a ragged
    last block: every per-step reward / flag row, the final state, counters, lifetime violations, returns and tallies
    are bit-identical."""
    def run(knob):
        ni.tune(split_blocks=knob)
        env = ni.make_batched(name, B, seed=21, autoreset=True, tally=True, max_episode_steps=9)
        env.reset()
        act = torch.empty(env.action_dim, env.ld, dtype=torch.float32, device=env.device)
        got = []
        for t in range(1, 61):
            env.fill_actions(500 + t, act)
            env.step(act[:, :B], layout="soa")
            got += [env.reward[:B].cpu().clone(), env.flags[:B].cpu().clone()]
        got += [env.state_soa[:, :B].cpu(), env.ctr[:B].cpu(), env.life_viol[:B].cpu(), env.ep_return[:B].cpu(), env.tally[:, :B].cpu()]
        env.close()
        return got
    a, b = run(256), run(0)
    _same(a, b)
    assert float(a[-1][0].sum()) > B               # every lane finished several episodes


def _run_policy(ni, split, policy, B, chunks, stream, max_steps, seed=5, name=NAME):
    """Closed-loop rollouts (nig_rollout_policy) through one handle; returns every observable as CPU tensors."""
    if split is not None:                      # None: keep the knob the caller set
        ni.tune(split_blocks=256 if split else 0)
    env = ni.make_batched(name, B, seed=seed, autoreset=True, tally=True, max_episode_steps=max_steps)
    env.set_policy(policy)
    env.reset()
    got = []
    for T in chunks:
        rew = fl = obs = act = None
        if stream != "none":
            rew = torch.full((T, env.ld), float("nan"), dtype=torch.float32, device=env.device)
            fl = torch.zeros(T, env.ld, dtype=torch.int32, device=env.device)
        if stream == "transitions":
            obs = torch.full((T, B, env.state_dim), float("nan"), dtype=torch.float32, device=env.device)
            act = torch.full((T, env.action_dim, env.ld), float("nan"), dtype=torch.float32, device=env.device)
        env.rollout_policy(T, rew, fl, obs, act)
        torch.cuda.synchronize()
        got += [t.cpu() if t is obs else t[..., :B].cpu() for t in (rew, fl, obs, act) if t is not None]
    got += [env.state_soa.cpu(), env.ctr.cpu(), env.life_viol.cpu(), env.ep_return.cpu(), env.tally.cpu()]
    env.close()
    return got


@pytest.mark.parametrize("stream", ["none", "rows", "transitions"])
@pytest.mark.parametrize("which", ["expert", "medium", "mixed", "random", "pid", "mpc", "constant", "uniform"])
def test_split_policy_form_equals_one_wave_form(ni, which, stream):
    """nig_rollout_policy in the three-wave form (csrc/nig_split_policy.hpp) against rollout_policy_kernel: every
    kind of on-device policy (feedback law with exploration normals, epsilon-mix with a uniform action, PID with
    memory carried across launches, constant, uniform random), with and without the transition stream."""
    S, A = 12, 3
    if which in ("expert", "medium", "mixed", "random"):
        policy = ni.behaviour_policy(NAME, which)
    else:
        policy = {"pid": lambda: ni.pid_agent(S, A), "mpc": lambda: ni.mpc_agent(S, A), "constant": lambda: ni.constant_agent(S, A),
                  "uniform": lambda: ni.random_agent(S, A)}[which]()
    kw = dict(policy=policy, B=1024, chunks=[9, 1, 14], stream=stream, max_steps=11)
    a = _run_policy(ni, True, **kw)
    b = _run_policy(ni, False, **kw)
    _same(a, b)


@pytest.mark.parametrize("B,split_blocks", [(1024 + 100, 256), (256 + 1, 256), (4 * 256, 2), (5 * 256, 2), (8 * 256 + 77, 4), (7 * 256, 4)])
def test_split_policy_form_with_a_ragged_tail_and_in_rounds(ni, B, split_blocks):
    """Round 3: the closed loop's whole 256-lane blocks run the three-wave form and a ragged last block the one-wave
    kernel (one call, two launches); batches of several rounds (here: rounds of 2 or 4 blocks) run it in rounds when the
    last round is at least 3/4 full ((4 blocks, 2 per round) and (8 + tail, 4): even; (7, 4): yes) -- since round 5 at most TWO
    rounds and only with the observation stream (three rounds, (5, 2), stay on the one-wave kernel: measured slower in rounds,
    profiles/r05/policy_rounds_cr.txt) -- all bit-identical to the one-wave kernel, PID memory and transition stream included."""
    import bench
    ni.tune(split_blocks=split_blocks)
    expect_split = B // 256 <= 2 * split_blocks
    assert bench.policy_kernel_name(ni, "cr", B).startswith("split_policy_kernel") == expect_split
    for policy in (ni.behaviour_policy(NAME, "medium"), ni.pid_agent(12, 3)):
        kw = dict(policy=policy, B=B, chunks=[7, 6], stream="transitions", max_steps=9)
        ni.tune(split_blocks=split_blocks)
        a = _run_policy(ni, None, **kw)
        b = _run_policy(ni, False, **kw)
        _same(a, b)


@pytest.mark.parametrize("stream", ["none", "rows", "transitions"])
@pytest.mark.parametrize("which", ["expert", "medium", "random", "mpc", "constant", "uniform", "pid"])
def test_powergrid_paired_closed_loop_equals_one_wave_form(ni, which, stream):
    """Round 4: nig_rollout_policy for PowerGrid batches of at most one 256-lane block per compute unit runs the PAIRED
    form (csrc/nig_pg_lds.hpp rollout_pg_pair_policy_kernel: the producer wave also draws the policy's own random numbers,
    the stepping wave evaluates the feedback law on its LDS image) -- every affine policy kind (get_dataset's behaviour
    laws power_grid.py:216-233 incl. the epsilon-mix with a uniform action, "MPC", constant, uniform random), with and
    without the transition stream, whole blocks + a ragged last block, three launches chained -- bit-identical to
    rollout_policy_kernel in rewards, flags, observations acted on, actions, final state, counters, returns, tallies.
    A PID policy keeps the one-wave kernel (its memory lives in registers): same call, same results.
    Round 5: with the observation stream ("transitions") the call STAYS on the paired form's register-resident stepper
    (pg_policy_reg_body writes the rows through the wave's reset image); round 4 switched to the LDS-resident stepper for it."""
    import bench
    S, A = 32, 8
    name = "PowerGrid-v0"
    ni.tune(split_blocks=256)
    assert bench.policy_kernel_name(ni, "pg", 1024, "pid" if which == "pid" else "affine") == (
        "rollout_policy_kernel<PowerGrid>" if which == "pid" else "rollout_pg_pair_policy_kernel<PolicyArgs> (pg_policy_reg_body)")
    if which in ("expert", "medium", "random"):
        policy = ni.behaviour_policy(name, which)
    else:
        policy = {"pid": lambda: ni.pid_agent(S, A), "mpc": lambda: ni.mpc_agent(S, A), "constant": lambda: ni.constant_agent(S, A),
                  "uniform": lambda: ni.random_agent(S, A)}[which]()
    for B in (1024, 512 + 77):
        kw = dict(policy=policy, B=B, chunks=[9, 1, 14], stream=stream, max_steps=11, name=name)
        a = _run_policy(ni, True, **kw)
        b = _run_policy(ni, False, **kw)
        _same(a, b)


@pytest.mark.parametrize("stream", ["none", "rows", "transitions"])
@pytest.mark.parametrize("which", ["expert", "medium", "random", "mpc", "constant", "uniform", "pid"])
def test_robot_assembly_three_wave_closed_loop_equals_one_wave_form(ni, which, stream):
    """Round 4: nig_rollout_policy for RobotAssembly batches of at most one 256-lane block per compute unit runs the three-wave
    closed-loop form in its BIG layout (csrc/nig_split_policy.hpp: two ring slots, no observation rows in the I -> C slot, the
    feedback matrix prefetched from a dense LDS copy) -- get_dataset's behaviour laws (robot_assembly.py:266-290: feedback +
    uniform perturbations + epsilon-mix, clip to +-2), "MPC", constant, uniform random and PID (memory across launches) --
    bit-identical to rollout_policy_kernel in rewards, flags, actions, final state, counters, returns, tallies, PID memory.
    Round 5: the transition stream's OBSERVATIONS too -- they ride in the producer -> integrator slot once the integrator has
    read the step's draws out of it (round 4 sent such calls to the one-wave kernel)."""
    import bench
    S, A = 24, 7
    ni.tune(split_blocks=256)
    assert bench.policy_kernel_name(ni, "ra", 1024) == "split_policy_kernel<RobotAssembly,4>"       # for every stream mode
    if which in ("expert", "medium", "random"):
        policy = ni.behaviour_policy(RA, which)
    else:
        policy = {"pid": lambda: ni.pid_agent(S, A), "mpc": lambda: ni.mpc_agent(S, A), "constant": lambda: ni.constant_agent(S, A),
                  "uniform": lambda: ni.random_agent(S, A)}[which]()
    for B in (1024, 256 + 50):
        kw = dict(policy=policy, B=B, chunks=[9, 1, 14], stream=stream, max_steps=11, name=RA)
        a = _run_policy(ni, True, **kw)
        b = _run_policy(ni, False, **kw)
        _same(a, b)


@pytest.mark.parametrize("max_steps", [1, 2, 3])
def test_split_form_resets_every_step(ni, max_steps):
    """Episodes of one to three steps: every lane (or half / a third of them) is renewed by the cooperative reset
    in every step, the integrator's heaviest path, while producer and recorder keep running ahead / behind."""
    kw = dict(B=2048, chunks=[40], outputs="aos", R=4, max_steps=max_steps)
    a, _ = _run(ni, True, **kw)
    b, _ = _run(ni, False, **kw)
    _same(a, b)


def test_split_form_at_the_headline_batch(ni):
    """65 536 lanes x 250 steps, default 500-step episodes, row-major trajectory: the benchmark's launch."""
    kw = dict(B=65536, chunks=[250], outputs="aos", R=16, max_steps=None)
    a, _ = _run(ni, True, **kw)
    b, _ = _run(ni, False, **kw)
    _same(a, b)


def test_split_form_in_two_rounds(ni):
    """131 072 lanes = two rounds of one block per CU (the form is used for batches whose rounds come out even)."""
    kw = dict(B=131072, chunks=[31], outputs="aos", R=8, max_steps=20)
    a, _ = _run(ni, True, **kw)
    b, _ = _run(ni, False, **kw)
    _same(a, b)


def test_tune_knob_roundtrip(ni):
    assert ni.tune(split_blocks=17)["split_blocks"] == 17
    assert ni.tune()["split_blocks"] == 17
    with pytest.raises(Exception):
        ni.tune(split_blocks=-2)
    assert ni.tune(split_blocks=256)["split_blocks"] == 256
