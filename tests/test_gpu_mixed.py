"""-m gpu: BASELINE config 4 -- all seven envs in one padded SoA batch, the fused rollout as ONE kernel
launch (nig_create_mixed / nig_mixed_rollout / nig_rollout_mixed, through ctypes), at the full size of
1 048 576 lanes: every lane against the per-segment kernels, first and last wave of every
segment against the CPU oracle bit for bit, and the bookkeeping identities over the whole batch."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SEVEN = ["ChemicalReactor-v0", "RobotAssembly-v0", "HVACControl-v0", "WaterTreatment-v0", "SteelAnnealing-v0",
         "PowerGrid-v0", "SupplyChain-v0"]          # the README's seven (README.md:24-32)
SURVEY7 = ["ChemicalReactor-v0", "PowerGrid-v0", "RobotAssembly-v0", "AdvancedChemicalReactor-v0",
           "AdvancedPowerGrid-v0", "HVACControl-v0", "WaterTreatment-v0"]


@pytest.fixture(scope="module")
def ni():
    import neorl_industrial_gym_amd as ni
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return ni


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class CMixed:
    """nig_create_mixed and friends the way a C host binds them (raw pointers, own buffers)."""

    def __init__(self, ni, names, counts, seed=0x5EED, env_index0=0, flags=None):
        self.ni, self.L = ni, ni._lib.lib()
        lib = ni._lib
        n = len(names)
        ids = (C.c_int32 * n)(*[self.L.nig_env_id(x.encode()) for x in names])
        cnt = (C.c_int64 * n)(*counts)
        self.h = C.c_void_p()
        fl = (lib.F_AUTORESET | lib.F_TALLY) if flags is None else flags
        lib.check(self.L.nig_create_mixed(n, ids, cnt, 0, C.c_uint64(seed), C.c_uint64(env_index0), fl, C.byref(self.h)))
        self.info = lib.MixedInfo()
        lib.check(self.L.nig_mixed_get_info(self.h, C.byref(self.info)))
        self.ld, self.S, self.A = int(self.info.ld), int(self.info.state_dim_max), int(self.info.action_dim_max)
        self.names, self.counts = list(names), [int(c) for c in counts]
        self.offsets = [int(self.info.offset[k]) for k in range(n)]

    def state(self):
        """[S_max, ld] float32 copy of the library-owned padded observation matrix (hipMemcpy of the raw pointer)."""
        import os
        hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        out = torch.empty(self.S, self.ld, dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        assert hip.hipMemcpy(out.data_ptr(), self.L.nig_mixed_state(self.h), self.S * self.ld * 4, 3) == 0   # device -> device
        return out

    def ring(self, R, t0=1000):
        ring = torch.zeros(R, self.A, self.ld, dtype=torch.float32, device="cuda:0")
        for s in range(R):
            self.ni._lib.check(self.L.nig_mixed_fill_actions(self.h, t0 + s, C.c_void_p(ring[s].data_ptr()), _stream()))
        return ring

    def reset(self):
        self.ni._lib.check(self.L.nig_mixed_reset(self.h, _stream()))

    def rollout(self, T, ring, rew=None, fl=None, obs=None):
        if obs is not None:                    # + the observation rows of every step, padded [T][S_max][ld]
            self.ni._lib.check(self.L.nig_mixed_rollout_obs(
                self.h, T, C.c_void_p(ring.data_ptr()), ring.stride(0), ring.shape[0], C.c_void_p(rew.data_ptr()),
                C.c_void_p(fl.data_ptr()), rew.stride(0), C.c_void_p(obs.data_ptr()), obs.stride(0), _stream()))
            return
        self.ni._lib.check(self.L.nig_mixed_rollout(
            self.h, T, C.c_void_p(ring.data_ptr()), ring.stride(0), ring.shape[0],
            None if rew is None else C.c_void_p(rew.data_ptr()), None if fl is None else C.c_void_p(fl.data_ptr()),
            0 if rew is None else rew.stride(0), _stream()))

    def segment_tally(self, k):
        lib = self.ni._lib
        out = torch.empty(lib.T_ROWS, dtype=torch.float64, device="cuda:0")
        lib.check(self.L.nig_reduce_tally(C.c_void_p(self.L.nig_mixed_segment(self.h, k)), C.c_void_p(out.data_ptr()), _stream()))
        return out.cpu().numpy()

    def close(self):
        self.L.nig_mixed_destroy(self.h)


def _segments(total, n=7):
    per = (total // n) // 256 * 256
    return [total - (n - 1) * per] + [per] * (n - 1)


@pytest.mark.parametrize("names", [SEVEN, SURVEY7], ids=["readme7", "survey7"])
def test_mixed_one_launch_at_full_size(ni, oracle, names):
    """1 048 576 lanes, seven segments, 60 fused steps in ONE launch, every step's observation rows written to the
    padded [T][S_max][ld] trajectory (mixed_rollout_kernel<2>)."""
    total, T, R = 1048576, 60, 8
    counts = _segments(total)
    mix = CMixed(ni, names, counts)
    assert mix.info.lanes == total and mix.ld % 256 == 0 and mix.S == 32 and mix.A == max(
        ni._lib.env_spec(ni._lib.lib().nig_env_id(n.encode())).action_dim for n in names)
    ring = mix.ring(R)
    rew = torch.full((T, mix.ld), float("nan"), dtype=torch.float32, device="cuda:0")
    fl = torch.zeros(T, mix.ld, dtype=torch.int32, device="cuda:0")
    obs = torch.full((T, mix.S, mix.ld), float("nan"), dtype=torch.float32, device="cuda:0")
    mix.reset()
    mix.rollout(T, ring, rew, fl, obs)
    torch.cuda.synchronize()
    st = mix.state()
    Lb = ni._lib
    for k, (name, o, n) in enumerate(zip(names, mix.offsets, counts)):
        S = int(Lb.env_spec(int(mix.info.env[k])).state_dim)
        # (2) padding rows of the shared matrix stay zero
        assert not bool(st[S:, o:o + n].any()), name
        # (3) bookkeeping identities over the segment (as test_full_size_properties_powergrid)
        f = fl[:, o:o + n]
        nv = ((f >> Lb.FLAG_NVIOL_SHIFT) & 3) + ((f >> 13) & 1) * 4
        resets = ((f & Lb.FLAG_DID_RESET) != 0).sum().item()
        done = ((f & (Lb.FLAG_TERMINATED | Lb.FLAG_TRUNCATED)) != 0).sum().item()
        assert resets == done, name
        tally = mix.segment_tally(k)
        assert int(tally[Lb.T_EPISODES]) == done, name
        assert not bool(((f & Lb.FLAG_INACTIVE) != 0).any()), name
        steps_now = (f[-1] >> Lb.FLAG_STEP_SHIFT) & 0xFFFF
        assert int(steps_now.min().item()) >= 1 and not bool(torch.isnan(rew[:, o:o + n]).any()), name      # every row written
        ncons = bin(0xF & ((1 << int(Lb.env_spec(int(mix.info.env[k])).n_constraints)) - 1)).count("1")
        assert tally[Lb.T_CONSTRAINTS] == ncons * tally[Lb.T_LEN_SUM], name
        assert tally[Lb.T_SATISFIED] == tally[Lb.T_CONSTRAINTS] - tally[Lb.T_VIOL], name
        assert tally[Lb.T_VIOL] <= int(nv.sum().item()), name          # finished episodes only
    # (4) every lane of every segment == the stand-alone per-env kernel on the same columns (same ring, same seeds)
    for k, (name, o, n) in enumerate(zip(names, mix.offsets, counts)):
        solo = ni.make_batched(name, n, seed=0x5EED, env_index0=o, autoreset=True, tally=True)
        A = solo.action_dim
        srew = torch.zeros(T, n, dtype=torch.float32, device="cuda:0")
        sfl = torch.zeros(T, n, dtype=torch.int32, device="cuda:0")
        sobs = torch.zeros(T, solo.state_dim, solo.ld, dtype=torch.float32, device="cuda:0")
        solo.reset()
        solo.rollout(T, ring[:, :A, o:o + n], srew, sfl, sobs)
        torch.cuda.synchronize()
        assert torch.equal(solo.state_soa.view(torch.int32), st[:solo.state_dim, o:o + n].view(torch.int32)), name
        assert torch.equal(srew, rew[:, o:o + n]) and torch.equal(sfl, fl[:, o:o + n]), name
        # the observation rows of every step == the stand-alone kernel's trajectory; rows >= S of the segment untouched
        assert torch.equal(sobs[:, :, :n].view(torch.int32), obs[:, :solo.state_dim, o:o + n].view(torch.int32)), name
        assert bool(torch.isnan(obs[:, solo.state_dim:, o:o + n]).all()), name
        keep = (sfl[T - 1] & Lb.FLAG_DID_RESET) == 0          # lanes that did not reset in the last step: its row is the final state
        assert torch.equal(obs[T - 1, :solo.state_dim, o:o + n][:, keep].view(torch.int32), st[:solo.state_dim, o:o + n][:, keep].view(torch.int32)), name
        del sobs
        # ... and the stand-alone kernel's first / last wave == the CPU oracle with the SAME actions
        for lo in (0, n - 64):
            acts = ring[:, :A, o + lo:o + lo + 64].permute(0, 2, 1).contiguous().cpu().numpy()      # [R, 64, A]
            ref = _oracle_with_ring(oracle, name, 64, T, acts, env0=o + lo)
            got = st[:solo.state_dim, o + lo:o + lo + 64].t().contiguous().cpu().numpy()
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (name, lo)
        solo.close()
    mix.close()


def _oracle_with_ring(oracle, name, n, T, acts, env0, seed=0x5EED):
    """The oracle's free-running generator (reset draws, step noise, auto-reset) with the actions of a ring:
    step k uses ring slot k % R.  Stepped one call at a time through the teacher-forced entry point."""
    sp = oracle.spec(name)
    R = acts.shape[0]
    max_steps = sp.max_episode_steps
    state = np.stack([oracle.reset(name, oracle.gen_reset_noise(name, seed, env0 + i, 0), flavor=oracle.MATH_POLY)[0]
                      if sp.k_reset else oracle.reset(name, np.zeros((1, 1)), flavor=oracle.MATH_POLY)[0] for i in range(n)])
    step = np.zeros(n, dtype=np.int32)
    for k in range(T):
        t = k + 1
        noise = (np.stack([oracle.gen_step_noise(name, seed, env0 + i, t) for i in range(n)]) if sp.k_step else None)
        r = oracle.step(name, state, acts[k % R], noise, step, max_steps=max_steps, flavor=oracle.MATH_POLY)
        state, step = r["state_next"], step + 1
        fin = (r["terminated"] | r["truncated"]) != 0
        for i in np.nonzero(fin)[0]:
            nz = oracle.gen_reset_noise(name, seed, env0 + i, t) if sp.k_reset else np.zeros((1, 1))
            state[i] = oracle.reset(name, nz, flavor=oracle.MATH_POLY)[0]
            step[i] = 0
    return state


def test_rollout_mixed_over_caller_owned_handles(ni):
    """nig_rollout_mixed on handles the caller created and bound to its own matrix (MixedBatchedEnv, ragged
    segment sizes, frozen lanes): equals the one-launch-per-segment form."""
    segs = [("PowerGrid-v0", 1000), ("ChemicalReactor-v0", 777), ("SupplyChain-v0", 256), ("RobotAssembly-v0", 1300)]
    T, R = 33, 5
    a = ni.MixedBatchedEnv(segs, seed=3, autoreset=True, tally=True, fused=True)
    b = ni.MixedBatchedEnv(segs, seed=3, autoreset=True, tally=True, fused=False)
    ring = torch.zeros(R, a.A_max, a.ld, dtype=torch.float32, device=a.device)
    for s in range(R):
        a.fill_actions(40 + s, ring[s])
    ra, fa = (torch.zeros(T, a.ld, dtype=torch.float32, device=a.device), torch.zeros(T, a.ld, dtype=torch.int32, device=a.device))
    rb, fb = torch.zeros_like(ra), torch.zeros_like(fa)
    oa = torch.full((T, a.S_max, a.ld), float("nan"), dtype=torch.float32, device=a.device)     # observation rows of every step
    ob = torch.full_like(oa, float("nan"))
    a.reset(); b.reset()
    a.rollout(T, ring, ra, fa, oa); b.rollout(T, ring, rb, fb, ob)
    a.rollout(7, ring); b.rollout(7, ring)                      # output-free form, odd length
    a.rollout(5, ring, ra[:5], fa[:5]); b.rollout(5, ring, rb[:5], fb[:5])      # reward + flags only
    # short launches: the paired bodies (ChemicalReactor, the plants: four steps per loop pass) and the two-step ones run their
    # tails only; the launch counter alternates between odd and even starts (paired / unpaired body of the same segment)
    for n in (1, 2, 3, 4, 1, 3):
        a.rollout(n, ring, ra[:n], fa[:n]); b.rollout(n, ring, rb[:n], fb[:n])
    torch.cuda.synchronize()
    assert torch.equal(a.state_soa.view(torch.int32), b.state_soa.view(torch.int32))
    assert torch.equal(ra, rb) and torch.equal(fa, fb)
    # ragged segments: whole blocks run the unpredicated bodies (PowerGrid the LDS-resident one), the last block of a
    # segment the predicated one; rows >= S of a segment, padding columns and rows of other segments stay untouched (NaN)
    assert torch.equal(oa.view(torch.int32), ob.view(torch.int32))
    for env, o in zip(a.envs, a.offsets):
        assert not bool(torch.isnan(oa[:, :env.state_dim, o:o + env.batch]).any())
        assert bool(torch.isnan(oa[:, env.state_dim:, o:o + env.batch]).all()) or env.state_dim == a.S_max
    for x, y in zip(a.envs, b.envs):
        assert torch.equal(x.ctr, y.ctr) and torch.equal(x.tally, y.tally) and x.counter == y.counter == T + 7 + 5 + 14
    a.close(); b.close()


def test_mixed_abi_argument_checks(ni):
    L, lib = ni._lib.lib(), ni._lib
    h = C.c_void_p()
    ids, cnt = (C.c_int32 * 2)(0, 99), (C.c_int64 * 2)(10, 10)
    assert L.nig_create_mixed(2, ids, cnt, 0, 1, 0, 0, C.byref(h)) == 1 and not h.value      # unknown env id
    ids = (C.c_int32 * 2)(0, 1)
    assert L.nig_create_mixed(13, ids, cnt, 0, 1, 0, 0, C.byref(h)) == 1                      # too many segments
    lib.check(L.nig_create_mixed(2, ids, cnt, 0, 1, 0, lib.F_AUTORESET, C.byref(h)))
    info = lib.MixedInfo()
    lib.check(L.nig_mixed_get_info(h, C.byref(info)))
    assert (info.n_segments, info.state_dim_max, info.action_dim_max, info.lanes, info.ld) == (2, 32, 8, 20, 512)
    assert list(info.offset)[:2] == [0, 256] and list(info.count)[:2] == [10, 10]
    ring = torch.zeros(1, 8, 512, device="cuda:0")
    rew = torch.zeros(512, device="cuda:0")
    assert L.nig_mixed_rollout(h, 3, C.c_void_p(ring.data_ptr()), 8 * 512, 1, C.c_void_p(rew.data_ptr()), None, 0, None) == 1   # reward without flags
    assert L.nig_mixed_rollout(h, 3, C.c_void_p(ring.data_ptr()), 100, 1, None, None, 0, None) == 1                             # slot too small
    lib.check(L.nig_mixed_reset(h, None))
    lib.check(L.nig_mixed_rollout(h, 3, C.c_void_p(ring.data_ptr()), 8 * 512, 1, None, None, 0, None))
    # nig_mixed_step == the per-segment step kernels on the same columns
    fl = torch.zeros(512, dtype=torch.int32, device="cuda:0")
    act = torch.rand(8, 512, device="cuda:0") * 2 - 1
    solo = [ni.make_batched(n, 10, seed=1, env_index0=o) for n, o in (("ChemicalReactor-v0", 0), ("PowerGrid-v0", 256))]
    for e, o in zip(solo, (0, 256)):
        e.reset()
        e.rollout(3, ring[:, :e.action_dim, o:o + 10])
    lib.check(L.nig_mixed_step(h, C.c_void_p(act.data_ptr()), C.c_void_p(rew.data_ptr()), C.c_void_p(fl.data_ptr()), None))
    for e, o in zip(solo, (0, 256)):
        _, r, _, _, info = e.step(act[:e.action_dim, o:o + 10], layout="soa")
        assert torch.equal(r, rew[o:o + 10]) and torch.equal(info.flags, fl[o:o + 10])
        e.close()
    torch.cuda.synchronize()
    assert L.nig_mixed_segment(h, 2) is None and L.nig_mixed_segment(h, 1)
    L.nig_mixed_destroy(h)
