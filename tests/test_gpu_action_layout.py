"""-m gpu: nig_rollout with a ROW-MAJOR action ring (include/nig.h: ld_act == 0, slot = [batch][A] -- the layout of a policy's
batched output, agents/base.py:106-141 predict() -> [n, A]) against the same actions as [A][ld] rows: every observable bit for
bit.  Two paths behind the one entry point: PowerGrid's wide form reads the row-major slots natively (two 16-byte loads per
lane), every other kernel form gets rows from a transposing copy the handle owns -- told apart here by device memory: the copy
is a hipMalloc of ring size, the native path allocates nothing."""
import pytest
import torch

from conftest import ENV_NAME

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ni():
    import neorl_industrial_gym_amd as ni
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    yield ni
    ni.tune(split_blocks=-1, wide_min_blocks=-1)


def _run(ni, key, B, layout, chunks, outputs, R, seed=31, max_steps=17):
    env = ni.make_batched(ENV_NAME[key], B, seed=seed, autoreset=True, tally=True, max_episode_steps=max_steps)
    rows = torch.empty(R, env.action_dim, env.ld, dtype=torch.float32, device=env.device)
    for s in range(R):
        env.fill_actions(300 + s, rows[s])
    ring = rows if layout == "rows" else rows[:, :, :B].permute(0, 2, 1).contiguous()        # [R, B, A]
    env.reset()
    outs = []
    for T in chunks:                               # every output buffer exists before the device's free memory is read
        rew = fl = obs = None
        if outputs != "none":
            rew = torch.full((T, env.ld), float("nan"), dtype=torch.float32, device=env.device)
            fl = torch.zeros(T, env.ld, dtype=torch.int32, device=env.device)
        if outputs == "aos":
            obs = torch.full((T, B, env.state_dim), float("nan"), dtype=torch.float32, device=env.device)
        outs.append((T, rew, fl, obs))
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    got = []
    for T, rew, fl, obs in outs:
        env.rollout(T, ring, rew, fl, obs)
        torch.cuda.synchronize()
        got += [t.cpu() if t is obs else t[..., :B].cpu() for t in (rew, fl, obs) if t is not None]
    used = free0 - torch.cuda.mem_get_info()[0]
    got += [env.state_soa.cpu(), env.ctr.cpu(), env.life_viol.cpu(), env.ep_return.cpu(), env.tally.cpu()]
    env.close()
    return got, used, min(R, max(chunks)) * env.action_dim * env.ld * 4        # bytes of the slots one call reads


def _same(a, b):
    assert len(a) == len(b)
    for i, (x, y) in enumerate(zip(a, b)):
        xv = x.contiguous().view(torch.int32) if x.dtype == torch.float32 else (x.contiguous().view(torch.int64) if x.dtype == torch.float64 else x)
        yv = y.contiguous().view(torch.int32) if y.dtype == torch.float32 else (y.contiguous().view(torch.int64) if y.dtype == torch.float64 else y)
        assert x.shape == y.shape and torch.equal(xv, yv), f"observable {i} differs"


@pytest.mark.parametrize("B", [4096, 4096 + 256])            # eight wide blocks; eight wide blocks + one 256-lane block (two launches)
@pytest.mark.parametrize("outputs", ["none", "rows", "aos"])
def test_powergrid_wide_form_reads_a_row_major_ring_natively(ni, outputs, B):
    """Whole 512-lane blocks in the wide form (knob at one block): chained launches with ring wrap, truncations, terminations and
    in-kernel resets; row-major ring == rows, and no ring-sized allocation appears on the device."""
    ni.tune(wide_min_blocks=1, split_blocks=0)
    kw = dict(key="pg", B=B, chunks=[200, 9], outputs=outputs, R=512)            # 512 slots x 8 x 4096 x 4 B = 64 MiB ring, 200 of them read
    a, used_a, ring_bytes = _run(ni, layout="rows", **kw)
    b, used_b, _ = _run(ni, layout="aos", **kw)
    _same(a, b)
    assert float(a[-1][0].sum()) > 0                                   # episodes finished and were tallied
    assert used_b < ring_bytes // 2, (used_b, ring_bytes)              # native: the library allocated nothing of ring size


@pytest.mark.parametrize("B,outputs,native", [(768, "aos", True),        # paired form, LDS-resident stepper (trajectory)
                                              (768, "rows", False),      # paired form, register-resident stepper: rows
                                              (66048, "rows", True),     # 258 blocks: past the paired regime, wide 256-lane form
                                              (66048 + 512, "aos", True)])
def test_powergrid_small_batch_forms_and_the_row_major_ring(ni, B, outputs, native):
    """PowerGrid below the wide threshold, default knobs: every LDS-resident form reads the row-major slots natively (the
    launcher's own predicate, rollout_rows_native), the paired regime's register-resident stepper (reward + flags / no outputs)
    takes the row copy; bit-identical to rows either way."""
    ni.tune(wide_min_blocks=-1, split_blocks=-1)
    R = 600 if B < 4096 else 16
    kw = dict(key="pg", B=B, chunks=[R, 5], outputs=outputs, R=R)
    a, _, ring_bytes = _run(ni, layout="rows", **kw)
    b, used_b, _ = _run(ni, layout="aos", **kw)
    _same(a, b)
    assert ring_bytes >= (8 << 20)
    assert (used_b < ring_bytes // 2) == native, (used_b, ring_bytes, native)


@pytest.mark.parametrize("key,B", [("pg", 4096 + 77), ("cr", 1024), ("cr", 1000), ("ra", 512)])
def test_other_forms_take_a_row_major_ring_through_the_row_copy(ni, key, B):
    """A ragged PowerGrid batch (wide + 256-lane + one-wave launches), ChemicalReactor
    and RobotAssembly in their three-wave and one-wave forms: the library transposes the ring into rows it owns (a ring-sized
    allocation appears: the slots a call reads, min(ring_len, n_steps) of them), results equal the rows' bit for bit; a second
    call reuses the buffer."""
    ni.tune(wide_min_blocks=-1, split_blocks=-1)
    kw = dict(key=key, B=B, chunks=[70, 6], outputs="aos", R=64)               # (70 steps: the ring wraps inside the call)
    a, _, ring_bytes = _run(ni, layout="rows", **kw)
    b, used_b, _ = _run(ni, layout="aos", **kw)
    _same(a, b)
    assert used_b >= ring_bytes // 2 or ring_bytes < (4 << 20), (used_b, ring_bytes)      # (small rings vanish in the allocator's granularity)


def test_row_major_ring_argument_checks(ni):
    import ctypes as C
    env = ni.make_batched("PowerGrid-v0", 1024, autoreset=True)
    env.reset()
    L = env._L
    ring = torch.zeros(4, 1024, 8, device=env.device)
    st = env._stream()
    # slot_stride smaller than one [batch][A] slot
    assert L.nig_rollout(env._h, 3, C.c_void_p(ring.data_ptr()), 0, 1024 * 8 - 1, 4, None, None, 0, None, 0, 0, st) != 0
    assert b"row-major action ring" in L.nig_last_error()
    # the step-API plan keeps [A][ld] rows
    plan = C.c_void_p()
    assert L.nig_plan_create(env._h, 3, C.c_void_p(ring.data_ptr()), 0, 1024 * 8, 4, None, None, 0, C.byref(plan)) != 0
    env.rollout(3, ring)                                               # and the good call goes through
    torch.cuda.synchronize()
    env.close()
