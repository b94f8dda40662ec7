"""BatchedIndustrialEnv: B independent env instances of one type stepped by ONE HIP kernel.

This is the batched form of IndustrialEnv (reference environments/base.py:19-228): the
same reset/step semantics per lane, with every per-step Python object of the reference
(`SafetyMetrics`, the `info` dict) replaced by a packed flag word per lane that can be
decoded lazily.  torch is used only for device memory and the stream handle; all compute
happens in libnig.so through the C ABI (include/nig.h).
"""
import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib
from .core import SafetyMetrics, make_box

ENV_IDS = {"ChemicalReactor-v0": 0, "PowerGrid-v0": 1, "RobotAssembly-v0": 2,
           # candidate rows: registered upstream (utils.py:30-31) but not instantiable there
           "AdvancedChemicalReactor-v0": 3, "AdvancedPowerGrid-v0": 4,
           # README-only upstream (README.md:28-32): build-specified plants, spec_plants.py
           "HVACControl-v0": 5, "WaterTreatment-v0": 6, "SteelAnnealing-v0": 7, "SupplyChain-v0": 8}


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class StepInfo:
    """Lazy view of the per-lane flag words of one step (the batched `info`).

    Fields are torch tensors on the env's device, decoded on first use:
      terminated, truncated, done, violation_count, critical_violations, critical_shutdown,
      constraint_violated [3,B], did_reset, inactive, step.
    """

    def __init__(self, flags: torch.Tensor, n_constraints: int = 3):
        self.flags = flags
        self.n_constraints = n_constraints

    @property
    def terminated(self):
        return (self.flags & _lib.FLAG_TERMINATED) != 0

    @property
    def truncated(self):
        return (self.flags & _lib.FLAG_TRUNCATED) != 0

    @property
    def done(self):
        return (self.flags & (_lib.FLAG_TERMINATED | _lib.FLAG_TRUNCATED)) != 0

    @property
    def violation_count(self):
        return ((self.flags >> _lib.FLAG_NVIOL_SHIFT) & 3) + ((self.flags >> 13) & 1) * 4

    @property
    def critical_violations(self):
        return (self.flags >> _lib.FLAG_NCRIT_SHIFT) & 3

    @property
    def critical_shutdown(self):
        return (self.flags & _lib.FLAG_SHUTDOWN) != 0

    @property
    def constraint_violated(self):
        return torch.stack([((self.flags >> (_lib.FLAG_VIOL_SHIFT + k)) & 1) != 0 for k in range(3)]
                           + [((self.flags >> 12) & 1) != 0])[:self.n_constraints]

    @property
    def did_reset(self):
        return (self.flags & _lib.FLAG_DID_RESET) != 0

    @property
    def inactive(self):
        return (self.flags & _lib.FLAG_INACTIVE) != 0

    @property
    def step(self):
        return (self.flags >> _lib.FLAG_STEP_SHIFT) & 0xFFFF

    def safety_metrics(self, lane: int) -> SafetyMetrics:
        """The reference's per-step SafetyMetrics object for one lane (base.py:118-124)."""
        f = int(self.flags[lane].item())
        nv = ((f >> _lib.FLAG_NVIOL_SHIFT) & 3) + ((f >> 13) & 1) * 4
        nc = (f >> _lib.FLAG_NCRIT_SHIFT) & 3
        n = self.n_constraints
        return SafetyMetrics(constraints_satisfied=n - nv, total_constraints=n, violation_count=nv,
                             critical_violations=nc, safety_score=(n - nv) / n)


class BatchedIndustrialEnv:
    """B lanes of `env_id` on one GPU.

    Args mirror IndustrialEnv.__init__ (base.py:22-29) plus the batch controls:
      batch            number of env instances (lanes)
      device           torch device ("cuda:0")
      seed             key of the counter-based generator used in fast mode
      env_index0       global index of lane 0 (sharding-invariant RNG streams)
      autoreset        finished lanes start their next episode inside the step kernel
      tally            keep per-lane episode tallies for evaluate_with_safety
    Layouts: `state` / `obs` is the library-owned SoA array viewed as [B, S] with strides
    (1, ld) -- a zero-copy view a policy network can consume directly; actions are
    accepted as [A, B] (native SoA, no copy) or [B, A] (transposed on the way in).
    """

    def __init__(self, env_id: str, batch: int, device="cuda:0", seed: int = 0x5EED, env_index0: int = 0,
                 max_episode_steps: Optional[int] = None, dt: Optional[float] = None, autoreset: bool = True,
                 tally: bool = False, bind_state: Optional[torch.Tensor] = None):
        if env_id not in ENV_IDS:
            available = ", ".join(ENV_IDS.keys())
            raise ValueError(f"Unknown environment '{env_id}'. Available: {available}")
        self.env_id = env_id
        self._eid = ENV_IDS[env_id]
        self._L = _lib.lib()
        self.spec = _lib.env_spec(self._eid)
        self.state_dim, self.action_dim = int(self.spec.state_dim), int(self.spec.action_dim)
        self.batch = int(batch)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("BatchedIndustrialEnv needs a HIP device (torch device type 'cuda'); "
                               "there is no CPU fallback")
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible to torch; BatchedIndustrialEnv has no CPU fallback")
        self.max_episode_steps = int(max_episode_steps) if max_episode_steps else int(self.spec.max_episode_steps)
        self.dt = float(dt) if dt else float(self.spec.dt)
        self.autoreset, self.tally_enabled = bool(autoreset), bool(tally)
        self._flags = (_lib.F_AUTORESET if autoreset else 0) | (_lib.F_TALLY if tally else 0)
        lay = _lib.layout_query(self._eid, self.batch, self._flags)
        self.ld = int(lay.ld)
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        with torch.cuda.device(dev_index):
            self._ws = torch.empty(int(lay.bytes), dtype=torch.uint8, device=self.device)
            assert self._ws.data_ptr() % 256 == 0
            h = C.c_void_p()
            _lib.check(self._L.nig_create(self._eid, self.batch, dev_index, C.c_uint64(seed), C.c_uint64(env_index0),
                                          self.max_episode_steps if max_episode_steps else 0,
                                          C.c_double(self.dt if dt else 0.0), self._flags,
                                          C.c_void_p(self._ws.data_ptr()), C.byref(h)))
        self._h = h
        self._dev_index = dev_index
        B, ld, S = self.batch, self.ld, self.state_dim

        def view(off, nbytes, dtype):
            return self._ws[off:off + nbytes].view(dtype)

        if bind_state is None:
            self.state_soa = view(lay.off_state, S * ld * 4, torch.float32).view(S, ld)[:, :B]   # [S, B]
        else:
            # caller-owned padded SoA array (mixed batches): a [>=S, B] float32 view with unit column stride
            assert bind_state.dtype == torch.float32 and bind_state.device == self.device
            assert bind_state.shape[0] >= S and bind_state.shape[1] == B and bind_state.stride(1) == 1
            _lib.check(self._L.nig_bind_state(self._h, C.c_void_p(bind_state.data_ptr()), bind_state.stride(0)))
            self._bound = bind_state
            self.state_soa = bind_state[:S]
        self.obs = self.state_soa.t()                                                       # [B, S] strided view
        self.ctr = view(lay.off_ctr, ld * 4, torch.int32)[:B]
        self.life_viol = view(lay.off_life_viol, ld * 8, torch.int64)[:B]
        if tally:
            self.ep_return = view(lay.off_ep_return, ld * 8, torch.float64)[:B]
            self.tally = view(lay.off_tally, _lib.T_ROWS * ld * 8, torch.float64).view(_lib.T_ROWS, ld)[:, :B]
        else:
            self.ep_return = self.tally = None
        self.reward = torch.zeros(B, dtype=torch.float32, device=self.device)
        self.reward64 = torch.zeros(B, dtype=torch.float64, device=self.device)
        self.flags = torch.zeros(B, dtype=torch.int32, device=self.device)
        self._act_soa = torch.zeros(self.action_dim, ld, dtype=torch.float32, device=self.device)
        self.observation_space = make_box(-np.inf, np.inf, (self.state_dim,), np.float32)
        self.action_space = make_box(-1.0, 1.0, (self.action_dim,), np.float32)

    # ------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.nig_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _noise(self, x, rows):
        """numpy/torch [rows, B] (or [B, rows]) fp64 -> contiguous device [rows, B]."""
        if x is None:
            return None
        t = torch.as_tensor(x, dtype=torch.float64, device=self.device)
        if t.shape == (self.batch, rows) and rows != self.batch:
            t = t.t()
        if t.shape != (rows, self.batch):
            raise ValueError(f"noise must have shape ({rows}, {self.batch}), got {tuple(t.shape)}")
        return t.contiguous()

    # ------------------------------------------------------------------
    @property
    def counter(self) -> int:
        t = C.c_uint32()
        _lib.check(self._L.nig_get_counter(self._h, C.byref(t)))
        return int(t.value)

    @counter.setter
    def counter(self, t: int):
        _lib.check(self._L.nig_set_counter(self._h, C.c_uint32(t)))

    def set_constraint_mask(self, mask: int):
        _lib.check(self._L.nig_set_constraint_mask(self._h, C.c_uint32(mask)))

    @property
    def current_step(self):
        return self.ctr & _lib.CTR_STEP_MASK

    @property
    def done(self):
        return (self.ctr & _lib.CTR_DONE) != 0

    @property
    def violation_count(self):
        return (self.ctr >> _lib.CTR_VIOL_SHIFT) & 0xFFFF

    @property
    def total_violations(self):
        """base.py:57,183: lifetime count = finished episodes + the running one."""
        running = torch.where(self.done, torch.zeros_like(self.ctr), self.violation_count)
        return self.life_viol + running.to(torch.int64)

    # ------------------------------------------------------------------
    def reset(self, mask=None, init_noise=None):
        """IndustrialEnv.reset (base.py:133-155) for all lanes or those where mask != 0.
        init_noise: optional [k_reset, B] fp64 draws (parity mode).  Returns the obs view."""
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        nz = self._noise(init_noise, int(self.spec.k_reset))
        with torch.cuda.device(self._dev_index):
            _lib.check(self._L.nig_reset(self._h, _ptr(m), _ptr(nz), self.batch, self._stream()))
        return self.obs

    def step(self, actions, step_noise=None, reset_noise=None, final_obs: Optional[torch.Tensor] = None,
             layout: Optional[str] = None):
        """IndustrialEnv.step (base.py:157-213) for every lane: one kernel launch.

        actions: [A, B] (SoA, zero-copy) or [B, A] tensor/array on any device; float32, or float64 --
        upstream clips without casting (base.py:167), so a float64 action vector makes NumPy evaluate
        the action-dependent arithmetic in float64: float64 input goes through nig_step64, which
        follows that (CR / PG / RA; the other envs take float32).  Other dtypes are cast to float32.
        `layout` ("soa"/"aos") disambiguates when A == B (default then: [B, A]).
        Returns (obs [B,S] view, reward f32 [B], terminated [B], truncated [B], StepInfo).
        """
        A, B = self.action_dim, self.batch
        a = torch.as_tensor(actions, device=self.device)
        act64 = a.dtype == torch.float64
        if not act64 and a.dtype != torch.float32:
            a = a.to(torch.float32)
        if layout in ("soa", "aos"):
            soa = layout == "soa"
        else:
            soa = (a.shape == (A, B)) and (A != B)
        if soa and a.shape != (A, B):
            raise ValueError(f"actions must have shape ({A}, {B}) for layout='soa', got {tuple(a.shape)}")
        if not soa and a.shape != (B, A):
            raise ValueError(f"actions must have shape ({A}, {B}) or ({B}, {A}), got {tuple(a.shape)}")
        if soa and a.stride(1) == 1 and a.stride(0) >= B:
            act, ld_act = a, a.stride(0)                      # native SoA, zero-copy
        elif (not soa) and a.stride(0) == 1 and a.stride(1) >= B:
            act, ld_act = a, a.stride(1)                      # [B,A] view of an SoA buffer, zero-copy
        elif act64:
            if getattr(self, "_act64_soa", None) is None:
                self._act64_soa = torch.zeros(self.action_dim, self.ld, dtype=torch.float64, device=self.device)
            self._act64_soa[:, :B].copy_(a if soa else a.t())
            act, ld_act = self._act64_soa, self.ld
        else:
            self._act_soa[:, :B].copy_(a if soa else a.t())   # one transposing copy
            act, ld_act = self._act_soa, self.ld
        sn = self._noise(step_noise, int(self.spec.k_step)) if int(self.spec.k_step) else None
        rn = self._noise(reset_noise, int(self.spec.k_reset))
        ld_obs = 0
        if final_obs is not None:
            assert final_obs.dtype == torch.float32 and final_obs.shape[0] == self.state_dim and final_obs.stride(1) == 1
            ld_obs = final_obs.stride(0)
        with torch.cuda.device(self._dev_index):
            _lib.check((self._L.nig_step64 if act64 else self._L.nig_step)(
                self._h, _ptr(act), ld_act, _ptr(sn), _ptr(rn), B, _ptr(self.reward), _ptr(self.reward64),
                _ptr(self.flags), _ptr(final_obs), ld_obs, self._stream()))
        info = StepInfo(self.flags, int(self.spec.n_constraints))
        return self.obs, self.reward, info.terminated, info.truncated, info

    def step_raw(self, act_soa: torch.Tensor, ld_act: int, reward=True, reward64=False, flags=True):
        """Hot-loop form: no tensor conversions, no decoding; `act_soa` is float32 [A, >=B]."""
        _lib.check(self._L.nig_step(self._h, C.c_void_p(act_soa.data_ptr()), ld_act, None, None, 0,
                                    _ptr(self.reward) if reward else None,
                                    _ptr(self.reward64) if reward64 else None,
                                    _ptr(self.flags) if flags else None, None, 0, self._stream()))

    def make_plan(self, n_steps: int, action_ring: torch.Tensor, reward_ring: Optional[torch.Tensor] = None,
                  flags_ring: Optional[torch.Tensor] = None) -> "StepPlan":
        """Record n_steps fast-mode steps as one hipGraph.  action_ring: float32 [R, A, ld>=B];
        step k of every replay reads slot k % R.  Optional reward/flags rings [R, B] (or [B])."""
        return StepPlan(self, n_steps, action_ring, reward_ring, flags_ring)

    def rollout(self, n_steps: int, action_ring: torch.Tensor, reward_out: Optional[torch.Tensor] = None,
                flags_out: Optional[torch.Tensor] = None, obs_out: Optional[torch.Tensor] = None):
        """n_steps fused steps in ONE kernel launch (state stays in registers; fast mode).
        action_ring: float32 [R, A, ld>=B] (rows), or contiguous [R, B, A] (row-major: what a policy network's batched output
        looks like; read natively by PowerGrid's wide form, transposed into rows by the library for every other kernel form --
        include/nig.h nig_rollout, ld_act == 0); step k reads slot k % R.
        reward_out float32 / flags_out int32: [n_steps, >=B] (per-step rows) or [B] (overwritten).
        obs_out: float32 trajectory of returned observations, [n_steps, S, >=B] (SoA rows) or
        contiguous [n_steps, B, S] (row-major transitions, D4RL layout; fastest to write)."""
        assert action_ring.dtype == torch.float32 and action_ring.dim() == 3 and action_ring.stride(2) == 1
        R = action_ring.shape[0]
        if action_ring.shape[1] == self.action_dim and action_ring.shape[2] >= self.batch:
            A, ld = action_ring.shape[1], action_ring.stride(1)                          # rows [R, A, ld]
        else:
            assert action_ring.shape[1:] == (self.batch, self.action_dim) and action_ring.stride(1) == self.action_dim, \
                "action_ring is [R, A, ld >= batch] or contiguous [R, batch, A]"
            A, ld = self.action_dim, 0                                                   # row-major [R, B, A]

        def out(t, dtype):
            if t is None:
                return None, 0
            assert t.dtype == dtype and t.stride(-1) == 1
            if t.dim() == 2:
                assert t.shape[0] >= n_steps
                return C.c_void_p(t.data_ptr()), t.stride(0)
            return C.c_void_p(t.data_ptr()), 0

        rp, rs = out(reward_out, torch.float32)
        fp, fs = out(flags_out, torch.int32)
        assert (rp is None) == (fp is None), "reward_out and flags_out go together (both or neither)"
        assert rp is None or rs == fs, "reward/flags outputs must share their row stride"
        assert obs_out is None or rp is not None, "an observation trajectory needs reward_out and flags_out too"
        op, ldo, so = None, 0, 0
        if obs_out is not None:
            assert obs_out.dtype == torch.float32 and obs_out.dim() == 3 and obs_out.stride(2) == 1
            assert obs_out.shape[0] >= n_steps
            if obs_out.shape[1:] == (self.batch, self.state_dim) and obs_out.stride(1) == self.state_dim \
                    and self.batch != self.state_dim:
                op, ldo, so = C.c_void_p(obs_out.data_ptr()), 0, obs_out.stride(0)          # row-major [T,B,S]
            else:
                assert obs_out.shape[1] == self.state_dim and obs_out.shape[2] >= self.batch
                op, ldo, so = C.c_void_p(obs_out.data_ptr()), obs_out.stride(1), obs_out.stride(0)
        with torch.cuda.device(self._dev_index):
            _lib.check(self._L.nig_rollout(self._h, int(n_steps), C.c_void_p(action_ring.data_ptr()), ld,
                                           action_ring.stride(0), R, rp, fp, rs or fs, op, ldo, so, self._stream()))

    def rollout_noise(self, n_steps: int, action_ring: torch.Tensor, step_noise: Optional[torch.Tensor],
                      reset_noise: Optional[torch.Tensor], reward_out: torch.Tensor, flags_out: torch.Tensor,
                      obs_out: torch.Tensor):
        """rollout() on the reference's RECORDED draws (nig_rollout_noise): the same kernel form the batch would
        take, with np.random's values injected.  action_ring float32 [n_steps, A, ld]; step_noise float64
        [n_steps, k_step, ld] (None for an env without step noise); reset_noise float64 [n_steps, k_reset, ld]
        (auto-reset handles: the initial-state draws of a lane that finishes in that step); reward_out /
        flags_out [n_steps, >=B]; obs_out float32 contiguous [n_steps, B, S]."""
        assert action_ring.dtype == torch.float32 and action_ring.dim() == 3 and action_ring.stride(2) == 1
        assert action_ring.shape[0] >= n_steps and action_ring.shape[1] == self.action_dim
        ldn = 0

        def nz(t):
            nonlocal ldn
            if t is None:
                return None, 0
            assert t.dtype == torch.float64 and t.dim() == 3 and t.stride(2) == 1 and t.shape[0] >= n_steps
            assert ldn in (0, t.stride(1)), "step_noise and reset_noise share their row pitch"
            ldn = t.stride(1)
            return C.c_void_p(t.data_ptr()), t.stride(0)

        sp, ss = nz(step_noise)
        rp, rs = nz(reset_noise)
        assert reward_out.dtype == torch.float32 and flags_out.dtype == torch.int32 and reward_out.dim() == 2
        assert reward_out.stride(0) == flags_out.stride(0) and reward_out.shape[0] >= n_steps
        assert obs_out.dtype == torch.float32 and obs_out.is_contiguous() and obs_out.shape[1:] == (self.batch, self.state_dim)
        with torch.cuda.device(self._dev_index):
            _lib.check(self._L.nig_rollout_noise(
                self._h, int(n_steps), C.c_void_p(action_ring.data_ptr()), action_ring.stride(1), action_ring.stride(0),
                action_ring.shape[0], sp, ss, rp, rs, ldn, C.c_void_p(reward_out.data_ptr()),
                C.c_void_p(flags_out.data_ptr()), reward_out.stride(0), C.c_void_p(obs_out.data_ptr()), obs_out.stride(0),
                self._stream()))

    # ------------------------------------------------------------------
    def set_policy(self, policy):
        """Install an on-device policy (policies.DevicePolicy) for rollout_policy()."""
        P = policy.to_struct() if hasattr(policy, "to_struct") else policy
        with torch.cuda.device(self._dev_index):
            _lib.check(self._L.nig_set_policy(self._h, C.byref(P), self._stream()))
        self._policy = policy

    def set_mlp_policy(self, weights):
        """Install the reference agents' actor (S->256->256->A, ReLU, tanh) for rollout_mlp():
        weights = [(W1 [S,256], b1), (W2 [256,256], b2), (W3 [256,A], b3)], host arrays, [in, out]."""
        W = [np.ascontiguousarray(np.asarray(x), dtype=np.float32) for pair in weights for x in pair]
        S, A, Hd = self.state_dim, self.action_dim, 256
        assert [w.shape for w in W] == [(S, Hd), (Hd,), (Hd, Hd), (Hd,), (Hd, A), (A,)], [w.shape for w in W]
        with torch.cuda.device(self._dev_index):
            _lib.check(self._L.nig_set_mlp_policy(self._h, Hd, *[w.ctypes.data_as(C.c_void_p) for w in W], self._stream()))

    def rollout_mlp(self, n_steps: int, reward_out=None, flags_out=None, obs_out=None, act_out=None):
        """Closed loop with the installed MLP actor evaluated by f32 MFMA inside the env kernel
        (same outputs as rollout_policy)."""
        self.rollout_policy(n_steps, reward_out, flags_out, obs_out, act_out, _fn=self._L.nig_rollout_mlp)

    def rollout_policy(self, n_steps: int, reward_out: Optional[torch.Tensor] = None,
                       flags_out: Optional[torch.Tensor] = None, obs_out: Optional[torch.Tensor] = None,
                       act_out: Optional[torch.Tensor] = None, _fn=None):
        """n_steps closed-loop steps (action = installed policy(observation)) in ONE launch.
        obs_out: float32 contiguous [n_steps, B, S] (observation the policy acted on);
        act_out: float32 [n_steps, A, >=B]; reward_out / flags_out: [n_steps, >=B] or [B]."""
        def out(t, dtype):
            if t is None:
                return None, 0
            assert t.dtype == dtype and t.stride(-1) == 1
            if t.dim() == 2:
                assert t.shape[0] >= n_steps
                return C.c_void_p(t.data_ptr()), t.stride(0)
            return C.c_void_p(t.data_ptr()), 0

        rp, rs = out(reward_out, torch.float32)
        fp, fs = out(flags_out, torch.int32)
        assert rp is None or fp is None or rs == fs
        op, so = None, 0
        if obs_out is not None:
            assert obs_out.dtype == torch.float32 and obs_out.is_contiguous()
            assert obs_out.shape[0] >= n_steps and obs_out.shape[1:] == (self.batch, self.state_dim)
            op, so = C.c_void_p(obs_out.data_ptr()), obs_out.stride(0)
        ap, lda, sa = None, 0, 0
        if act_out is not None:
            assert act_out.dtype == torch.float32 and act_out.dim() == 3 and act_out.stride(2) == 1
            assert act_out.shape[0] >= n_steps and act_out.shape[1] == self.action_dim and act_out.shape[2] >= self.batch
            ap, lda, sa = C.c_void_p(act_out.data_ptr()), act_out.stride(1), act_out.stride(0)
        with torch.cuda.device(self._dev_index):
            _lib.check((_fn or self._L.nig_rollout_policy)(self._h, int(n_steps), rp, fp, rs or fs, op, so, ap, lda, sa,
                                                           self._stream()))

    def get_dataset(self, quality: str = "mixed", scale: int = 1, chunk: int = 100):
        """Batched env.get_dataset(quality): every lane is one episode of the reference's
        data-collection loop (chemical_reactor.py:324-420, power_grid.py:194-249,
        robot_assembly.py:246-308) driven by the on-device behaviour policy; `scale` multiplies
        the upstream episode count.  Needs batch >= episodes*scale and autoreset=False.
        Returns the D4RL-style dict as torch tensors on the device (episode-major order)."""
        from .policies import DATASET_SHAPE, behaviour_policy
        assert not self.autoreset, "dataset generation needs autoreset=False (one episode per lane)"
        n_eps, cap = DATASET_SHAPE[self.env_id][quality]
        n_eps *= int(scale)
        if n_eps > self.batch:
            raise ValueError(f"batch {self.batch} < {n_eps} episodes")
        B, S, A = self.batch, self.state_dim, self.action_dim
        self.set_policy(behaviour_policy(self.env_id, quality))
        mask = torch.zeros(B, dtype=torch.uint8, device=self.device)
        mask[:n_eps] = 1
        self.ctr.fill_(_lib.CTR_DONE)                 # lanes beyond n_eps stay idle
        self.reset(mask=mask)
        obs_c, act_c, rew_c, live_c, term_c = [], [], [], [], []
        done_steps = 0
        while done_steps < cap:
            T = min(chunk, cap - done_steps)
            obs = torch.empty(T, B, S, dtype=torch.float32, device=self.device)
            act = torch.empty(T, A, self.ld, dtype=torch.float32, device=self.device)
            rew = torch.empty(T, self.ld, dtype=torch.float32, device=self.device)
            fl = torch.empty(T, self.ld, dtype=torch.int32, device=self.device)
            self.rollout_policy(T, rew, fl, obs, act)
            live = (fl[:, :n_eps] & _lib.FLAG_INACTIVE) == 0
            obs_c.append(obs[:, :n_eps]); act_c.append(act[:, :, :n_eps]); rew_c.append(rew[:, :n_eps])
            live_c.append(live)
            both = (fl[:, :n_eps] & (_lib.FLAG_TERMINATED | _lib.FLAG_TRUNCATED)) != 0
            term_c.append(both if self.env_id == "ChemicalReactor-v0" else (fl[:, :n_eps] & _lib.FLAG_TERMINATED) != 0)
            done_steps += T
            if not bool(live[-1].any().item()):
                break
        live = torch.cat(live_c).t()                                  # [episodes, steps] -> episode-major
        sel = live.reshape(-1)
        obs = torch.cat(obs_c).permute(1, 0, 2).reshape(-1, S)[sel]
        act = torch.cat(act_c).permute(2, 0, 1).reshape(-1, A)[sel]
        rew = torch.cat(rew_c).t().reshape(-1)[sel]
        term = torch.cat(term_c).t().reshape(-1)[sel]
        out = {"observations": obs, "actions": act, "rewards": rew, "terminals": term}
        if self.env_id == "ChemicalReactor-v0":
            out["timeouts"] = torch.zeros_like(term)
        return out

    def fill_actions(self, t: int, out: Optional[torch.Tensor] = None):
        """Synthetic uniform [-1,1) actions of the bench workload for launch counter t -> [A, B]."""
        if out is None:
            out = torch.empty(self.action_dim, self.ld, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self._dev_index):
            _lib.check(self._L.nig_fill_actions(self._h, C.c_uint32(t), _ptr(out), out.stride(0), self._stream()))
        return out

    # ------------------------------------------------------------------
    def set_state(self, state=None, current_step=None, violation_count=None, done=None):
        """Teacher forcing: state [B,S] or [S,B]; counters optional (per-lane arrays)."""
        B, S = self.batch, self.state_dim
        st = None
        if state is not None:
            t = torch.as_tensor(state, dtype=torch.float32, device=self.device)
            st = (t.t() if t.shape == (B, S) else t).contiguous()    # [B,S] wins when B == S
            assert st.shape == (S, B)
        c = None
        if current_step is not None or violation_count is not None or done is not None:
            cs = torch.as_tensor(0 if current_step is None else current_step, device=self.device).to(torch.int64)
            vc = torch.as_tensor(0 if violation_count is None else violation_count, device=self.device).to(torch.int64)
            dn = torch.as_tensor(False if done is None else done, device=self.device).to(torch.int64)
            word = (cs & _lib.CTR_STEP_MASK) | (dn * _lib.CTR_DONE) | ((vc & 0xFFFF) << _lib.CTR_VIOL_SHIFT)
            word = torch.broadcast_to(word, (B,))
            c = torch.where(word >= 2 ** 31, word - 2 ** 32, word).to(torch.int32).contiguous()
        with torch.cuda.device(self._dev_index):
            _lib.check(self._L.nig_set_state(self._h, _ptr(st), B, _ptr(c), self._stream()))

    def get_state(self) -> torch.Tensor:
        """A contiguous [B, S] copy of the current state."""
        return self.obs.contiguous()

    def get_safety_metrics(self):
        """Per-lane SafetyMetrics fields of the last step as an int32 tensor [5, B]
        (rows: constraints_satisfied, total_constraints, violation_count, critical_violations,
        satisfied = safety_score * total)."""
        out = torch.empty(5, self.batch, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self._dev_index):
            _lib.check(self._L.nig_get_safety_metrics(self._h, _ptr(self.flags), _ptr(out), self.batch, self._stream()))
        return out

    def reduce_tally(self) -> torch.Tensor:
        """Device reduction of the per-lane tallies -> float64 [T_ROWS] partial vector."""
        out = torch.empty(_lib.T_ROWS, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self._dev_index):
            _lib.check(self._L.nig_reduce_tally(self._h, _ptr(out), self._stream()))
        return out


class StepPlan:
    """n consecutive env.step() launches replayed as one hipGraph (include/nig.h nig_plan_*)."""

    def __init__(self, env: BatchedIndustrialEnv, n_steps: int, action_ring: torch.Tensor,
                 reward_ring: Optional[torch.Tensor], flags_ring: Optional[torch.Tensor]):
        assert action_ring.dtype == torch.float32 and action_ring.dim() == 3 and action_ring.stride(2) == 1
        R, A, ld = action_ring.shape[0], action_ring.shape[1], action_ring.stride(1)
        assert A == env.action_dim and ld >= env.batch
        self.env, self.n_steps = env, int(n_steps)
        self._keep = (action_ring, reward_ring, flags_ring)

        def out(t, dtype):
            if t is None:
                return None, 0
            assert t.dtype == dtype and t.stride(-1) == 1
            return C.c_void_p(t.data_ptr()), (t.stride(0) if t.dim() == 2 else 0)

        rp, rs = out(reward_ring, torch.float32)
        fp, fs = out(flags_ring, torch.int32)
        assert rs == fs or rp is None or fp is None, "reward/flags rings must share their slot stride"
        p = C.c_void_p()
        with torch.cuda.device(env._dev_index):
            _lib.check(env._L.nig_plan_create(env._h, self.n_steps, C.c_void_p(action_ring.data_ptr()), ld,
                                              action_ring.stride(0), R, rp, fp, rs or fs, C.byref(p)))
        self._p = p

    def launch(self):
        _lib.check(self.env._L.nig_plan_launch(self._p, self.env._stream()))

    def close(self):
        if getattr(self, "_p", None):
            self.env._L.nig_plan_destroy(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MixedBatchedEnv:
    """Several env types in one padded batch (BASELINE config "all envs mixed-batch, heterogeneous
    state dims, padded SoA").  Lanes are grouped in contiguous, 256-aligned segments, one env type
    each (every wavefront is homogeneous); all segments share ONE observation matrix
    `state_soa` [S_max, LD] (rows >= S of a segment stay zero) and one action matrix layout
    [A_max, LD].  Each segment is a BatchedIndustrialEnv bound to its columns; a step or rollout
    launches one kernel per segment, each on its own HIP stream."""

    def __init__(self, segments, device="cuda:0", seed: int = 0x5EED, autoreset: bool = True, tally: bool = False,
                 env_index0: int = 0, fused: bool = True):
        self.device = torch.device(device)
        segs = list(segments.items()) if isinstance(segments, dict) else list(segments)
        self.S_max = max(int(_lib.env_spec(ENV_IDS[e]).state_dim) for e, _ in segs)
        self.A_max = max(int(_lib.env_spec(ENV_IDS[e]).action_dim) for e, _ in segs)
        offs, off = [], 0
        for _, n in segs:
            offs.append(off)
            off += (int(n) + 255) // 256 * 256
        self.ld = off
        self.batch = sum(int(n) for _, n in segs)
        self.state_soa = torch.zeros(self.S_max, self.ld, dtype=torch.float32, device=self.device)
        self.obs = self.state_soa.t()
        self.reward = torch.zeros(self.ld, dtype=torch.float32, device=self.device)
        self.flags = torch.zeros(self.ld, dtype=torch.int32, device=self.device)
        self.offsets = offs
        self.envs = []
        for (e, n), o in zip(segs, offs):
            self.envs.append(BatchedIndustrialEnv(e, int(n), device=device, seed=seed, env_index0=env_index0 + o,
                                                  autoreset=autoreset, tally=tally,
                                                  bind_state=self.state_soa[:, o:o + int(n)]))
        self._streams = [torch.cuda.Stream(device=self.device) for _ in self.envs]
        self.fused = bool(fused)
        n = len(self.envs)
        self._hs = (C.c_void_p * n)(*[e._h for e in self.envs])
        self._offs = (C.c_int64 * n)(*offs)

    def _fan(self, fn):
        cur = torch.cuda.current_stream(self.device)
        ev = torch.cuda.Event(); ev.record(cur)
        for env, st, o in zip(self.envs, self._streams, self.offsets):
            st.wait_event(ev)
            with torch.cuda.stream(st):
                fn(env, o)
            done = torch.cuda.Event(); done.record(st)
            cur.wait_event(done)

    def reset(self):
        self._fan(lambda env, o: env.reset())
        return self.obs

    def step(self, actions_soa: torch.Tensor):
        """actions_soa: float32 [A_max, LD] (rows >= A of a segment ignored).  Returns the padded
        (obs view [LD, S_max], reward [LD], flags [LD])."""
        assert actions_soa.shape == (self.A_max, self.ld) and actions_soa.stride(1) == 1

        def f(env, o):
            env.step(actions_soa[:env.action_dim, o:o + env.batch], layout="soa")
            self.reward[o:o + env.batch].copy_(env.reward)
            self.flags[o:o + env.batch].copy_(env.flags)
        self._fan(f)
        return self.obs, self.reward, self.flags

    def rollout(self, n_steps: int, action_ring: torch.Tensor, reward_out=None, flags_out=None, obs_out=None):
        """action_ring: float32 [R, A_max, LD]; optional reward/flags outputs [n_steps, LD]; optional obs_out float32
        [n_steps, S_max, LD]: the observation every env.step returns, in the padded SoA layout of the state matrix
        (rows >= S of a segment are left as they are)."""
        assert action_ring.shape[1:] == (self.A_max, self.ld)
        if obs_out is not None:
            assert reward_out is not None and obs_out.dtype == torch.float32 and obs_out.dim() == 3
            assert obs_out.shape[0] >= n_steps and obs_out.shape[1] >= self.S_max and obs_out.shape[2] == self.ld
            assert obs_out.stride(2) == 1 and obs_out.stride(1) == self.ld
        if self.fused:
            assert action_ring.dtype == torch.float32 and action_ring.stride(2) == 1 and action_ring.stride(1) == self.ld
            assert (reward_out is None) == (flags_out is None)
            os_ = 0
            if reward_out is not None:
                assert reward_out.dtype == torch.float32 and flags_out.dtype == torch.int32
                assert reward_out.shape[-1] == self.ld and flags_out.shape[-1] == self.ld
                os_ = reward_out.stride(0) if reward_out.dim() == 2 else 0
                assert (flags_out.stride(0) if flags_out.dim() == 2 else 0) == os_
            L = self.envs[0]._L
            with torch.cuda.device(self.envs[0]._dev_index):
                _lib.check(L.nig_rollout_mixed_obs(self._hs, self._offs, len(self.envs), int(n_steps),
                                                   C.c_void_p(action_ring.data_ptr()), self.ld, action_ring.stride(0),
                                                   action_ring.shape[0], _ptr(reward_out), _ptr(flags_out), os_,
                                                   _ptr(obs_out), self.ld, 0 if obs_out is None else obs_out.stride(0),
                                                   self.envs[0]._stream()))
            return

        def f(env, o):
            env.rollout(n_steps, action_ring[:, :env.action_dim, o:o + env.batch],
                        None if reward_out is None else reward_out[:, o:o + env.batch],
                        None if flags_out is None else flags_out[:, o:o + env.batch],
                        None if obs_out is None else obs_out[:, :env.state_dim, o:o + env.batch])
        self._fan(f)

    def fill_actions(self, t: int, out: torch.Tensor):
        self._fan(lambda env, o: env.fill_actions(t, out[:env.action_dim, o:o + env.batch]))
        return out

    def reduce_tally(self):
        return [env.reduce_tally() for env in self.envs]

    def close(self):
        for e in self.envs:
            e.close()
