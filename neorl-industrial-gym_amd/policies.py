"""Policies that run ON the device inside the fused closed-loop rollout ("nig-policy-v1",
include/nig.h), with a host `predict()` of the same float32 arithmetic so the very same
object also works with the single-env classes and the reference-shaped evaluation loop.

Families (all the reference's non-neural agents):
  * baseline agents of benchmarks/baseline_agents.py:28-114: constant_agent, mpc_agent
    ("MPC" = proportional pull to the origin), pid_agent, random_agent;
  * behaviour policies of get_dataset: behaviour_policy(env_id, quality)
    (chemical_reactor.py:364-393, power_grid.py:216-233, robot_assembly.py:266-290).
The neural actors (agents/networks.py) stay outside: pass any object with
`predict_device(obs_tensor)` / `predict(obs)` to evaluate_with_safety for those.
"""
from typing import Optional, Sequence

import numpy as np

from . import _lib

f32 = np.float32
_DIMS = {"ChemicalReactor-v0": (12, 3), "PowerGrid-v0": (32, 8), "RobotAssembly-v0": (24, 7)}


class DevicePolicy:
    """a = clip(b + W @ obs (+ noise, + epsilon-uniform mixture), lo, hi)  or a PID law."""
    is_trained = True          # evaluate_with_safety's gate (utils.py:69-70)

    def __init__(self, state_dim: int, action_dim: int, kind: int = _lib.POLICY_AFFINE, W=None, b=None, sigma=None,
                 half_range=None, p_uniform: float = 0.0, uniform_range: float = 1.0, clip=(-np.inf, np.inf),
                 kp: float = 0.0, ki: float = 0.0, kd: float = 0.0, setpoint=None):
        assert state_dim <= 32 and action_dim <= 10      # NIG_MAX_STATE_DIM / NIG_MAX_ACTION_DIM (include/nig.h)
        self.state_dim, self.action_dim, self.kind = state_dim, action_dim, kind
        z = lambda: np.zeros(action_dim, dtype=f32)     # noqa: E731
        self.W = np.zeros((action_dim, state_dim), dtype=f32) if W is None else np.asarray(W, dtype=f32)
        self.b = z() if b is None else np.asarray(b, dtype=f32)
        self.sigma = z() if sigma is None else np.asarray(sigma, dtype=f32)
        self.half_range = z() if half_range is None else np.asarray(half_range, dtype=f32)
        self.p_uniform, self.uniform_range = f32(p_uniform), f32(uniform_range)
        self.clip = (f32(clip[0]), f32(clip[1]))
        self.kp, self.ki, self.kd = f32(kp), f32(ki), f32(kd)
        self.setpoint = z() if setpoint is None else np.asarray(setpoint, dtype=f32)
        self._integ, self._eprev = z(), z()
        assert self.W.shape == (action_dim, state_dim)

    @property
    def stochastic(self) -> bool:
        return bool(np.any(self.sigma != 0) or np.any(self.half_range != 0) or self.p_uniform > 0)

    def to_struct(self) -> _lib.Policy:
        P = _lib.Policy()
        P.kind = self.kind
        for j in range(self.action_dim):
            for k in range(self.state_dim):
                P.Wt[k][j] = float(self.W[j, k])
            P.b[j], P.sigma[j], P.half_range[j] = float(self.b[j]), float(self.sigma[j]), float(self.half_range[j])
            P.setpoint[j] = float(self.setpoint[j])
        P.p_uniform, P.uniform_range = float(self.p_uniform), float(self.uniform_range)
        P.clip_lo, P.clip_hi = float(self.clip[0]), float(self.clip[1])
        P.kp, P.ki, P.kd = float(self.kp), float(self.ki), float(self.kd)
        return P

    def predict(self, observations, deterministic: bool = True):
        """Host evaluation of the deterministic part in the device's float32 term order
        (agent.predict contract, agents/base.py:106-141: [n,S] -> [n,A])."""
        obs = np.asarray(observations, dtype=f32)
        single = obs.ndim == 1
        obs = np.atleast_2d(obs)
        out = np.zeros((obs.shape[0], self.action_dim), dtype=f32)
        for i, o in enumerate(obs):
            if self.kind == _lib.POLICY_PID:
                e = self.setpoint - o[:self.action_dim]
                self._integ = (self._integ + e).astype(f32)
                u = ((self.kp * e + self.ki * self._integ) + self.kd * (e - self._eprev)).astype(f32)
                self._eprev = e
            else:
                u = self.b.copy()
                for k in range(self.state_dim):
                    if np.any(self.W[:, k] != 0):
                        u = (u + self.W[:, k] * o[k]).astype(f32)
            out[i] = np.minimum(np.maximum(u, self.clip[0]), self.clip[1])
        return out[0] if single else out


def constant_agent(state_dim: int, action_dim: int, constant_action: Optional[Sequence[float]] = None) -> DevicePolicy:
    """ConstantAgent, baseline_agents.py:102-113"""
    return DevicePolicy(state_dim, action_dim, b=np.zeros(action_dim) if constant_action is None else constant_action)


def mpc_agent(state_dim: int, action_dim: int) -> DevicePolicy:
    """MPC_Agent, baseline_agents.py:83-99: 0.5 * (0 - state[:A]) clipped to [-1, 1]."""
    W = np.zeros((action_dim, state_dim), dtype=f32)
    for j in range(action_dim):
        W[j, j] = -0.5
    return DevicePolicy(state_dim, action_dim, W=W, clip=(-1.0, 1.0))


def pid_agent(state_dim: int, action_dim: int, kp=1.0, ki=0.1, kd=0.01, setpoint=None) -> DevicePolicy:
    """PIDControllerAgent, baseline_agents.py:44-80 (integral never reset, as upstream)."""
    return DevicePolicy(state_dim, action_dim, kind=_lib.POLICY_PID, kp=kp, ki=ki, kd=kd, setpoint=setpoint,
                        clip=(-1.0, 1.0))


def random_agent(state_dim: int, action_dim: int, action_low: float = -1.0, action_high: float = 1.0) -> DevicePolicy:
    """RandomAgent, baseline_agents.py:28-41 (symmetric range)."""
    assert action_low == -action_high, "the device mixture draws from U(-r, r)"
    return DevicePolicy(state_dim, action_dim, p_uniform=1.0, uniform_range=action_high)


# (episodes, step cap) of each get_dataset quality
from .spec_plants import PLANTS as _PLANT_LIST  # noqa: E402

_SPEC_PLANTS = {p["name"]: p for p in _PLANT_LIST}
for _p in _PLANT_LIST:
    _DIMS[_p["name"]] = (len(_p["y"]) + len(_p["act"]) + 3, len(_p["act"]))

DATASET_SHAPE = {
    "ChemicalReactor-v0": {"expert": (100, 400), "medium": (200, 350), "mixed": (300, 300), "random": (500, 200)},
    "PowerGrid-v0": {"expert": (100, 1000), "medium": (150, 1000), "mixed": (200, 1000), "random": (80, 1000)},
    "RobotAssembly-v0": {"expert": (120, 1000), "medium": (180, 1000), "mixed": (250, 1000), "random": (100, 1000)},
}
# README-only envs: (episodes, step cap) chosen here, nothing upstream to follow
for _p in _PLANT_LIST:
    DATASET_SHAPE[_p["name"]] = {"expert": (100, 500), "medium": (150, 500), "mixed": (200, 500), "random": (100, 500)}


def behaviour_policy(env_id: str, quality: str) -> DevicePolicy:
    """The data-collection policy of env.get_dataset(quality) as a device policy."""
    S, A = _DIMS[env_id]
    W = np.zeros((A, S), dtype=np.float64)
    b = np.zeros(A)
    if env_id == "ChemicalReactor-v0":                       # chemical_reactor.py:333-393
        nl = {"expert": 0.1, "medium": 0.3, "mixed": 0.5}.get(quality, 1.0)
        if quality == "expert":
            W[0, 0], b[0] = -0.5 / 50, 0.5 * 320.0 / 50      # -temp_error*0.5, temp_error=(T-320)/50
            W[1, 0], b[1] = 0.3 / 50, -0.3 * 320.0 / 50      #  temp_error*0.3
            W[2, 10], b[2] = -0.2 / 50, 0.2 * 55.0 / 50      # -level_error*0.2
            return DevicePolicy(S, A, W=W, b=b, sigma=[nl * 0.1] * 3, clip=(-1.0, 1.0))
        if nl >= 1.0:
            # "random": np.random.random() < (1 - 1.0) never holds (:380), so every action is uniform(-1, 1, 3) (:389):
            # no feedback term at all (found by tests/golden/behaviour_laws.npz: the reference's law with its draws
            # patched to zero is identically 0 here)
            return DevicePolicy(S, A, p_uniform=1.0, uniform_range=1.0, clip=(-1.0, 1.0))
        W[0, 0], b[0] = -0.2 / 50, 0.2 * 320.0 / 50
        return DevicePolicy(S, A, W=W, b=b, sigma=[nl * 0.3, nl * 0.5, nl * 0.3], p_uniform=nl, uniform_range=1.0,
                            clip=(-1.0, 1.0))
    if env_id == "PowerGrid-v0":                             # power_grid.py:216-233 (no policy-side clip)
        if quality == "expert":
            W[:, 0] = -0.5
            W[:, 17:25] = 0.1 / A
            W[:, 9:17] = -0.1 / A
            return DevicePolicy(S, A, W=W)
        if quality == "random":
            return DevicePolicy(S, A, p_uniform=1.0, uniform_range=5.0)
        W[:, 0] = -0.3
        return DevicePolicy(S, A, W=W, p_uniform=0.4, uniform_range=3.0)
    if env_id == "RobotAssembly-v0":                         # robot_assembly.py:266-292, np.clip(action, -2, 2)
        tgt = [0.3, 0.0, 0.4]
        if quality == "expert":
            for j in range(3):
                W[j, j], b[j] = -2.0, 2.0 * tgt[j]
            for i in range(4):
                W[3 + i, 10 + i] = -0.1
            return DevicePolicy(S, A, W=W, b=b, clip=(-2.0, 2.0))
        if quality == "random":
            return DevicePolicy(S, A, p_uniform=1.0, uniform_range=1.0, clip=(-2.0, 2.0))
        for j in range(3):
            W[j, j], b[j] = -1.0, tgt[j]
        return DevicePolicy(S, A, W=W, b=b, half_range=[0, 0, 0, 0.5, 0.5, 0.5, 0.5], p_uniform=0.3,
                            uniform_range=0.8, clip=(-2.0, 2.0))
    if env_id in _SPEC_PLANTS:                               # build-specified plants: proportional control on the table
        return _spec_behaviour(_SPEC_PLANTS[env_id], quality)
    raise ValueError(env_id)


def _spec_behaviour(P, quality: str) -> DevicePolicy:
    """Data-collection policies of the four README-only envs (no upstream get_dataset exists for them):
    a proportional controller read off the plant table -- actuator j is driven against the weighted
    setpoint errors of the variables it acts on, u_j = -kp * sum_i w_i sign(G_ij) (y_i - sp_i) / span_i --
    plus Gaussian exploration noise (expert), a weaker gain with an epsilon-uniform mixture (medium,
    mixed), or uniform actions (random)."""
    ys, acts = P["y"], P["act"]
    NP_, A = len(ys), len(acts)
    S = NP_ + A + 3
    if quality == "random":
        return DevicePolicy(S, A, p_uniform=1.0, uniform_range=1.0, clip=(-1.0, 1.0))
    kp, sigma, eps = {"expert": (2.0, 0.05, 0.0), "medium": (1.0, 0.15, 0.1)}.get(quality, (0.5, 0.3, 0.3))
    W = np.zeros((A, S))
    b = np.zeros(A)
    aidx = {a["name"]: j for j, a in enumerate(acts)}
    for i, y in enumerate(ys):
        if y["w"] == 0.0:
            continue
        span = max(abs(y["hi"] - y["lo"]), 1e-6)
        for name, g in y["gains"].items():
            c = -kp * np.sign(g) * min(1.0, y["w"] * 10.0) / span * 10.0
            W[aidx[name], i] += c
            b[aidx[name]] -= c * y["sp"]
    return DevicePolicy(S, A, W=W, b=b, sigma=[sigma] * A, p_uniform=eps, uniform_range=1.0, clip=(-1.0, 1.0))


class MLPPolicy:
    """The deterministic actor of the reference's agents -- (S -> 256 -> 256 -> A) ReLU MLP with a
    tanh head (agents/networks.py:47-70,125-144; cql.py:339-343) -- kept on the GPU.

    `weights` = [(W1 [S,H], b1 [H]), (W2 [H,H], b2 [H]), (W3 [H,A], b3 [A])] as exported from the
    agent (Flax Dense kernels are [in, out]).  evaluate_with_safety() calls `predict_device`, so
    observations and actions never leave the device; `predict` is the host form of the same net
    (agent.predict contract, agents/base.py:106-141)."""
    is_trained = True

    def __init__(self, weights, device="cuda:0"):
        import torch
        self.device = torch.device(device)
        self.layers = [(torch.as_tensor(np.asarray(W), dtype=torch.float32, device=self.device).contiguous(),
                        torch.as_tensor(np.asarray(b), dtype=torch.float32, device=self.device).contiguous())
                       for W, b in weights]
        self.state_dim = self.layers[0][0].shape[0]
        self.action_dim = self.layers[-1][0].shape[1]
        self.weights = [(np.asarray(W, dtype=f32), np.asarray(b, dtype=f32)) for W, b in weights]
        # the fused MFMA kernel covers the reference actor shape; anything else goes through torch GEMMs
        self.fusable = (len(self.weights) == 3 and self.weights[0][0].shape[1] == 256
                        and self.weights[1][0].shape == (256, 256))

    def predict_device(self, obs):
        import torch
        x = obs
        for i, (W, b) in enumerate(self.layers):
            x = torch.addmm(b, x, W)
            x = torch.relu(x) if i + 1 < len(self.layers) else torch.tanh(x)
        return x

    def predict(self, observations, deterministic: bool = True):
        import torch
        o = torch.as_tensor(np.asarray(observations, dtype=f32), device=self.device)
        single = o.dim() == 1
        out = self.predict_device(o.reshape(-1, self.state_dim)).cpu().numpy()
        return out[0] if single else out
