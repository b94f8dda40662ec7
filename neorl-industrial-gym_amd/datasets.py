"""env.get_dataset(quality) for the single-env classes.

What upstream does (chemical_reactor.py:324-420, power_grid.py:194-249, robot_assembly.py:246-308):
roll a per-quality behaviour policy through env.step and keep (observation acted on, action, reward,
end-of-episode flag).  Here that is ONE collector driven by a per-(env, quality) behaviour record:
a state-feedback law, its exploration noise and an optional epsilon-gate with a uniform fallback.
Two things of the reference are part of the contract and kept exactly, because seeded runs are
compared with the reference's own output (tests/golden/datasets.npz):

* the ORDER of the draws from the global np.random stream (gate first where there is one, then one
  normal per noisy action dimension in dimension order, or one uniform vector), and
* the DTYPE of the action vector handed to env.step -- float32 for the ChemicalReactor expert
  (three np.float32 elements), float64 everywhere else (a Python float or a float64 array takes
  part) -- since env.step follows NumPy's promotion from the action's dtype (include/nig.h
  nig_step / nig_step64).  Feedback terms are evaluated in the precision NumPy would use for them.
"""
from dataclasses import dataclass
from typing import Callable, Optional, Tuple

import numpy as np

f32 = np.float32


@dataclass
class Behaviour:
    """How one dataset quality of one env acts and how much of it is collected."""
    episodes: int
    step_cap: int
    law: Optional[Callable] = None            # obs -> action vector (None: no feedback branch)
    gate: Optional[float] = None              # P(feedback branch); None = no gate draw at all
    fallback: Optional[Tuple[float, float]] = None   # uniform(lo, hi, A) when the gate fails / no law
    clip: Optional[float] = None              # np.clip(action, -clip, clip) before env.step (and before storing)
    stop_flag: str = "terminated"             # what goes into 'terminals': 'terminated' or 'done'
    timeouts: bool = False                    # add an all-False 'timeouts' array (ChemicalReactor)


# ---------------------------------------------------------------------------------------------------
# ChemicalReactor-v0 (chemical_reactor.py:333-393): proportional pull of temperature (and level) with
# Gaussian exploration per action dimension; the non-expert qualities gate it against uniform actions.
# ---------------------------------------------------------------------------------------------------
_CR_SHAPE = {"expert": (100, 400, 0.1), "medium": (200, 350, 0.3), "mixed": (300, 300, 0.5)}   # else: (500, 200, 1.0)


def _cr_law(env, level: float, expert: bool):
    """Rows of (observation index, setpoint, span, gain) or None, and the noise scale per action dimension."""
    T = (0, env.temp_target, 50)
    rows = ([(T, f32(-0.5)), (T, f32(0.3)), ((10, 55, 50), f32(-0.2))] if expert else [(T, f32(-0.2)), None, None])
    scales = [level * 0.1] * 3 if expert else [level * 0.3, level * 0.5, level * 0.3]

    def law(obs):
        out = []
        for row, sd in zip(rows, scales):
            eps = np.random.normal(0, sd)                       # Python float
            if row is None:
                out.append(eps)                                 # stays a Python float: the vector becomes float64
            else:
                (i, target, span), gain = row
                out.append(((obs[i] - target) / span) * gain + eps)    # np.float32 throughout (weak Python scalars)
        return np.array(out)
    return law


def _cr_behaviour(env, quality) -> Behaviour:
    episodes, cap, level = _CR_SHAPE.get(quality, (500, 200, 1.0))      # any other string collects 'random' (:345-347)
    expert = quality == "expert"
    return Behaviour(episodes, cap, law=_cr_law(env, level, expert), gate=None if expert else 1 - level,
                     fallback=(-1, 1), clip=1, stop_flag="done", timeouts=True)


# ---------------------------------------------------------------------------------------------------
# PowerGrid-v0 (power_grid.py:197-233) and RobotAssembly-v0 (robot_assembly.py:248-292)
# ---------------------------------------------------------------------------------------------------
def _pg_behaviour(env, quality) -> Behaviour:
    episodes = {"expert": 100, "medium": 150, "mixed": 200, "random": 80}[quality]       # n_samples // 1000
    A = env.action_dim
    if quality == "random":
        return Behaviour(episodes, 1000, fallback=(-5, 5))
    if quality == "expert":                   # frequency droop plus an even share of the load/generation gap
        def law(obs):
            gap = np.sum(obs[17:25]) - np.sum(obs[9:17])
            u = (-0.5 * obs[0]) * np.ones(A)
            u += 0.1 * gap / A
            return u
        return Behaviour(episodes, 1000, law=law)
    return Behaviour(episodes, 1000, law=lambda obs: (-0.3 * obs[0]) * np.ones(A), gate=0.6, fallback=(-3, 3))


def _ra_behaviour(env, quality) -> Behaviour:
    episodes = {"expert": 120, "medium": 180, "mixed": 250, "random": 100}[quality]
    if quality == "random":
        return Behaviour(episodes, 1000, fallback=(-1, 1), clip=2.0)
    if quality == "expert":                   # Cartesian pull on the first three joints, damping on the last four
        law = lambda obs: np.concatenate([2.0 * (env.target_position - obs[0:3]), -0.1 * obs[10:14]])     # noqa: E731
        return Behaviour(episodes, 1000, law=law, clip=2.0)
    law = lambda obs: np.concatenate([1.0 * (env.target_position - obs[0:3]), np.random.uniform(-0.5, 0.5, 4)])   # noqa: E731
    return Behaviour(episodes, 1000, law=law, gate=0.7, fallback=(-0.8, 0.8), clip=2.0)


_BEHAVIOURS = {"ChemicalReactor-v0": _cr_behaviour, "PowerGrid-v0": _pg_behaviour, "RobotAssembly-v0": _ra_behaviour}


def _act(b: Behaviour, obs, A: int, gate_draw):
    """One behaviour action; draws from the global stream in the reference's order."""
    if b.law is not None and (b.gate is None or gate_draw() < b.gate):
        u = b.law(obs)
    else:
        u = np.random.uniform(b.fallback[0], b.fallback[1], A)
    return u if b.clip is None else np.clip(u, -b.clip, b.clip)


def collect(env, b: Behaviour):
    """Roll the behaviour through env.step; a transition is (observation acted on, action as handed to
    step, reward, stop flag).  An episode ends on done or at the step cap."""
    # ChemicalReactor draws its gate with np.random.random, the other two with np.random.rand
    gate_draw = np.random.random if b.timeouts else np.random.rand
    rows = {"observations": [], "actions": [], "rewards": [], "terminals": []}
    for _ in range(b.episodes):
        obs, _ = env.reset()
        for _ in range(b.step_cap):
            action = _act(b, obs, env.action_dim, gate_draw)
            nxt, reward, terminated, truncated, _ = env.step(action)
            finished = terminated or truncated
            rows["observations"].append(obs)
            rows["actions"].append(action)
            rows["rewards"].append(reward)
            rows["terminals"].append(finished if b.stop_flag == "done" else terminated)
            if finished:
                break
            obs = nxt
    out = {"observations": np.array(rows["observations"], dtype=np.float32),
           "actions": np.array(rows["actions"], dtype=np.float32),
           "rewards": np.array(rows["rewards"], dtype=np.float32),
           "terminals": np.array(rows["terminals"], dtype=bool)}
    if b.timeouts:
        out["timeouts"] = np.zeros(len(rows["terminals"]), dtype=bool)
    return out


def get_dataset(env, quality="mixed"):
    kind = type(env).ENV_ID
    if kind not in _BEHAVIOURS:
        raise KeyError(kind)
    if kind != "ChemicalReactor-v0" and quality not in ("expert", "medium", "mixed", "random"):
        raise KeyError(quality)              # upstream: dict lookup
    return collect(env, _BEHAVIOURS[kind](env, quality))
