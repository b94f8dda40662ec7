"""make / make_batched / evaluate_with_safety -- mirror of neorl_industrial/utils.py."""
from typing import Any, Dict, Optional

import numpy as np
import torch

from . import _lib
from .batched import BatchedIndustrialEnv
from .envs import (AdvancedChemicalReactorEnv, AdvancedPowerGridEnv, ChemicalReactorEnv, HVACControlEnv, PowerGridEnv,
                   RobotAssemblyEnv, SteelAnnealingEnv, SupplyChainEnv, WaterTreatmentEnv)
from .parallel import all_reduce_partial, metrics_from_partial

_REGISTRY = {
    "ChemicalReactor-v0": ChemicalReactorEnv,
    "PowerGrid-v0": PowerGridEnv,
    "RobotAssembly-v0": RobotAssemblyEnv,
    "AdvancedChemicalReactor-v0": AdvancedChemicalReactorEnv,     # utils.py:30-31: registered upstream but
    "AdvancedPowerGrid-v0": AdvancedPowerGridEnv,                 # not instantiable there (candidate rows)
    # README.md:28-32 lists these four; upstream ni.make() raises for them (no implementation).  Build-specified.
    "HVACControl-v0": HVACControlEnv,
    "WaterTreatment-v0": WaterTreatmentEnv,
    "SteelAnnealing-v0": SteelAnnealingEnv,
    "SupplyChain-v0": SupplyChainEnv,
}


def make(env_id: str, **kwargs) -> Any:
    """utils.py:12-39.  Unknown ids raise the reference's ValueError text.  The two 'Advanced*'
    ids are registered as upstream; there they raise TypeError on construction (abstract methods
    missing, SURVEY.md finding 2), here they are built from the source text (candidate rows)."""
    if env_id not in _REGISTRY:
        available = ", ".join(_REGISTRY.keys())
        raise ValueError(f"Unknown environment '{env_id}'. Available: {available}")
    return _REGISTRY[env_id](**kwargs)


def make_batched(env_id: str, batch: int, **kwargs) -> BatchedIndustrialEnv:
    """B independent instances of `env_id` on one GPU (no upstream equivalent: the reference
    has no vectorised env)."""
    return BatchedIndustrialEnv(env_id, batch, **kwargs)


def _predict(agent, obs_dev: torch.Tensor) -> torch.Tensor:
    """agent.predict contract (agents/base.py:106-141): float32 [n,S] -> float32 [n,A].
    An agent exposing `predict_device(obs_tensor)` is called without leaving the GPU."""
    if hasattr(agent, "predict_device"):
        return agent.predict_device(obs_dev)
    obs = obs_dev.contiguous().cpu().numpy()
    act = agent.predict(obs, deterministic=True)
    return torch.as_tensor(np.asarray(act, dtype=np.float32))


def _round_is_over(env, t: int, on_device: bool) -> bool:
    """All lanes finished?  Reading the answer synchronises the stream, so a device-resident
    policy is only asked every 16 steps (finished lanes are frozen; extra steps are no-ops)."""
    if on_device and (t % 16) != 0 and t <= env.max_episode_steps:
        return False
    return bool(env.done.all().item())


def evaluate_with_safety(agent: Any, env: Any, n_episodes: int = 100, record_video: bool = False,
                         render: bool = False, step_noise_fn=None, reset_noise_fn=None) -> Dict[str, Any]:
    """utils.py:42-154.  With a single env this is the reference's loop verbatim in behaviour;
    with a BatchedIndustrialEnv (created with tally=True, autoreset=False) the n_episodes
    episodes run in parallel lanes and the 13 aggregates come from the device tallies."""
    if not hasattr(agent, "is_trained") or not agent.is_trained:
        raise RuntimeError("Agent must be trained before evaluation")
    if isinstance(env, BatchedIndustrialEnv):
        return _evaluate_batched(agent, env, n_episodes, step_noise_fn, reset_noise_fn)

    runs = [_play_episode(agent, env, render) for _ in range(n_episodes)]
    return _summarise(runs, n_episodes)


def _play_episode(agent, env, render):
    """One episode of the reference's evaluation loop (utils.py:80-118): returns
    (return, length, sum of violation_count, sum of critical_violations, shutdown steps,
    per-step satisfaction rates).  `ret += reward` keeps the reward's own type, as upstream."""
    obs, _ = env.reset()
    ret, length, viol, crit, shut, rates = 0.0, 0, 0, 0, 0, []
    finished = False
    while not finished:
        obs, reward, terminated, truncated, info = env.step(agent.predict(obs[None], deterministic=True)[0])
        finished = terminated or truncated
        ret += reward
        length += 1
        sm = info.get("safety_metrics")
        if sm is not None:
            viol += sm.violation_count
            crit += sm.critical_violations
            rates.append(sm.satisfaction_rate)
        shut += 1 if info.get("critical_shutdown", False) else 0
        if render:
            try:
                env.render()
            except Exception:
                pass
    return ret, length, viol, crit, shut, rates


def _summarise(runs, n_episodes):
    """The 13 aggregates of utils.py:128-152 from per-episode tuples."""
    rets = [r[0] for r in runs]
    lens = [r[1] for r in runs]
    rates = [x for r in runs for x in r[5]]
    viol = sum(r[2] for r in runs)
    wins = sum(1 for x in rets if x > 0)
    return {
        "return_mean": np.mean(rets), "return_std": np.std(rets), "return_min": np.min(rets), "return_max": np.max(rets),
        "length_mean": np.mean(lens), "length_std": np.std(lens),
        "safety_violations": viol, "safety_violations_per_episode": viol / n_episodes,
        "critical_violations": sum(r[3] for r in runs), "emergency_shutdowns": sum(r[4] for r in runs),
        "constraint_satisfaction_rate": np.mean(rates) if rates else 1.0,
        "successful_episodes": wins, "success_rate": wins / n_episodes,
    }


def _evaluate_batched(agent, env: BatchedIndustrialEnv, n_episodes: int, step_noise_fn=None, reset_noise_fn=None,
                      group=None, reduce_across_ranks: bool = True) -> Dict[str, Any]:
    """Episodes in parallel lanes.  Rounds of up to B episodes: reset the needed lanes, step
    until every one of them is done (finished lanes are frozen by the kernel), tallies
    accumulate on the device; one reduction (+ all-gather across ranks) at the end.

    step_noise_fn(round, t) / reset_noise_fn(round) may supply recorded draws (parity tests)."""
    if not env.tally_enabled or env.autoreset:
        raise ValueError("batched evaluation needs make_batched(..., tally=True, autoreset=False)")
    B = env.batch
    remaining, rnd = int(n_episodes), 0
    device_policy = hasattr(agent, "to_struct") and step_noise_fn is None and reset_noise_fn is None
    fused_mlp = getattr(agent, "fusable", False) and step_noise_fn is None and reset_noise_fn is None
    if device_policy:
        env.set_policy(agent)
    elif fused_mlp:
        env.set_mlp_policy(agent.weights)
        device_policy = True
    while remaining > 0 and device_policy:
        # the agent runs ON the device: one fused launch plays every episode of the round to its end
        k = min(B, remaining)
        mask = torch.zeros(B, dtype=torch.uint8, device=env.device)
        mask[:k] = 1
        env.ctr.fill_(_lib.CTR_DONE)
        env.reset(mask=mask)
        # the Advanced envs truncate on the step AFTER the cap (advanced_chemical_reactor.py:351 and
        # advanced_power_grid.py:331 test episode_step before its increment): one step more for them
        extra = 1 if env.env_id.startswith("Advanced") else 0
        (env.rollout_mlp if fused_mlp else env.rollout_policy)(env.max_episode_steps + extra)
        remaining -= k
    while remaining > 0:
        k = min(B, remaining)
        mask = torch.zeros(B, dtype=torch.uint8, device=env.device)
        mask[:k] = 1
        env.reset(mask=mask, init_noise=None if reset_noise_fn is None else reset_noise_fn(rnd))
        t = 0
        while True:
            act = _predict(agent, env.obs)
            env.step(act, step_noise=None if step_noise_fn is None else step_noise_fn(rnd, t), layout="aos")
            t += 1
            if _round_is_over(env, t, hasattr(agent, "predict_device")) or t > _lib.MAX_EPISODE_STEPS:
                break
        remaining -= k
        rnd += 1
    partial = env.reduce_tally()
    total = all_reduce_partial(partial, group) if reduce_across_ranks else partial
    import torch.distributed as dist
    n_total = n_episodes
    if reduce_across_ranks and dist.is_available() and dist.is_initialized():
        n_total = int(round(float(total[_lib.T_EPISODES].item())))
    return metrics_from_partial(total, n_total)


def uniform_action_statistics(env_id: str, batch: int, episodes_per_lane: int, device="cuda:0", seed: int = 0xBEEF,
                              plan_steps: int = 250, outputs: str = "min") -> Dict[str, Any]:
    """Episode statistics of the reference's measurement loop (performance_benchmark.py:106-133: a uniform random action per
    step, reset on done) in FAST MODE -- fused rollout launches, in-kernel generator, auto-reset -- over the first
    `episodes_per_lane` episodes of every lane.  A fixed episode count per lane is an unbiased sample ("episodes finished
    inside a window of steps" favours short ones); every step gets a fresh action (slot k of a `plan_steps`-slot ring, refilled
    for every launch).  All reductions run on the device from the per-step flag words and rewards the kernel writes.
    Returns sums over the counted episodes: episodes, steps, viol (base.py:179-183 violation counts), crit, c0..c2 (steps on
    which built-in constraint k failed), term / trunc / shut (how the episodes ended), ret (sum of rewards), hist (episode
    lengths, index = length), launches.  outputs: "min" = reward + flag rows; "full" also writes the observation
    trajectory (the kernel instantiation bench.py's headline times)."""
    L = _lib
    env = make_batched(env_id, batch, device=device, seed=seed, autoreset=True)
    dev, B, K, P = env.device, env.batch, int(episodes_per_lane), int(plan_steps)
    env.reset()
    ring = torch.empty(P, env.action_dim, env.ld, dtype=torch.float32, device=dev)
    rew = torch.empty(P, env.ld, dtype=torch.float32, device=dev)
    fl = torch.empty(P, env.ld, dtype=torch.int32, device=dev)
    traj = torch.empty(P, B, env.state_dim, dtype=torch.float32, device=dev) if outputs == "full" else None
    epcount = torch.zeros(B, dtype=torch.int32, device=dev)
    acc = {k: torch.zeros((), dtype=torch.int64, device=dev)
           for k in ("steps", "viol", "crit", "c0", "c1", "c2", "episodes", "term", "trunc", "shut")}
    ret = torch.zeros((), dtype=torch.float64, device=dev)
    hist = torch.zeros(env.max_episode_steps + 1, dtype=torch.int64, device=dev)
    launches = t = 0
    while True:
        for s in range(P):
            t += 1
            env.fill_actions(7000 + t, ring[s])
        env.rollout(P, ring, rew, fl, traj)
        f = fl[:, :B]
        done = (f & (L.FLAG_TERMINATED | L.FLAG_TRUNCATED)) != 0
        cum = done.cumsum(0, dtype=torch.int32)
        valid = (epcount.unsqueeze(0) + cum - done.to(torch.int32)) < K          # the step belongs to one of the lane's first K episodes
        acc["steps"] += valid.sum()
        acc["viol"] += ((((f >> L.FLAG_NVIOL_SHIFT) & 3) + ((f >> 13) & 1) * 4) * valid).sum()
        acc["crit"] += (((f >> L.FLAG_NCRIT_SHIFT) & 3) * valid).sum()
        for k in range(3):
            acc[f"c{k}"] += (((f >> (L.FLAG_VIOL_SHIFT + k)) & 1) * valid).sum()
        ret += (rew[:, :B].to(torch.float64) * valid).sum()
        dv = done & valid
        acc["episodes"] += dv.sum()
        acc["term"] += (dv & ((f & L.FLAG_TERMINATED) != 0)).sum()
        acc["trunc"] += (dv & ((f & L.FLAG_TRUNCATED) != 0)).sum()
        acc["shut"] += (dv & ((f & L.FLAG_SHUTDOWN) != 0)).sum()
        hist += torch.bincount(((f >> L.FLAG_STEP_SHIFT) & 0xFFFF)[dv].to(torch.int64), minlength=hist.numel())[:hist.numel()]
        epcount += cum[-1]
        launches += 1
        if int(epcount.min().item()) >= K:
            break
        if launches * P > K * env.max_episode_steps + P:
            raise RuntimeError("a lane did not finish its episodes within episodes_per_lane x max_episode_steps steps")
    out: Dict[str, Any] = {k: int(v.item()) for k, v in acc.items()}
    out.update(ret=float(ret.item()), hist=hist.cpu().numpy(), launches=launches, batch=B, episodes_per_lane=K)
    env.close()
    return out
