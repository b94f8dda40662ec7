"""Multi-GPU sharding of a batch of env instances and the one exchange the path has.

Env instances are independent (base.py:157-213 touches only `self`), so a job of
`total` lanes is cut into contiguous lane ranges, one per rank (one process per GPU),
with the counter-based RNG keyed by the GLOBAL lane index: results do not depend on the
number of ranks.  The only communication is at the end of a rollout: every rank's
partial tally vector (NIG_T_ROWS doubles, ~100 bytes) is all-gathered (RCCL over xGMI
when the backend is "nccl"; gloo in the CPU tests) and combined in rank order on every
rank -- bit-reproducible sums, exact integer counts, min/max handled in the same pass.
"""
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib


# True: all_gather_partials calls the backend's all-gather even in a one-rank group (bench.py NIG_BENCH_FORCE_PG=1 and
# tests/test_gpu_rccl_one_rank.py: the RCCL call path executed on a one-GPU box).  Default: a one-rank job has nothing to gather.
ALWAYS_COLLECTIVE = False


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of `total` lanes: (first global lane, lane count) of `rank`."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(int(total), int(world))
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def combine_partials(partials: torch.Tensor) -> torch.Tensor:
    """[world, T_ROWS] float64 -> [T_ROWS]; fixed rank order (sequential fp64 adds)."""
    p = partials.to(torch.float64)
    out = torch.zeros(_lib.T_ROWS, dtype=torch.float64, device=p.device)
    for r in range(p.shape[0]):
        out = out + p[r]
    out[_lib.T_RET_MIN] = p[:, _lib.T_RET_MIN].min()
    out[_lib.T_RET_MAX] = p[:, _lib.T_RET_MAX].max()
    return out


def all_gather_partials(partial: torch.Tensor, group=None) -> torch.Tensor:
    """Every rank's partial vector, [world, T_ROWS], in rank order (identical on every rank)."""
    import torch.distributed as dist
    flat = partial.contiguous().reshape(-1)
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not ALWAYS_COLLECTIVE):
        return flat.reshape(1, -1)
    world = dist.get_world_size(group)
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat, group=group)          # ncclAllGather (RCCL) / gloo allgather
    return torch.stack(gathered)


def all_reduce_partial(partial: torch.Tensor, group=None) -> torch.Tensor:
    """All-gather every rank's partial vector and combine identically everywhere."""
    return combine_partials(all_gather_partials(partial, group))


def metrics_from_partial(p, n_episodes: Optional[int] = None) -> Dict[str, float]:
    """The 13-key result of evaluate_with_safety (utils.py:128-152) from a combined tally."""
    p = np.asarray(torch.as_tensor(p).cpu(), dtype=np.float64)
    n = float(p[_lib.T_EPISODES])
    if n_episodes is None:
        n_episodes = int(n)
    if n <= 0:
        raise ValueError("no finished episodes in the tally")
    rmean = p[_lib.T_RET_SUM] / n
    lmean = p[_lib.T_LEN_SUM] / n
    rvar = max(p[_lib.T_RET_SQ] / n - rmean * rmean, 0.0)     # population std, np.std default
    lvar = max(p[_lib.T_LEN_SQ] / n - lmean * lmean, 0.0)
    viol = int(round(p[_lib.T_VIOL]))
    succ = int(round(p[_lib.T_SUCCESS]))
    return {
        "return_mean": float(rmean), "return_std": float(np.sqrt(rvar)),
        "return_min": float(p[_lib.T_RET_MIN]), "return_max": float(p[_lib.T_RET_MAX]),
        "length_mean": float(lmean), "length_std": float(np.sqrt(lvar)),
        "safety_violations": viol, "safety_violations_per_episode": viol / n_episodes,
        "critical_violations": int(round(p[_lib.T_CRIT])),
        "emergency_shutdowns": int(round(p[_lib.T_SHUTDOWN])),
        # mean over steps of constraints_satisfied / total_constraints (utils.py:109,144-147); the tally carries
        # both sums, so any number of enabled constraints (4 for AdvancedChemicalReactor, fewer after
        # remove_safety_constraint, 0 -> every step's rate is 1.0) comes out right
        "constraint_satisfaction_rate": float(p[_lib.T_SATISFIED] / p[_lib.T_CONSTRAINTS]) if p[_lib.T_CONSTRAINTS] > 0 else 1.0,
        "successful_episodes": succ, "success_rate": succ / n_episodes,
    }
