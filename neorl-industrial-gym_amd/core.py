"""Boundary types of the hot path.

Field names and meanings follow the reference's `core/types.py` (SafetyConstraint :56-64,
SafetyMetrics :67-103, DatasetQuality :48-54) because downstream agents and harnesses read them;
the implementations are this package's own.  On the device a step's SafetyMetrics is two small
integers inside the per-lane flag word (include/nig.h NIG_FLAG_*); these classes are what the
Python surface rebuilds from it on demand.
"""
import enum
from dataclasses import dataclass, field
from typing import Any, Callable, Dict, Sequence, Tuple

import numpy as np


class DatasetQuality(enum.Enum):
    EXPERT = "expert"
    MEDIUM = "medium"
    MIXED = "mixed"
    RANDOM = "random"


@dataclass
class SafetyConstraint:
    """A named predicate over (state, action): check_fn returns True while the constraint holds.
    `penalty` is added to the reward of every violating step; a `critical` violation also ends
    the episode with an extra -1000 (reference step template, environments/base.py:179-198)."""
    name: str
    check_fn: Callable[[Any, Any], bool]
    penalty: float
    critical: bool = False
    description: str = ""


@dataclass
class SafetyMetrics:
    """Outcome of evaluating every constraint on one step's pre-state."""
    constraints_satisfied: int
    total_constraints: int
    violation_count: int
    critical_violations: int
    safety_score: float
    adaptive_threshold: float = 0.95
    confidence_interval: Tuple[float, float] = (0.0, 1.0)
    violation_severity: Dict[str, float] = field(default=None)

    def __post_init__(self):
        self.violation_severity = {} if self.violation_severity is None else self.violation_severity

    @property
    def satisfaction_rate(self) -> float:
        total = self.total_constraints
        return self.constraints_satisfied / total if total else 1.0

    @property
    def adaptive_safety_score(self) -> float:
        lo, hi = self.confidence_interval
        return max(self.safety_score - 0.1 * abs(hi - lo), 0.0)

    def update_adaptive_threshold(self, performance_history: Sequence[float]) -> None:
        recent = np.asarray(performance_history[-10:], dtype=float)
        if recent.size == 10:
            self.adaptive_threshold = float(np.clip(recent.mean() - 2.0 * recent.std(), 0.8, 0.99))


class Box:
    """Stand-in for gymnasium.spaces.Box when gymnasium is not installed: bounds, shape, dtype,
    `sample()` (from the global NumPy stream, like the envs) and `contains()`."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        self.shape = tuple(np.shape(low)) if shape is None else tuple(shape)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()
        self._lo = np.where(np.isfinite(self.low), self.low, -1e6).astype(np.float64)
        self._hi = np.where(np.isfinite(self.high), self.high, 1e6).astype(np.float64)

    def sample(self):
        return np.random.uniform(self._lo, self._hi).astype(self.dtype)

    def contains(self, x) -> bool:
        x = np.asarray(x)
        return x.shape == self.shape and bool((x >= self.low).all() and (x <= self.high).all())

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"


def make_box(low, high, shape, dtype=np.float32):
    """gymnasium's Box when available, the stand-in otherwise."""
    try:
        import gymnasium
        return gymnasium.spaces.Box(low=low, high=high, shape=shape, dtype=dtype)
    except Exception:
        return Box(low, high, shape, dtype)
