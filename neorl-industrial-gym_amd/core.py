"""Types at the boundary, mirroring neorl_industrial/core/types.py:48-103 (names and
field meaning kept so downstream agents / harnesses are unchanged)."""
from dataclasses import dataclass
from enum import Enum
from typing import Any, Callable, Dict, List, Tuple

import numpy as np


class DatasetQuality(Enum):
    """core/types.py:48-54"""
    EXPERT = "expert"
    MEDIUM = "medium"
    MIXED = "mixed"
    RANDOM = "random"


@dataclass
class SafetyConstraint:
    """core/types.py:56-64.  check_fn(state, action) -> bool (True = satisfied)."""
    name: str
    check_fn: Callable[[Any, Any], bool]
    penalty: float
    critical: bool = False
    description: str = ""


@dataclass
class SafetyMetrics:
    """core/types.py:67-103"""
    constraints_satisfied: int
    total_constraints: int
    violation_count: int
    critical_violations: int
    safety_score: float
    adaptive_threshold: float = 0.95
    confidence_interval: Tuple[float, float] = (0.0, 1.0)
    violation_severity: Dict[str, float] = None

    def __post_init__(self):
        if self.violation_severity is None:
            self.violation_severity = {}

    @property
    def satisfaction_rate(self) -> float:
        if self.total_constraints == 0:
            return 1.0
        return self.constraints_satisfied / self.total_constraints

    @property
    def adaptive_safety_score(self) -> float:
        base_score = self.safety_score
        confidence_penalty = abs(self.confidence_interval[1] - self.confidence_interval[0]) * 0.1
        return max(0.0, base_score - confidence_penalty)

    def update_adaptive_threshold(self, performance_history: List[float]) -> None:
        if len(performance_history) >= 10:
            mean_perf = np.mean(performance_history[-10:])
            std_perf = np.std(performance_history[-10:])
            self.adaptive_threshold = max(0.8, min(0.99, mean_perf - 2 * std_perf))


class Box:
    """Minimal stand-in for gymnasium.spaces.Box (base.py:60-72) used when gymnasium is
    not installed: low/high/shape/dtype, sample(), contains()."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        self.shape = tuple(shape) if shape is not None else np.shape(low)
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1e6)
        hi = np.where(np.isfinite(self.high), self.high, 1e6)
        return np.random.uniform(lo, hi).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"


def make_box(low, high, shape, dtype=np.float32):
    try:
        import gymnasium  # noqa: F401
        return gymnasium.spaces.Box(low=low, high=high, shape=shape, dtype=dtype)
    except Exception:
        return Box(low, high, shape, dtype)
