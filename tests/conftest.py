import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

ENV_NAME = {"cr": "ChemicalReactor-v0", "pg": "PowerGrid-v0", "ra": "RobotAssembly-v0"}
KEYS = ["cr", "pg", "ra"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(key, fam):
    return dict(np.load(os.path.join(GOLDEN, f"{key}_{fam}.npz")))


def golden_meta():
    with open(os.path.join(GOLDEN, "META.json")) as f:
        return json.load(f)


def result_of(g4):
    return json.loads(bytes(g4["result_json"]).decode())


def rel_err(a, b, floor=1e-6):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    same_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
    with np.errstate(invalid="ignore"):
        e = np.abs(a - b) / np.maximum(np.abs(b), floor)
    e[both_nan | same_inf] = 0.0
    return e


# Tolerance the build commits to (SURVEY.md A.6 / BASELINE north_star): float32 outputs within
# 1e-5 relative (abs floor 1e-6) of the reference; integer outputs exact.
RTOL = 1e-5


class StubAgent:
    """Same elementwise policy as oracle/gen_golden.py's StubAgent, rebuilt from the numbers
    stored in the g4 fixture: a_j = clip((obs[i_j] - c_j) * k_j, -1, 1) in float32."""
    is_trained = True

    def __init__(self, g4):
        self.idx = np.asarray(g4["agent_idx"], dtype=np.int64)
        self.ref = np.asarray(g4["agent_ref"], dtype=np.float32)
        self.gain = np.asarray(g4["agent_gain"], dtype=np.float32)

    def predict(self, obs, deterministic=True):
        obs = np.asarray(obs, dtype=np.float32)
        return np.clip((obs[:, self.idx] - self.ref) * self.gain, np.float32(-1), np.float32(1)).astype(np.float32)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O
