"""TEST INFRASTRUCTURE ONLY -- loader for the *reference's own* NumPy hot path.

This container only: /root/reference does not exist on the GPU box, so nothing in
`tests -m gpu`, `__graft_entry__.smoke()` or `bench.py` may import this module.
It is used by `oracle/gen_golden.py` (fixture generation) and by the not-gpu test
that cross-checks the C restatement against the live reference when it is present.

The reference package's root `__init__` pulls jax/flax/optax (absent here), but its
env hot path is NumPy-only (SURVEY.md finding 1).  We therefore
  * pre-seed `sys.modules` with a minimal `gymnasium` (Env.reset no-op, spaces.Box)
    and a dummy `jax.numpy` (only used for an isinstance check, base.py:163), and
  * register an empty shell package `neorl_industrial` whose __path__ points at the
    reference sources, so the sub-modules import unmodified.
Nothing from /root/reference is copied; its code runs in place.
"""
import importlib
import os
import sys
import types

import numpy as np

REF_SRC = "/root/reference/src/neorl_industrial"


def reference_available() -> bool:
    return os.path.isdir(REF_SRC)


def load_reference():
    """Return the reference's `neorl_industrial.utils` module (make, evaluate_with_safety)."""
    if not reference_available():
        raise RuntimeError("reference sources not present at " + REF_SRC)
    sys.dont_write_bytecode = True  # never write .pyc into the read-only reference
    if "neorl_industrial.utils" in sys.modules and getattr(
            sys.modules["neorl_industrial"], "_nig_shell", False):
        return sys.modules["neorl_industrial.utils"]

    gym = types.ModuleType("gymnasium")
    spaces = types.ModuleType("gymnasium.spaces")

    class Env:  # gymnasium.Env.reset only seeds self.np_random; the envs use global np.random
        def reset(self, *, seed=None, options=None):
            pass

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.dtype = np.dtype(dtype)
            self.shape = tuple(shape) if shape is not None else np.shape(low)
            self.low = np.full(self.shape, low, dtype=dtype)
            self.high = np.full(self.shape, high, dtype=dtype)

    spaces.Box = Box
    gym.Env = Env
    gym.spaces = spaces
    jax = types.ModuleType("jax")
    jnp = types.ModuleType("jax.numpy")
    jnp.ndarray = type("_Never", (), {})
    jax.numpy = jnp
    sys.modules.update({"gymnasium": gym, "gymnasium.spaces": spaces,
                        "jax": jax, "jax.numpy": jnp})
    root = types.ModuleType("neorl_industrial")
    root.__path__ = [REF_SRC]
    root._nig_shell = True
    sys.modules["neorl_industrial"] = root
    return importlib.import_module("neorl_industrial.utils")


class NoiseTap:
    """Route the reference's global `np.random.normal/uniform` through a private
    Generator and record every value handed back, in call order (fp64).

    `forced` lets a caller replay a prescribed list of draws instead (teacher forcing).
    """

    def __init__(self, seed: int):
        self.gen = np.random.Generator(np.random.PCG64(seed))
        self.log = []
        self.forced = None
        self._saved = None

    # -- draw hooks -------------------------------------------------------
    def _emit(self, vals):
        vals = np.asarray(vals, dtype=np.float64)
        if self.forced is not None:
            n = vals.size
            take = np.asarray(self.forced[:n], dtype=np.float64).reshape(vals.shape)
            assert take.size == n, "forced noise exhausted"
            self.forced = self.forced[n:]
            vals = take
        self.log.extend(np.ravel(vals).tolist())
        return vals

    def normal(self, loc=0.0, scale=1.0, size=None):
        v = self._emit(self.gen.normal(loc, scale, size))
        return float(v) if size is None and v.ndim == 0 else v

    def uniform(self, low=0.0, high=1.0, size=None):
        v = self._emit(self.gen.uniform(low, high, size))
        return float(v) if size is None and v.ndim == 0 else v

    # -- lifecycle --------------------------------------------------------
    def __enter__(self):
        self._saved = (np.random.normal, np.random.uniform)
        np.random.normal = self.normal
        np.random.uniform = self.uniform
        return self

    def __exit__(self, *exc):
        np.random.normal, np.random.uniform = self._saved

    def take(self):
        out = np.array(self.log, dtype=np.float64)
        self.log = []
        return out
