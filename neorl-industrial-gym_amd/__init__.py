"""neorl-industrial-gym on MI355X: the IndustrialEnv.step() hot path as HIP kernels.

Usage mirrors the reference package (`import neorl_industrial as ni`):

    import neorl_industrial_gym_amd as ni
    env = ni.make('ChemicalReactor-v0')                     # single env, reference surface
    benv = ni.make_batched('ChemicalReactor-v0', 65536)     # 65536 lanes, one kernel per step
    metrics = ni.evaluate_with_safety(agent, env, n_episodes=100)

Everything numerical runs in libnig.so (csrc/, C ABI in include/nig.h).  There is no CPU
fallback: importing the package without the built extension raises ImportError.
"""
from . import _lib
from .core import DatasetQuality, SafetyConstraint, SafetyMetrics

_lib.lib()   # fail loudly here if libnig.so is missing

from .batched import BatchedIndustrialEnv, MixedBatchedEnv, StepInfo  # noqa: E402
from .envs import (AdvancedChemicalReactorEnv, AdvancedPowerGridEnv, ChemicalReactorEnv, HVACControlEnv,  # noqa: E402
                   IndustrialEnv, PowerGridEnv, RobotAssemblyEnv, SteelAnnealingEnv, SupplyChainEnv, WaterTreatmentEnv)
from .policies import (DevicePolicy, MLPPolicy, behaviour_policy, constant_agent, mpc_agent,  # noqa: E402
                       pid_agent, random_agent)
from .utils import evaluate_with_safety, make, make_batched, uniform_action_statistics  # noqa: E402

def tune(split_blocks=None, wide_min_blocks=None):
    """Process-wide kernel-selection knobs of libnig (include/nig.h nig_tune); results never depend on them.
    split_blocks: largest batch, in 256-lane blocks, that rollout() runs in the three-wave form (0 = never).
    wide_min_blocks: smallest batch, in 512-lane blocks, that rollout() runs in the wide form (PowerGrid).
    -1 removes an explicit setting (back to the per-device defaults: split_blocks = the device's compute-unit count,
    wide_min_blocks = 1.5 x the compute units + 1 -- the first batch that no longer fits one round of the 256-lane form).
    Returns the current settings."""
    L = _lib.lib()
    if split_blocks is not None:
        _lib.check(L.nig_tune(_lib.TUNE_SPLIT_BLOCKS, int(split_blocks)))
    if wide_min_blocks is not None:
        _lib.check(L.nig_tune(_lib.TUNE_WIDE_MIN_BLOCKS, int(wide_min_blocks)))
    return {"split_blocks": int(L.nig_tune_get(_lib.TUNE_SPLIT_BLOCKS)),
            "wide_min_blocks": int(L.nig_tune_get(_lib.TUNE_WIDE_MIN_BLOCKS))}


__version__ = "0.5.0"        # generator "nig-philox-v3" since round 4 (v2 + PowerGrid's reset load factors from spare low bytes; libnig: nig_version())
GENERATOR = "nig-philox-v3"
__all__ = [
    "__version__", "DatasetQuality", "SafetyConstraint", "SafetyMetrics", "IndustrialEnv",
    "ChemicalReactorEnv", "PowerGridEnv", "RobotAssemblyEnv", "AdvancedChemicalReactorEnv", "AdvancedPowerGridEnv",
    "HVACControlEnv", "WaterTreatmentEnv", "SteelAnnealingEnv", "SupplyChainEnv", "BatchedIndustrialEnv", "MixedBatchedEnv", "StepInfo",
    "make", "make_batched", "evaluate_with_safety", "uniform_action_statistics", "tune", "DevicePolicy", "MLPPolicy", "behaviour_policy", "constant_agent",
    "mpc_agent", "pid_agent", "random_agent",
]
