"""Single-environment drop-in classes: the reference's IndustrialEnv surface
(environments/base.py:19-228) over ONE lane of the HIP batch kernel.

`ni.make('ChemicalReactor-v0')` returns one of these.  The arithmetic of reset/step runs
on the GPU through libnig.so; this file is host glue only: it draws the process noise
from NumPy's global RNG in exactly the reference's call order (so `np.random.seed(k)`
reproduces the reference's trajectories), moves ~100 bytes per step, and rebuilds the
reference's `info` dict / SafetyMetrics objects from the kernel's flag word.
For throughput use BatchedIndustrialEnv (ni.make_batched); this class exists so existing
single-env callers (agents, harnesses, evaluate_with_safety) run unchanged.
"""
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .batched import BatchedIndustrialEnv
from .core import SafetyConstraint, SafetyMetrics

f32 = np.float32


class IndustrialEnv:
    """Host mirror of environments/base.py:19-228 (one env instance, B = 1)."""

    ENV_ID: str = ""
    _no_clip = False          # the Advanced envs override step() and never clip

    def __init__(self, max_episode_steps: Optional[int] = None, dt: Optional[float] = None,
                 device="cuda:0", noise: str = "numpy", seed: int = 0x5EED):
        if noise not in ("numpy", "device"):
            raise ValueError("noise must be 'numpy' (reference-compatible global np.random) or 'device'")
        self._noise_mode = noise
        self._b = BatchedIndustrialEnv(self.ENV_ID, 1, device=device, seed=seed, max_episode_steps=max_episode_steps,
                                       dt=dt, autoreset=False, tally=False)
        sp = self._b.spec
        self.state_dim, self.action_dim = self._b.state_dim, self._b.action_dim       # base.py:41-42
        self.max_episode_steps, self.dt = self._b.max_episode_steps, self._b.dt       # base.py:43-44
        self._builtin = self._builtin_constraints()
        assert len(self._builtin) == int(sp.n_constraints)
        self.safety_constraints: List[SafetyConstraint] = list(self._builtin)          # base.py:47
        self.current_step = 0                                                          # base.py:50-53
        self.state = None
        self.done = False
        self.info: Dict[str, Any] = {}
        self.violation_count = 0                                                       # base.py:56-57
        self.total_violations = 0
        self.observation_space = self._b.observation_space                             # base.py:60-72
        self.action_space = self._b.action_space
        self._last_metrics: Optional[SafetyMetrics] = None
        self._needs_reset = True
        self._sig, self._mask, self._custom = None, 0, []
        # host-side I/O buffers of the host-buffer ABI (nig_reset_host / nig_step_host): no torch op per step
        import ctypes as C
        self._C = C
        self._st_buf = np.zeros((self.state_dim, 1), dtype=np.float32)
        self._act_buf = np.zeros((self.action_dim, 1), dtype=np.float32)
        self._act64_buf = np.zeros((self.action_dim, 1), dtype=np.float64)
        self._rew_buf = np.zeros(1, dtype=np.float64)
        self._fl_buf = np.zeros(1, dtype=np.uint32)
        self._p_st, self._p_act = self._st_buf.ctypes.data_as(C.c_void_p), self._act_buf.ctypes.data_as(C.c_void_p)
        self._p_act64 = self._act64_buf.ctypes.data_as(C.c_void_p)
        self._p_rew, self._p_fl = self._rew_buf.ctypes.data_as(C.c_void_p), self._fl_buf.ctypes.data_as(C.c_void_p)

    # -- per-env host pieces (overridden) --------------------------------------
    def _builtin_constraints(self) -> List[SafetyConstraint]:
        raise NotImplementedError

    def _draw_reset_noise(self) -> np.ndarray:
        raise NotImplementedError

    def _draw_step_noise(self) -> Optional[np.ndarray]:
        raise NotImplementedError

    def _get_safety_info(self, state) -> Dict[str, Any]:
        """base.py:126-131"""
        return {"safety_metrics": {}, "constraint_values": {}}

    # -- helpers ---------------------------------------------------------------
    def _pull_state(self) -> np.ndarray:
        return self._b.obs.cpu().numpy()[0].copy()   # out-of-band read (teacher forcing in tests)

    def _sync_constraint_mask(self):
        """Device mask of the built-ins still in the list + the user-added constraints (host-evaluated).
        Recomputed only when the list changed."""
        sig = tuple(map(id, self.safety_constraints))
        if sig != self._sig:
            mask = 0
            for k, c in enumerate(self._builtin):
                # a built-in stays on the device only while the identical object is still in the list
                if any(c is x for x in self.safety_constraints):
                    mask |= 1 << k
            self._b.set_constraint_mask(mask)
            self._sig, self._mask = sig, mask
            self._custom = [c for c in self.safety_constraints if not any(c is b for b in self._builtin)]
        return self._mask

    # -- API -------------------------------------------------------------------
    def reset(self, *, seed: Optional[int] = None, options: Optional[Dict] = None) -> Tuple[np.ndarray, Dict]:
        """base.py:133-155.  As upstream, `seed` does not touch the global np.random stream the
        envs draw from (gymnasium's Env.reset only seeds self.np_random)."""
        self.current_step = 0
        self.done = False
        self.violation_count = 0
        nz = None
        if self._noise_mode == "numpy":
            nz = np.ascontiguousarray(self._draw_reset_noise(), dtype=np.float64)
        _lib.check(self._b._L.nig_reset_host(self._b._h, None if nz is None or nz.size == 0 else nz.ctypes.data_as(self._C.c_void_p),
                                             self._p_st, None))
        self.state = self._st_buf[:, 0].copy()
        self._needs_reset = False
        obs = self.state.copy()
        info = self._get_safety_info(self.state)
        info.update({"step": self.current_step, "violations": self.violation_count,
                     "total_violations": self.total_violations})
        return obs, info

    def step(self, action) -> Tuple[np.ndarray, Any, bool, bool, Dict]:
        """base.py:157-213"""
        if self.done or self._needs_reset:
            raise RuntimeError("Environment is done. Call reset() first.")
        action = np.asarray(action)
        # base.py:163-167 clips without casting: a float64 action vector (what get_dataset and the baseline agents
        # build) makes NumPy evaluate the action-dependent arithmetic in float64 -> nig_step_host64 follows that
        act64 = action.dtype == np.float64
        a_in = action.reshape(self.action_dim) if act64 else action.astype(np.float32, copy=False).reshape(self.action_dim)
        a_clip = a_in if self._no_clip else np.clip(a_in, self.action_space.low, self.action_space.high)   # base.py:167
        state_pre = self.state
        mask = self._sync_constraint_mask()
        custom = self._custom

        sn = None
        if self._noise_mode == "numpy":
            sn = self._draw_step_noise()
            sn = None if sn is None else np.ascontiguousarray(sn, dtype=np.float64)
        # one call: upload action (+ noise), step kernel, download state / reward / flag word, one sync
        sn_p = None if sn is None else sn.ctypes.data_as(self._C.c_void_p)
        if act64:
            self._act64_buf[:, 0] = a_in
            _lib.check(self._b._L.nig_step_host64(self._b._h, self._p_act64, sn_p, self._p_st, self._p_rew, self._p_fl, None))
        else:
            self._act_buf[:, 0] = a_in
            _lib.check(self._b._L.nig_step_host(self._b._h, self._p_act, sn_p, self._p_st, self._p_rew, self._p_fl, None))
        new_state = self._st_buf[:, 0].copy()
        flags = int(self._fl_buf[0])
        reward64 = float(self._rew_buf[0])
        # reward type as upstream: np.float32 while everything was float32 (CR, build-specified plants), np.float64 once a
        # float64 action took part (CR: np.float32 - np.float64), a Python float where the env returns float(...)
        if int(self._b.spec.reward_is_f32):
            reward: Any = (np.float64(reward64) if act64 and self.ENV_ID == "ChemicalReactor-v0" else f32(reward64))
        else:
            reward = reward64

        nv = ((flags >> _lib.FLAG_NVIOL_SHIFT) & 3) + ((flags >> 13) & 1) * 4
        nc = (flags >> _lib.FLAG_NCRIT_SHIFT) & 3
        terminated = bool(flags & _lib.FLAG_TERMINATED)
        truncated = bool(flags & _lib.FLAG_TRUNCATED)
        n_total = len(self.safety_constraints)
        satisfied = bin(mask).count("1") - nv
        viol, crit = nv, nc

        # constraints added by the user are arbitrary Python callables (base.py:220-222): they are
        # evaluated here on the pre-state exactly as base.py:94-124 / 179-183 do, after the built-ins
        custom_crit = 0
        for c in custom:
            try:
                ok = bool(c.check_fn(state_pre, a_clip))
            except Exception:
                ok = False
            if ok:
                satisfied += 1
            else:
                viol += 1
                if c.critical:
                    crit += 1
                    custom_crit += 1
        if custom:
            extra_count = 0
            for c in custom:           # base.py:179-183 second evaluation; exceptions propagate upstream too
                if not c.check_fn(state_pre, a_clip):
                    reward = reward + c.penalty
                    extra_count += 1
            if custom_crit > 0 and nc == 0:
                terminated = True
                reward = reward - 1000.0
                self._b.set_state(current_step=self.current_step + 1,
                                  violation_count=self.violation_count + nv + extra_count, done=True)
            self.violation_count += extra_count
            self.total_violations += extra_count

        self.violation_count += nv
        self.total_violations += nv
        self.state = new_state
        self.current_step += 1
        self.done = terminated or truncated
        safety_metrics = SafetyMetrics(
            constraints_satisfied=satisfied, total_constraints=n_total, violation_count=viol,
            critical_violations=crit, safety_score=(satisfied / n_total) if n_total > 0 else 1.0)
        self._last_metrics = safety_metrics
        obs = self.state.copy()
        info = self._get_safety_info(self.state)
        info.update({"step": self.current_step, "violations": self.violation_count,
                     "total_violations": self.total_violations, "safety_metrics": safety_metrics,
                     "critical_shutdown": bool(flags & _lib.FLAG_SHUTDOWN) or crit > 0})
        return obs, reward, terminated, truncated, info

    def get_safety_metrics(self) -> SafetyMetrics:
        """Named by the reference README / callers (benchmarks/industrial_benchmarks.py:148) but only
        defined upstream on the non-instantiable Advanced envs; here: SafetyMetrics of the last step."""
        if self._last_metrics is None:
            n = len(self.safety_constraints)
            return SafetyMetrics(constraints_satisfied=n, total_constraints=n, violation_count=0,
                                 critical_violations=0, safety_score=1.0)
        return self._last_metrics

    def add_safety_constraint(self, constraint: SafetyConstraint) -> None:
        """base.py:220-222"""
        self.safety_constraints.append(constraint)

    def remove_safety_constraint(self, name: str) -> None:
        """base.py:224-228"""
        self.safety_constraints = [c for c in self.safety_constraints if c.name != name]

    def get_dataset(self, quality: str = "mixed"):
        """base.py:215-218 -- dataset generation is a 'next' row of the scope table."""
        from .datasets import get_dataset
        return get_dataset(self, quality)

    def close(self):
        self._b.close()


class ChemicalReactorEnv(IndustrialEnv):
    """environments/chemical_reactor.py:11-87"""
    ENV_ID = "ChemicalReactor-v0"

    def __init__(self, **kwargs):
        if "max_episode_steps" in kwargs or "dt" in kwargs:
            # upstream passes both positionally AND **kwargs to super().__init__ (chemical_reactor.py:62-69)
            raise TypeError("ChemicalReactorEnv() got multiple values for keyword argument "
                            "'max_episode_steps' or 'dt' (hard-coded to 500 / 0.1 upstream)")
        super().__init__(**kwargs)
        self.temp_min, self.temp_max, self.temp_target = 280.0, 350.0, 320.0            # :72-74
        self.pressure_min, self.pressure_max, self.pressure_target = 101325.0, 506625.0, 253312.5   # :76-78
        self.reaction_rate_constant, self.heat_capacity, self.reactor_volume = 0.1, 4.18e3, 1000.0  # :81-83
        self.temp_noise_std, self.pressure_noise_std = 1.0, 5000.0                      # :86-87

    def _builtin_constraints(self):
        return [   # chemical_reactor.py:38-60
            SafetyConstraint("temperature_limit", self._temperature_constraint, -100.0, True,
                             "Reactor temperature must stay below 350K"),
            SafetyConstraint("pressure_limit", self._pressure_constraint, -50.0, True,
                             "Reactor pressure must stay below 5 atm"),
            SafetyConstraint("level_safety", self._level_constraint, -25.0, False,
                             "Reactor level must stay between 20-90%"),
        ]

    # host restatements for introspection only (the device evaluates the built-ins), :292-305
    def _temperature_constraint(self, state, action) -> bool:
        temp = state[0] if len(state.shape) == 1 else state[0, 0]
        return bool(temp <= 350.0)

    def _pressure_constraint(self, state, action) -> bool:
        pressure = state[1] if len(state.shape) == 1 else state[1, 0]
        return bool(pressure <= 506625.0)

    def _level_constraint(self, state, action) -> bool:
        level = state[10] if len(state.shape) == 1 else state[10, 0]
        return bool(20 <= level <= 90)

    def _draw_reset_noise(self):   # call order of chemical_reactor.py:93-103
        n = np.random.normal
        return np.array([n(0, 2), n(0, 10000), n(0, 5), n(0, 3), n(0, 0.1), n(0, 2), n(0, 1), n(0, 5)],
                        dtype=np.float64)

    def _draw_step_noise(self):    # chemical_reactor.py:149,159
        return np.array([np.random.normal(0, 1.0 / 10), np.random.normal(0, 5000.0 / 10)], dtype=np.float64)

    def _get_safety_info(self, state):   # chemical_reactor.py:307-322
        return {
            "safety_metrics": {"temperature": state[0], "pressure": state[1], "level": state[10],
                               "emergency_stop": state[8], "alarm_status": state[9]},
            "constraint_values": {"temp_margin": self.temp_max - state[0],
                                  "pressure_margin": self.pressure_max - state[1],
                                  "level_in_bounds": 20 <= state[10] <= 90},
        }


def _pg_frequency_constraint(state, action) -> bool:      # power_grid.py:10-14
    return bool(abs(state[0]) < 0.5)


def _pg_voltage_constraint(state, action) -> bool:        # power_grid.py:17-21
    v = state[1:9]
    return bool(np.all((v >= 0.95) & (v <= 1.05)))


def _pg_generation_constraint(state, action) -> bool:     # power_grid.py:24-30
    new_gen = state[9:17] + action
    return bool(np.all((new_gen >= 0) & (new_gen <= np.ones(8) * 100)))


class PowerGridEnv(IndustrialEnv):
    """environments/power_grid.py:33-88"""
    ENV_ID = "PowerGrid-v0"

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.base_load = np.array([50, 60, 45, 55, 40, 65, 35, 50])                     # :82
        self.load_variation, self.inertia_constant, self.damping_factor = 0.2, 5.0, 1.0  # :83-85
        self.generation_cost = np.array([25, 30, 28, 35, 32, 27, 40, 33])               # :88

    def _builtin_constraints(self):
        return [   # power_grid.py:53-72
            SafetyConstraint("frequency_stability", _pg_frequency_constraint, -50.0, True),
            SafetyConstraint("voltage_limits", _pg_voltage_constraint, -30.0, True),
            SafetyConstraint("generation_limits", _pg_generation_constraint, -20.0, False),
        ]

    def _draw_reset_noise(self):   # power_grid.py:98-108
        return np.concatenate([np.random.normal(0, 0.01, 8), np.random.normal(0, 2, 8),
                               np.random.uniform(-0.2, 0.2, 8), np.random.normal(0, 10, 7)])

    def _draw_step_noise(self):    # power_grid.py:136,140,144
        return np.concatenate([np.random.normal(0, 0.005, 8), np.random.normal(0, 1, 8),
                               np.random.normal(0, 2, 7)])


def _ra_force_constraint(state, action) -> bool:          # robot_assembly.py:10-15
    return bool(np.all(np.abs(state[18:21]) < 50.0))


def _ra_collision_constraint(state, action) -> bool:      # robot_assembly.py:18-25
    p = state[0:3]
    return bool(np.all((p >= np.array([-0.5, -0.5, 0.0])) & (p <= np.array([0.5, 0.5, 0.8]))))


def _ra_velocity_constraint(state, action) -> bool:       # robot_assembly.py:28-32
    return bool(np.all(np.abs(state[7:14]) < 2.0))


class RobotAssemblyEnv(IndustrialEnv):
    """environments/robot_assembly.py:35-92"""
    ENV_ID = "RobotAssembly-v0"

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.link_lengths = np.array([0.3, 0.3, 0.25, 0.25, 0.15, 0.1, 0.05])           # :85
        self.joint_limits_low = np.array([-np.pi] * 7)                                  # :86-87
        self.joint_limits_high = np.array([np.pi] * 7)
        self.target_position = np.array([0.3, 0.0, 0.4])                                # :90
        self.insertion_depth, self.alignment_tolerance = 0.05, 0.005                    # :91-92

    def _builtin_constraints(self):
        return [   # robot_assembly.py:56-75
            SafetyConstraint("force_limits", _ra_force_constraint, -100.0, True),
            SafetyConstraint("collision_avoidance", _ra_collision_constraint, -200.0, True),
            SafetyConstraint("velocity_limits", _ra_velocity_constraint, -50.0, False),
        ]

    def _draw_reset_noise(self):   # robot_assembly.py:118-122
        return np.random.uniform(np.array([-np.pi] * 7) * 0.5, np.array([np.pi] * 7) * 0.5, 7)

    def _draw_step_noise(self):    # deterministic step
        return None


class _AdvancedEnv(IndustrialEnv):
    """Shared host glue of the two Advanced envs.  CANDIDATE ROWS: upstream
    neither class can be instantiated (abstract hooks missing, non-existent dataclass kwargs) and
    step() reads an attribute nothing sets, so these follow the source text's evident intent:
    step() overridden wholesale (no action clip, no base constraint loop, no -1000 shutdown),
    deterministic reset, episode_step = 0 at reset, SafetyMetrics built from the core fields."""
    _no_clip = True
    _VIOLATIONS = ()

    def __init__(self, **kwargs):
        kwargs.setdefault("noise", "device")            # nothing to draw: deterministic
        super().__init__(**kwargs)
        from .core import make_box
        lo, hi = self._ACTION_BOX
        self.action_space = make_box(np.array(lo, dtype=f32), np.array(hi, dtype=f32), (self.action_dim,), f32)

    def _builtin_constraints(self):
        return [SafetyConstraint(n, (lambda s, a, _n=n: True), 0.0, False,
                                 "evaluated on the device; see get_safety_metrics()") for n in self._VIOLATIONS]

    def _draw_reset_noise(self):
        return np.zeros(0)

    def _draw_step_noise(self):
        return None

    def step(self, action):
        obs, reward, terminated, truncated, info = super().step(action)
        fl = int(self._fl_buf[0])
        names = [n for k, n in enumerate(self._VIOLATIONS) if (fl >> (2 + k if k < 3 else 12)) & 1]
        info["violation_types"] = names
        info.update(self._extra_info(obs, action, fl))
        return obs, float(reward), terminated, truncated, info


class AdvancedChemicalReactorEnv(_AdvancedEnv):
    """environments/advanced_chemical_reactor.py (20-D CSTR, 6 actions)."""
    ENV_ID = "AdvancedChemicalReactor-v0"
    _ACTION_BOX = ([0.0, 0.0, 0.0, 273.15, 0.0, 0.0], [0.01, 0.01, 3000.0, 473.15, 100.0, 1.0])      # :148-155
    _VIOLATIONS = ("temperature_limit", "pressure_limit", "temperature_margin", "pressure_margin")    # :433-443

    def _extra_info(self, obs, action, fl):      # info dict of :354-361 (state-derived entries)
        return {"conversion": float(obs[17]), "residence_time": float(obs[16]),
                "emergency_shutdown": bool(np.asarray(action)[5] > 0.5),
                "pressure_relief_active": bool(np.asarray(action)[4] > 0)}


class AdvancedPowerGridEnv(_AdvancedEnv):
    """environments/advanced_power_grid.py (8-bus grid, 4 generators)."""
    ENV_ID = "AdvancedPowerGrid-v0"
    _ACTION_BOX = ([10.0, 8.0, 7.0, 9.0, 0.95, 0.95, 0.0, 0.0], [50.0, 40.0, 35.0, 45.0, 1.05, 1.05, 20.0, 1.0])   # :163-178
    _VIOLATIONS = ("frequency_deviation", "voltage_deviation", "generation_limits")                               # :511-523

    def _extra_info(self, obs, action, fl):      # :334-343
        H = np.array([5.0, 4.0, 3.5, 4.5], dtype=f32)
        return {"system_frequency": float(np.sum(obs[16:20] * H) / np.sum(H)),
                "total_generation": float(np.sum(obs[20:24])), "total_load": float(np.sum(obs[24:28])),
                "emergency_active": bool(np.asarray(action)[7] > 0.5), "load_shedding_amount": float(np.asarray(action)[6])}


class _SpecPlantEnv(IndustrialEnv):
    """The four README-only environments (README.md:28-32): BUILD-SPECIFIED plants, no reference
    implementation exists (spec_plants.py holds the model and its numbers).  Same base-class surface
    as the three real envs: clip, constraint check on the pre-state, penalties, critical shutdown."""
    _PLANT = None

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        P = self._PLANT
        self.variable_names = [y["name"] for y in P["y"]] + [a["name"] for a in P["act"]] + \
            ["effort", "effort_integral", "elapsed_time"]

    def _builtin_constraints(self):
        def box(first, count, lo, hi):
            return lambda state, action: bool(np.all((np.asarray(state)[first:first + count] >= f32(lo)) &
                                                     (np.asarray(state)[first:first + count] <= f32(hi))))
        return [SafetyConstraint(name, box(first, count, lo, hi), float(pen), bool(crit),
                                 f"rows {first}..{first + count - 1} within [{lo}, {hi}] (build-specified)")
                for (name, first, count, lo, hi, pen, crit) in self._PLANT["constraints"]]

    def _draw_reset_noise(self):
        return np.array([np.random.normal(0, y["sd0"]) for y in self._PLANT["y"]], dtype=np.float64)

    def _draw_step_noise(self):
        sd = self._PLANT["noise_sd"]
        return np.array([np.random.normal(0, sd[0]), np.random.normal(0, sd[1])], dtype=np.float64)

    def _get_safety_info(self, state):
        return {"safety_metrics": {c[0]: bool(chk.check_fn(state, None)) for c, chk in zip(self._PLANT["constraints"], self._builtin)},
                "constraint_values": {n: float(state[i]) for i, n in enumerate(self.variable_names)}}


def _spec_env(name, plant):
    return type(name, (_SpecPlantEnv,), {"ENV_ID": plant["name"], "_PLANT": plant,
                                         "__doc__": f"{plant['name']}: build-specified plant (README-only upstream)."})


from .spec_plants import HVAC as _HVAC, STEEL as _STEEL, SUPPLY as _SUPPLY, WATER as _WATER   # noqa: E402

HVACControlEnv = _spec_env("HVACControlEnv", _HVAC)
WaterTreatmentEnv = _spec_env("WaterTreatmentEnv", _WATER)
SteelAnnealingEnv = _spec_env("SteelAnnealingEnv", _STEEL)
SupplyChainEnv = _spec_env("SupplyChainEnv", _SUPPLY)
