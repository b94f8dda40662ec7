#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- generate golden vectors by RUNNING the reference.

Runs in the build container only (needs /root/reference).  Output: small .npz
fixtures under tests/golden/ holding *data* (inputs, recorded RNG draws, outputs of
the reference's IndustrialEnv.step / reset / evaluate_with_safety).  No reference
source text is stored.  Re-run with:  python oracle/gen_golden.py

Fixture families:
  <env>_g1.npz  single-step tuples harvested from real rollouts under several policies
  <env>_g2.npz  teacher-forced threshold / edge cases (one ulp either side of constants)
  <env>_g3.npz  full rollouts (ragged, concatenated) with per-step outputs + episode sums
  <env>_g4.npz  evaluate_with_safety() result dict for a fixed stub agent + its noise
  <env>_g5.npz  the G1 inputs with the action handed over as float64 (float32-valued)     [arg: g5]
  <env>_g6.npz  the G1 states stepped with genuine float64 action values                  [arg: g6]
  datasets.npz  env.get_dataset(quality) under np.random.seed(123): heads + summaries      [arg: datasets]
  baseline_agents.npz  act() sequences of benchmarks/baseline_agents.py on recorded obs   [arg: agents]
  behaviour_laws.npz   get_dataset(quality)'s action law with its own noise draws patched to zero: (obs, action)  [arg: laws]
  cr_info.npz   ChemicalReactor's reset / step info dicts (_get_safety_info margins) along a seeded run          [arg: info]
  reference_stats.npz  per-episode outcomes of the reference's own measurement loop (performance_benchmark.py:106-133)
                under uniform float32 actions and the reference's own np.random draws, thousands of episodes per env  [arg: stats]
Reference semantics pinned: NumPy 2.2.6; float32 actions (G1-G4) and float64 actions (G5, G6, datasets).
"""
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_import import NoiseTap, load_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
ENVS = {"cr": "ChemicalReactor-v0", "pg": "PowerGrid-v0", "ra": "RobotAssembly-v0"}
f32 = np.float32


def ulp_step(x, k):
    """k-th float32 neighbour of x (k may be negative)."""
    x = f32(x)
    for _ in range(abs(k)):
        x = np.nextafter(x, f32(np.inf) if k > 0 else f32(-np.inf), dtype=f32)
    return x


class StepLog:
    """Accumulates per-step tuples in the common fixture schema."""

    def __init__(self, S, A, K):
        self.S, self.A, self.K = S, A, K
        self.rows = {k: [] for k in (
            "state_pre", "action", "noise", "step_pre", "viol_pre", "state_next", "reward",
            "terminated", "truncated", "viol", "crit", "bits", "violations_after")}

    def add(self, **kw):
        for k, v in kw.items():
            self.rows[k].append(v)

    def arrays(self):
        r = self.rows
        n = len(r["reward"])
        return {
            "state_pre": np.asarray(r["state_pre"], dtype=f32).reshape(n, self.S),
            "action": np.asarray(r["action"], dtype=f32).reshape(n, self.A),
            "noise": np.asarray(r["noise"], dtype=np.float64).reshape(n, self.K),
            "step_pre": np.asarray(r["step_pre"], dtype=np.int32),
            "viol_pre": np.asarray(r["viol_pre"], dtype=np.int32),
            "state_next": np.asarray(r["state_next"], dtype=f32).reshape(n, self.S),
            "reward": np.asarray(r["reward"], dtype=np.float64),
            "terminated": np.asarray(r["terminated"], dtype=np.uint8),
            "truncated": np.asarray(r["truncated"], dtype=np.uint8),
            "viol": np.asarray(r["viol"], dtype=np.int32),
            "crit": np.asarray(r["crit"], dtype=np.int32),
            "bits": np.asarray(r["bits"], dtype=np.uint8).reshape(n, 3),
            "violations_after": np.asarray(r["violations_after"], dtype=np.int32),
        }


def constraint_bits(env, state_pre, action_raw):
    a = np.clip(action_raw, env.action_space.low, env.action_space.high)
    return [1 if bool(c.check_fn(state_pre, a)) else 0 for c in env.safety_constraints]


def do_step(env, tap, action, log, K):
    """One reference env.step() with recording.  `action` must be float32."""
    assert action.dtype == np.float32
    state_pre = env.state.copy()
    step_pre, viol_pre = env.current_step, env.violation_count
    bits = constraint_bits(env, state_pre, action)
    tap.take()
    obs, reward, term, trunc, info = env.step(action)
    noise = tap.take()
    assert noise.size == K, (noise.size, K)
    sm = info["safety_metrics"]
    assert sm.violation_count == 3 - sum(bits)
    log.add(state_pre=state_pre, action=action.copy(), noise=noise, step_pre=step_pre,
            viol_pre=viol_pre, state_next=obs.copy(), reward=float(reward),
            terminated=int(bool(term)), truncated=int(bool(trunc)),
            viol=sm.violation_count, crit=sm.critical_violations, bits=bits,
            violations_after=info["violations"])
    return obs, reward, term, trunc, info


def force(env, state, step=0, viol=0):
    env.state = np.asarray(state, dtype=f32).copy()
    env.current_step = int(step)
    env.violation_count = int(viol)
    env.done = False


def forced_step(env, tap, state, action, noise, log, K, step=0, viol=0):
    force(env, state, step, viol)
    tap.forced = list(np.asarray(noise, dtype=np.float64).ravel())
    try:
        return do_step(env, tap, np.asarray(action, dtype=f32), log, K)
    finally:
        tap.forced = None


# ----------------------------------------------------------------------------------
# policies used to harvest G1 / G3 (all produce float32 actions)
# ----------------------------------------------------------------------------------
def make_policies(key, A, rng):
    def uniform(obs):
        return rng.uniform(-1, 1, A).astype(f32)

    def zeros(obs):
        return np.zeros(A, dtype=f32)

    def wide(obs):  # exercises the clip at the Box bounds
        return rng.uniform(-2.5, 2.5, A).astype(f32)

    def bang(obs):
        return np.where(rng.random(A) < 0.5, -1.0, 1.0).astype(f32)

    pols = [("uniform", uniform), ("zeros", zeros), ("wide", wide), ("bang", bang)]
    if key == "cr":
        def hot(obs):
            return np.array([1.0, -1.0, 1.0], dtype=f32)

        def drain(obs):
            return np.array([-0.2, 0.3, -1.0], dtype=f32)

        def pid(obs):  # behaviour-policy shape of chemical_reactor.py get_dataset('expert'), cast f32
            te = (obs[0] - f32(320.0)) / f32(50)
            le = (obs[10] - f32(55)) / f32(50)
            return np.array([-te * f32(0.5), te * f32(0.3), -le * f32(0.2)], dtype=f32) + \
                rng.normal(0, 0.01, 3).astype(f32)
        pols += [("hot", hot), ("drain", drain), ("pid", pid)]
    if key == "pg":
        def balance(obs):
            imb = f32(np.sum(obs[17:25]) - np.sum(obs[9:17]))
            return (f32(-0.5) * obs[0] * np.ones(A, dtype=f32) + f32(0.1) * imb / f32(A)).astype(f32)
        pols += [("balance", balance)]
    if key == "ra":
        tgt = np.array([0.3, 0.0, 0.4], dtype=f32)

        def reach(obs):
            e = (tgt - obs[0:3]).astype(f32)
            return np.concatenate([f32(2.0) * e, f32(-0.1) * obs[10:14]]).astype(f32)
        pols += [("reach", reach)]
    return pols


def harvest_g1(utils, key, n_target, seed):
    env = utils.make(ENVS[key])
    S, A = env.state_dim, env.action_dim
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    pols = make_policies(key, A, rng)
    with NoiseTap(seed) as tap:
        env.reset()
        K = None
        # discover K (draws per step)
        tap.take()
        env.step(np.zeros(A, dtype=f32))
        K = tap.take().size
        log = StepLog(S, A, K)
        per = n_target // len(pols)
        for name, pol in pols:
            got = 0
            while got < per:
                obs, _ = env.reset()
                tap.take()
                done = False
                while not done and got < per:
                    obs, r, te, tr, info = do_step(env, tap, pol(obs), log, K)
                    done = te or tr
                    got += 1
    return log.arrays(), K


# ----------------------------------------------------------------------------------
# G2: teacher-forced edge cases
# ----------------------------------------------------------------------------------
def g2_cr(utils, seed):
    env = utils.make(ENVS["cr"])
    rng = np.random.Generator(np.random.PCG64(seed))
    K = 2
    log = StepLog(12, 3, K)
    nominal = np.array([320, 253312.5, 50, 30, 0.5, 95, 295, 0, 0, 0, 60, 0], dtype=f32)

    def base():
        s = nominal.copy()
        s[[0, 2, 3, 4, 5, 6, 10]] += rng.normal(0, [1, 2, 1, 0.05, 1, 0.5, 2]).astype(f32)
        return s

    def act():
        return rng.uniform(-1, 1, 3).astype(f32)

    with NoiseTap(seed) as tap:
        env.reset()

        def go(s, a=None, noise=(0.0, 0.0), step=0, viol=0):
            return forced_step(env, tap, s, act() if a is None else a, noise, log, K, step, viol)

        # (a) constraint thresholds on the pre-state
        for idx, thr in ((0, 350.0), (1, 506625.0), (10, 20.0), (10, 90.0)):
            for k in (-2, -1, 0, 1, 2):
                s = base(); s[idx] = ulp_step(thr, k); go(s)
        # level thresholds seen by reward / is_done on the NEXT state: feed'=20 -> level'=level
        for thr in (5.0, 95.0, 30.0, 80.0, 0.0, 100.0):
            for k in (-2, -1, 0, 1, 2):
                s = base(); s[3] = 20.0; s[10] = ulp_step(thr, k)
                go(s, np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), 0.0], dtype=f32))
        # (b) e-stop / alarm pass-through thresholds, e-stop mode dynamics
        for v in (ulp_step(0.5, -1), f32(0.5), ulp_step(0.5, 1), f32(1.0), f32(0.25)):
            for _ in range(4):
                s = base(); s[8] = v; go(s, noise=rng.normal(0, [0.1, 500]))
                s = base(); s[9] = v; go(s, noise=rng.normal(0, [0.1, 500]))
                s = base(); s[8] = v; s[9] = 1.0; go(s)
        # (c) relief valve + pressure floor (Python min/max int leakage paths)
        for rel in (0.0, 1e-3, 50.0, 99.9, 100.0, 150.0, -5.0):
            for P in (101325.0, 150000.0, 253312.5, 480000.0, 506000.0, 506625.0, 507000.0,
                      520000.0, 606625.0, 706625.0, 110000.0):
                s = base(); s[7] = rel; s[1] = P; go(s, noise=(0.0, float(rng.normal(0, 500))))
        # (d) flow clamps
        for cool in (10.0, 10.05, 10.1, 99.9, 99.95, 100.0, 5.0, 120.0):
            for a1 in (-1.0, 1.0, 0.0):
                s = base(); s[2] = cool; go(s, np.array([0.1, a1, 0.0], dtype=f32))
        for feed in (5.0, 5.05, 5.1, 49.9, 49.95, 50.0, 2.0, 60.0, 20.0):
            for a2 in (-1.0, 1.0, 0.0):
                s = base(); s[3] = feed; go(s, np.array([0.0, 0.2, a2], dtype=f32))
        # (e) concentration floor, catalyst floor
        for conc in (0.0, 1e-4, 2e-3, -0.01, 3e-3, 5.0):
            for feed in (5.0, 30.0, 50.0):
                s = base(); s[4] = conc; s[3] = feed; go(s)
        for cat in (50.0, 50.00005, 50.0005, 50.002, 49.0, 100.0):
            for T in (330.0, 341.0):
                s = base(); s[5] = cat; s[0] = T; go(s)
        # (f) next-state temperature thresholds 340 / 345 / 350, dense in ulps
        for thr in (340.0, 345.0, 350.0):
            for k in range(-12, 13):
                s = base(); s[0] = f32(thr - 0.05); a = act()
                force(env, s); tap.forced = [0.0, 0.0]
                o0 = env.step(a)[0]; tap.forced = None; tap.take()
                target = float(ulp_step(thr, k))
                n0 = (target - float(o0[0])) / 0.1
                go(s, a, noise=(n0, float(rng.normal(0, 500))))
        # (g) next-state pressure thresholds 480000 / 506625
        for thr in (480000.0, 506625.0):
            for k in range(-12, 13):
                s = base(); s[1] = f32(thr - 2000.0); a = act()
                n0 = float(rng.normal(0, 0.1))
                force(env, s); tap.forced = [n0, 0.0]
                o0 = env.step(a)[0]; tap.forced = None; tap.take()
                # relief not active below 506625 so P' is linear in n1 there
                target = float(ulp_step(thr, k))
                n1 = target - float(o0[1])
                go(s, a, noise=(n0, n1))
        # (h) batch time limit
        for k in range(-6, 7):
            s = base(); s[11] = ulp_step(49.9, k); go(s)
        for bt in (49.7, 49.8, 50.0, 50.1, 0.0):
            s = base(); s[11] = bt; go(s)
        # (i) truncation boundary + running violation counter
        for step in (0, 1, 497, 498, 499):
            s = base(); go(s, step=step, viol=step // 7)
            s = base(); s[10] = 15.0; go(s, step=step, viol=3)
            s = base(); s[0] = 351.0; s[1] = 510000.0; s[10] = 95.5; go(s, step=step, viol=11)
        # (j) actions outside the Box (clip) incl. exact bounds
        for a in ([1.0, -1.0, 1.0], [1.5, -3.0, 7.0], [-1.0000001, 1.0000001, 0.0], [0, 0, 0]):
            s = base(); go(s, np.array(a, dtype=f32), noise=rng.normal(0, [0.1, 500]))
        # (k) random broad states (both modes) for general coverage
        for _ in range(600):
            s = base()
            s[0] = rng.uniform(300, 356); s[1] = rng.uniform(9e4, 5.3e5)
            s[2] = rng.uniform(8, 102); s[3] = rng.uniform(4, 52); s[4] = rng.uniform(-0.05, 1.5)
            s[5] = rng.uniform(49, 100); s[6] = rng.uniform(285, 305)
            s[7] = rng.choice([0.0, 0.0, rng.uniform(0, 100)])
            s[8] = rng.choice([0.0, 0.0, 0.0, 1.0]); s[9] = rng.choice([0.0, 1.0])
            s[10] = rng.uniform(0, 100); s[11] = rng.uniform(0, 50.2)
            go(s, rng.uniform(-1.3, 1.3, 3).astype(f32), noise=rng.normal(0, [0.1, 500]),
               step=int(rng.integers(0, 500)), viol=int(rng.integers(0, 50)))
    return log.arrays(), K


def g2_pg(utils, seed):
    env = utils.make(ENVS["pg"])
    rng = np.random.Generator(np.random.PCG64(seed))
    K = 23
    log = StepLog(32, 8, K)
    base_load = np.array([50, 60, 45, 55, 40, 65, 35, 50], dtype=f32)

    def base():
        s = np.zeros(32, dtype=f32)
        s[1:9] = 1.0 + rng.normal(0, 0.01, 8)
        s[9:17] = base_load + rng.normal(0, 2, 8)
        s[17:25] = base_load * (1.0 + rng.uniform(-0.2, 0.2, 8))
        s[25:32] = rng.normal(0, 10, 7)
        return s

    def act():
        return rng.uniform(-1, 1, 8).astype(f32)

    def noise(scale=1.0):
        return np.concatenate([rng.normal(0, 0.005, 8), rng.normal(0, 1, 8), rng.normal(0, 2, 7)]) * scale

    with NoiseTap(seed) as tap:
        env.reset()

        def go(s, a=None, nz=None, step=0, viol=0):
            return forced_step(env, tap, s, act() if a is None else a,
                               np.zeros(K) if nz is None else nz, log, K, step, viol)

        # (a) frequency thresholds 0.5 (constraint, pre) and 1.0 (is_done, post; imbalance == f keeps f'=f)
        for sign in (1.0, -1.0):
            for thr in (0.5, 1.0, 0.25):
                for k in (-2, -1, 0, 1, 2):
                    s = base(); f = ulp_step(sign * thr, k)
                    s[0] = f; s[9:17] = base_load; s[17:25] = base_load; s[17] = f32(50.0) - f
                    go(s, np.zeros(8, dtype=f32))
                    s = base(); s[0] = f; go(s, nz=noise())
        # (b) voltage thresholds (pre-state constraint 0.95/1.05, next-state is_done 0.9/1.1)
        for thr in (0.95, 1.05, 0.9, 1.1):
            for k in (-2, -1, 0, 1, 2):
                for bus in (0, 3, 7):
                    s = base(); s[1 + bus] = ulp_step(thr, k); go(s)
                s = base(); s[1:9] = ulp_step(thr, k); go(s)
        for thr in (0.9, 1.1):  # crossing via the fp64 noise add
            for d in (-3e-8, -1e-8, 0.0, 1e-8, 3e-8, 6e-8, -6e-8):
                s = base(); s[4] = f32(1.0); nz = np.zeros(K); nz[3] = (thr - 1.0) + d; go(s, nz=nz)
        # (c) generation limits 0 / 100 (constraint uses gen + clipped action; dynamics clip)
        for g, a in ((99.5, 0.5), (99.75, 0.5), (100.0, 0.0), (100.0, 1.0), (0.25, -0.5), (0.5, -0.5),
                     (0.0, 0.0), (0.0, -1.0), (99.0, 3.0), (1.0, -3.0), (100.5, -0.25), (-0.5, 0.25)):
            for unit in (0, 5):
                s = base(); s[9 + unit] = g; av = act(); av[unit] = a; go(s, av, noise())
        # (d) load floor at zero through the fp64 add
        for ld, nl in ((0.5, -0.5), (0.5, -0.75), (0.25, -0.2499999), (0.0, 0.0), (1e-3, -2e-3), (3.0, -2.0)):
            s = base(); s[17 + 2] = ld; nz = noise(); nz[8 + 2] = nl; go(s, nz=nz)
        # (e) truncation boundary (default max_episode_steps = 1000)
        for step in (0, 997, 998, 999):
            s = base(); go(s, nz=noise(), step=step, viol=step // 100)
            s = base(); s[0] = 0.7; go(s, nz=noise(), step=step, viol=2)
        # (f) clip of the action
        for a in (np.full(8, 1.0), np.full(8, -1.0), rng.uniform(-4, 4, 8), np.full(8, 1.0000001)):
            s = base(); go(s, a.astype(f32), noise())
        # (g) broad random states
        for _ in range(500):
            s = base()
            s[0] = rng.uniform(-1.2, 1.2)
            s[1:9] = rng.uniform(0.88, 1.12, 8) if rng.random() < 0.5 else s[1:9]
            s[9:17] = rng.uniform(-2, 103, 8) if rng.random() < 0.5 else s[9:17]
            s[17:25] = rng.uniform(0, 90, 8)
            go(s, rng.uniform(-1.5, 1.5, 8).astype(f32), noise(rng.choice([1.0, 3.0])),
               step=int(rng.integers(0, 1000)), viol=int(rng.integers(0, 9)))
    return log.arrays(), K


def _ra_fk(q):
    L = [0.3, 0.3, 0.25, 0.25, 0.15, 0.1, 0.05]
    x = sum(L[i] * math.cos(q[i]) for i in (0, 2, 4, 6))
    z = sum(L[i] * math.sin(q[i]) for i in (0, 2, 4, 6))
    y = sum(L[i] * math.sin(q[i]) for i in (1, 3, 5))
    return x, y, z


def g2_ra(utils, seed):
    env = utils.make(ENVS["ra"])
    rng = np.random.Generator(np.random.PCG64(seed))
    K = 0
    log = StepLog(24, 7, K)

    def state_from_q(q):
        s = np.zeros(24, dtype=f32)
        s[0:3] = _ra_fk(q); s[3:7] = [0, 0, 0, 1]; s[7:14] = q
        return s

    def base():
        return state_from_q(rng.uniform(-math.pi / 2, math.pi / 2, 7))

    def act():
        return rng.uniform(-1, 1, 7).astype(f32)

    with NoiseTap(seed) as tap:
        env.reset(); tap.take()

        def go(s, a=None, step=0, viol=0):
            return forced_step(env, tap, s, act() if a is None else a, [], log, K, step, viol)

        # (a) pre-state constraint thresholds
        for comp in (18, 19, 20):
            for thr in (50.0, -50.0, 30.0, 80.0, -80.0):
                for k in (-1, 0, 1):
                    s = base(); s[comp] = ulp_step(thr, k); go(s)
        for comp, thrs in ((0, (-0.5, 0.5, -0.6, 0.6)), (1, (-0.5, 0.5)), (2, (0.0, 0.8, -0.1, 0.9))):
            for thr in thrs:
                for k in (-2, -1, 0, 1, 2):
                    s = base(); s[comp] = ulp_step(thr, k); go(s)
        for j in (0, 3, 6):
            for thr in (2.0, -2.0):
                for k in (-1, 0, 1):
                    s = base(); s[7 + j] = ulp_step(thr, k); go(s)
        # (b) joint limit clip at +-pi (f32(pi) > pi: stored value exceeds the f64 limit)
        for qv in (3.1, 3.14, 3.1415927, 3.2, -3.1, -3.1415927, -3.3, 3.0415927):
            for a0 in (1.0, -1.0, 0.0):
                s = base(); s[7 + 2] = qv; a = act(); a[2] = a0; go(s, a)
        # (c) next-state workspace exit (is_done) and in-bounds cases
        for q in ([0, 0, 0, 0, 0, 0, 0], [0.3, 0, -0.3, 0, 0.2, 0, 0.1], [1.2, 0.2, 1.0, -0.3, 0.8, 0.1, 0.5],
                  [-1.2, 1.5, -1.0, 1.5, -0.8, 1.5, -0.5], [1.5, -1.5, 1.5, -1.5, 1.5, -1.5, 1.5]):
            go(state_from_q(np.array(q, dtype=np.float64)), np.zeros(7, dtype=f32))
            go(state_from_q(np.array(q, dtype=np.float64)))
        # (d) completion > 0.95:  q0=pi/2, q2=0, q4=-pi/2, q6=0 -> p=(0.30, 0, 0.15)
        qc = np.array([math.pi / 2, 0, 0, 0, -math.pi / 2, 0, 0.0])
        for d6 in (0.0, 1e-3, 5e-3, 2e-2, 5e-2, 9e-2, 0.1, 0.12):
            q = qc.copy(); q[6] = d6
            go(state_from_q(q), np.zeros(7, dtype=f32))
        for dz in (0.0, 0.2, 0.3, 0.33, 0.34):  # depth score between 0 and 1
            q = qc.copy(); q[4] = -math.pi / 2 + dz * 4
            go(state_from_q(q), np.zeros(7, dtype=f32))
        # (e) contact spring: reach p=(0.3,0,0.4) with the last two even links (2-link IK)
        x, z, l1, l2 = 0.05, 0.10, 0.15, 0.05
        c2 = (x * x + z * z - l1 * l1 - l2 * l2) / (2 * l1 * l2)
        th2 = math.acos(c2)
        th1 = math.atan2(z, x) - math.atan2(l2 * math.sin(th2), l1 + l2 * math.cos(th2))
        qk = np.array([math.pi / 2, 0, 0, 0, th1, 0, th1 + th2])
        for dq in (0.0, 1e-3, 1e-2, 3e-2, 6e-2, 0.1, 0.15, 0.19, 0.2, 0.21, 0.25, -0.05, -0.2):
            q = qk.copy(); q[6] += dq
            go(state_from_q(q), np.zeros(7, dtype=f32))
            q = qk.copy(); q[1] = dq * 0.1
            go(state_from_q(q), np.zeros(7, dtype=f32))
        # (f) truncation boundary
        for step in (0, 997, 998, 999):
            q = np.array([1.4, 0.1, 1.3, -0.1, 1.2, 0.05, 1.0])  # inside workspace
            go(state_from_q(q), step=step, viol=step // 50)
        # (g) clip of the action
        for a in (np.full(7, 1.0), np.full(7, -1.0), rng.uniform(-4, 4, 7), np.full(7, -1.0000001)):
            go(base(), a.astype(f32))
        # (h) broad random (pre-state position decoupled from q on purpose: velocity term)
        for _ in range(500):
            s = base()
            if rng.random() < 0.5:
                s[0:3] += rng.normal(0, 0.05, 3).astype(f32)
            if rng.random() < 0.3:
                s[7:14] = rng.uniform(-3.3, 3.3, 7)
            if rng.random() < 0.2:
                s[18:21] = rng.uniform(-90, 90, 3)
            s[14:18] = rng.normal(0, 1, 4); s[21:24] = rng.uniform(0, 1, 3)
            go(s, rng.uniform(-1.5, 1.5, 7).astype(f32), step=int(rng.integers(0, 1000)),
               viol=int(rng.integers(0, 9)))
    return log.arrays(), K


# ----------------------------------------------------------------------------------
# G3 rollouts / G4 evaluate_with_safety
# ----------------------------------------------------------------------------------
class StubAgent:
    """Fixed elementwise policy: a_j = clip(k_j * (obs[i_j] - c_j), -1, 1) in float32.

    One subtract + one multiply per action => IEEE-exact, reproducible on any host.
    The same parameters are re-created in tests/ from the values stored in the fixture.
    """
    is_trained = True

    def __init__(self, idx, ref, gain):
        self.idx = np.asarray(idx, dtype=np.int64)
        self.ref = np.asarray(ref, dtype=f32)
        self.gain = np.asarray(gain, dtype=f32)
        self.trace = []

    def predict(self, obs, deterministic=True):
        obs = np.asarray(obs, dtype=f32)
        a = np.clip((obs[:, self.idx] - self.ref) * self.gain, f32(-1), f32(1)).astype(f32)
        self.trace.append(a[0].copy())
        return a


STUB = {
    "cr": dict(idx=[0, 0, 10], ref=[320.0, 320.0, 55.0], gain=[-0.05, 0.03, -0.02]),
    "pg": dict(idx=[0] * 8, ref=[0.0] * 8, gain=[-0.5, -0.4, -0.3, -0.2, -0.5, -0.4, -0.3, -0.2]),
    "ra": dict(idx=[0, 1, 2, 7, 8, 9, 10], ref=[0.3, 0.0, 0.4, 0, 0, 0, 0],
               gain=[-2.0, -2.0, -2.0, -0.1, -0.1, -0.1, -0.1]),
}


def record_rollouts(utils, key, n_episodes, seed, policy_name, KR, K, use_eval=False):
    env = utils.make(ENVS[key])
    S, A = env.state_dim, env.action_dim
    rng = np.random.Generator(np.random.PCG64(seed + 7))
    ep = dict(init_noise=[], init_state=[], length=[], ret=[], viol=[], crit=[], shut=[], term_last=[],
              trunc_last=[])
    st = dict(action=[], noise=[], obs=[], reward=[], term=[], trunc=[], viol=[], crit=[])
    agent = StubAgent(**STUB[key])
    result = None
    with NoiseTap(seed) as tap:
        orig_reset, orig_step = env.reset, env.step

        def reset(**kw):
            tap.take()
            obs, info = orig_reset(**kw)
            n = tap.take(); assert n.size == KR
            ep["init_noise"].append(n); ep["init_state"].append(obs.copy())
            for k in ("length", "viol", "crit", "shut"):
                ep[k].append(0)
            ep["ret"].append(0.0)
            ep["term_last"].append(0); ep["trunc_last"].append(0)
            return obs, info

        def step(action):
            action = np.asarray(action, dtype=f32)
            tap.take()
            obs, r, te, tr, info = orig_step(action)
            n = tap.take(); assert n.size == K
            sm = info["safety_metrics"]
            st["action"].append(action.copy()); st["noise"].append(n); st["obs"].append(obs.copy())
            st["reward"].append(float(r)); st["term"].append(int(bool(te))); st["trunc"].append(int(bool(tr)))
            st["viol"].append(sm.violation_count); st["crit"].append(sm.critical_violations)
            ep["length"][-1] += 1
            ep["ret"][-1] = ep["ret"][-1] + r      # same accumulation expression as utils.py:99
            ep["viol"][-1] += sm.violation_count; ep["crit"][-1] += sm.critical_violations
            ep["shut"][-1] += int(bool(info["critical_shutdown"]))
            ep["term_last"][-1] = int(bool(te)); ep["trunc_last"][-1] = int(bool(tr))
            return obs, r, te, tr, info

        env.reset, env.step = reset, step
        if use_eval:
            result = utils.evaluate_with_safety(agent, env, n_episodes=n_episodes)
        else:
            pols = dict(make_policies(key, A, rng))
            pol = pols[policy_name]
            for _ in range(n_episodes):
                obs, _ = env.reset()
                done = False
                while not done:
                    obs, r, te, tr, info = env.step(pol(obs))
                    done = te or tr
    n = len(st["reward"])
    E = len(ep["length"])
    out = {
        "ep_init_noise": np.asarray(ep["init_noise"], dtype=np.float64).reshape(E, KR),
        "ep_init_state": np.asarray(ep["init_state"], dtype=f32).reshape(E, S),
        "ep_length": np.asarray(ep["length"], dtype=np.int32),
        "ep_return": np.asarray([float(x) for x in ep["ret"]], dtype=np.float64),
        "ep_viol": np.asarray(ep["viol"], dtype=np.int64),
        "ep_crit": np.asarray(ep["crit"], dtype=np.int64),
        "ep_shutdown": np.asarray(ep["shut"], dtype=np.int64),
        "ep_offsets": np.concatenate([[0], np.cumsum(ep["length"])]).astype(np.int64),
        "action": np.asarray(st["action"], dtype=f32).reshape(n, A),
        "noise": np.asarray(st["noise"], dtype=np.float64).reshape(n, K),
        "obs": np.asarray(st["obs"], dtype=f32).reshape(n, S),
        "reward": np.asarray(st["reward"], dtype=np.float64),
        "terminated": np.asarray(st["term"], dtype=np.uint8),
        "truncated": np.asarray(st["trunc"], dtype=np.uint8),
        "viol": np.asarray(st["viol"], dtype=np.int32),
        "crit": np.asarray(st["crit"], dtype=np.int32),
    }
    if result is not None:
        out["result_json"] = np.frombuffer(json.dumps(
            {k: float(v) for k, v in result.items()}, sort_keys=True).encode(), dtype=np.uint8)
        for k in ("idx", "ref", "gain"):
            out["agent_" + k] = np.asarray(STUB[key][k], dtype=np.float64)
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    utils = load_reference()
    meta = {"numpy": np.__version__, "envs": ENVS}
    KR = {"cr": 8, "pg": 31, "ra": 7}
    g2 = {"cr": g2_cr, "pg": g2_pg, "ra": g2_ra}
    g1_target = {"cr": 4200, "pg": 4200, "ra": 4200}       # >= 4096 tuples per env
    g3_eps = {"cr": 64, "pg": 64, "ra": 64}                 # 64 episodes per env
    for key in ENVS:
        arr, K = harvest_g1(utils, key, g1_target[key], seed=1000 + len(key) + ord(key[0]))
        np.savez_compressed(os.path.join(OUT, f"{key}_g1.npz"), **arr)
        print(key, "g1", arr["reward"].shape[0], "K", K,
              "crit>0:", int((arr["crit"] > 0).sum()), "term:", int(arr["terminated"].sum()))
        arr2, K2 = g2[key](utils, seed=2000 + ord(key[0]))
        assert K2 == K
        np.savez_compressed(os.path.join(OUT, f"{key}_g2.npz"), **arr2)
        print(key, "g2", arr2["reward"].shape[0], "crit>0:", int((arr2["crit"] > 0).sum()),
              "term:", int(arr2["terminated"].sum()), "trunc:", int(arr2["truncated"].sum()))
        g3 = record_rollouts(utils, key, g3_eps[key], 3000 + ord(key[0]), "uniform", KR[key], K)
        np.savez_compressed(os.path.join(OUT, f"{key}_g3.npz"), **g3)
        print(key, "g3 episodes", g3["ep_length"].size, "steps", g3["reward"].size,
              "len min/mean/max", g3["ep_length"].min(), g3["ep_length"].mean(), g3["ep_length"].max())
        g4 = record_rollouts(utils, key, 20, 4000 + ord(key[0]), None, KR[key], K, use_eval=True)
        np.savez_compressed(os.path.join(OUT, f"{key}_g4.npz"), **g4)
        print(key, "g4 steps", g4["reward"].size, bytes(g4["result_json"]).decode())
        meta[key] = {"K_step": K, "K_reset": KR[key]}
    with open(os.path.join(OUT, "META.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] in ("datasets", "g5", "g6", "agents", "laws", "info", "stats")):
    main()


HEAD = 900     # transitions kept per dataset: > 2 ChemicalReactor episodes, ~150 PowerGrid, ~450 RobotAssembly episodes


def gen_dataset_fixtures():
    """get_dataset() heads + summary statistics from a seeded reference run (np.random.seed(123))."""
    utils = load_reference()
    out = {}
    for key, name in ENVS.items():
        for q in (("expert", "medium") if key == "cr" else ("expert", "random", "mixed")):
            env = utils.make(name)
            np.random.seed(123)
            d = env.get_dataset(q)
            n = len(d["rewards"])
            out[f"{key}_{q}_n"] = np.array(n)
            out[f"{key}_{q}_reward_mean"] = np.array(float(d["rewards"].astype(np.float64).mean()))
            out[f"{key}_{q}_n_terminals"] = np.array(int(d["terminals"].sum()))
            for k in ("observations", "actions", "rewards", "terminals"):
                out[f"{key}_{q}_{k}"] = d[k][:HEAD]
            print(key, q, n, float(d["rewards"].mean()), int(d["terminals"].sum()))
    np.savez_compressed(os.path.join(OUT, "datasets.npz"), **out)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "datasets":
    gen_dataset_fixtures()


def gen_g5():
    """G5: the G1 inputs again, but with the action handed to the reference as float64
    (what get_dataset and the baseline agents do upstream): documents how far the float64-contaminated
    arithmetic moves the outputs from the float32-action semantics this build pins."""
    utils = load_reference()
    for key, name in ENVS.items():
        d = dict(np.load(os.path.join(OUT, f"{key}_g1.npz")))
        n = min(1200, len(d["reward"]))
        env = utils.make(name)
        K = d["noise"].shape[1]
        log = StepLog(env.state_dim, env.action_dim, K)
        with NoiseTap(5) as tap:
            env.reset(); tap.take()
            for i in range(n):
                force(env, d["state_pre"][i], int(d["step_pre"][i]), int(d["viol_pre"][i]))
                tap.forced = list(d["noise"][i])
                a64 = d["action"][i].astype(np.float64)
                state_pre = env.state.copy()
                bits = constraint_bits(env, state_pre, a64)
                tap.take()
                obs, reward, term, trunc, info = env.step(a64)
                noise = tap.take(); tap.forced = None
                sm = info["safety_metrics"]
                log.add(state_pre=state_pre, action=d["action"][i], noise=noise, step_pre=int(d["step_pre"][i]),
                        viol_pre=int(d["viol_pre"][i]), state_next=obs.copy(), reward=float(reward),
                        terminated=int(bool(term)), truncated=int(bool(trunc)), viol=sm.violation_count,
                        crit=sm.critical_violations, bits=bits, violations_after=info["violations"])
        arr = log.arrays()
        np.savez_compressed(os.path.join(OUT, f"{key}_g5.npz"), **arr)
        same = (arr["state_next"].view(np.uint32) == d["state_next"][:n].view(np.uint32)).mean()
        print(key, "g5", n, "state words identical to the float32-action run: %.4f" % same,
              "flag mismatches:", int((arr["terminated"] != d["terminated"][:n]).sum()))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "g5":
    gen_g5()


def gen_g6():
    """G6: the G1 states and noise again, stepped with GENUINE float64 action values (not float32-representable):
    the arithmetic the reference really performs for its own callers (get_dataset, baseline agents).  A third of
    the rows lie beyond the clip bounds.  Stored: the float64 actions and the reference's outputs."""
    utils = load_reference()
    rng = np.random.Generator(np.random.PCG64(66))
    for key, name in ENVS.items():
        d = dict(np.load(os.path.join(OUT, f"{key}_g1.npz")))
        n = min(1500, len(d["reward"]))
        env = utils.make(name)
        K = d["noise"].shape[1]
        log = StepLog(env.state_dim, env.action_dim, K)
        a64_rows = []
        with NoiseTap(6) as tap:
            env.reset(); tap.take()
            for i in range(n):
                force(env, d["state_pre"][i], int(d["step_pre"][i]), int(d["viol_pre"][i]))
                tap.forced = list(d["noise"][i])
                a64 = d["action"][i].astype(np.float64) * (1.3 if i % 3 == 0 else 0.9) + rng.normal(0.0, 1e-3, env.action_dim)
                assert a64.dtype == np.float64
                state_pre = env.state.copy()
                bits = constraint_bits(env, state_pre, a64)
                tap.take()
                obs, reward, term, trunc, info = env.step(a64)
                noise = tap.take(); tap.forced = None
                sm = info["safety_metrics"]
                a64_rows.append(a64.copy())
                log.add(state_pre=state_pre, action=a64.astype(f32), noise=noise, step_pre=int(d["step_pre"][i]),
                        viol_pre=int(d["viol_pre"][i]), state_next=obs.copy(), reward=float(reward),
                        terminated=int(bool(term)), truncated=int(bool(trunc)), viol=sm.violation_count,
                        crit=sm.critical_violations, bits=bits, violations_after=info["violations"])
        arr = log.arrays()
        arr["action64"] = np.asarray(a64_rows, dtype=np.float64)
        arr["reward_is_float64"] = np.array(isinstance(reward, (float, np.float64)) and not isinstance(reward, np.float32))
        np.savez_compressed(os.path.join(OUT, f"{key}_g6.npz"), **arr)
        print(key, "g6", n, "reward type", type(reward).__name__)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "g6":
    gen_g6()


def load_baseline_agents():
    """benchmarks/baseline_agents.py is NumPy-only; benchmarks/__init__ pulls modules that need scipy/jax, so it
    is bypassed with a shell package exactly like the root package."""
    import importlib
    import types
    load_reference()
    if "neorl_industrial.benchmarks" not in sys.modules:
        shell = types.ModuleType("neorl_industrial.benchmarks")
        shell.__path__ = ["/root/reference/src/neorl_industrial/benchmarks"]
        sys.modules["neorl_industrial.benchmarks"] = shell
    return importlib.import_module("neorl_industrial.benchmarks.baseline_agents")


def gen_agents():
    """act() sequences of the reference's baseline agents (benchmarks/baseline_agents.py:28-114) on recorded
    observations: Constant, MPC (proportional), PID (with its never-reset integral) and Random (draws recorded).
    Stored per env: the observation sequence (from the G3 rollouts, episode after episode, so the PID memory
    runs across episode boundaries as it does upstream) and every agent's action sequence."""
    ba = load_baseline_agents()
    out = {}
    for key, name in ENVS.items():
        g3 = dict(np.load(os.path.join(OUT, f"{key}_g3.npz")))
        obs = g3["obs"][:600].astype(f32)                       # [n, S] float32 observations as env.step returns them
        S, A = obs.shape[1], g3["action"].shape[1]
        setpoint = np.linspace(0.1, 0.5, A)
        agents = {
            "constant": ba.ConstantAgent(S, A, constant_action=np.linspace(-0.3, 0.4, A)),
            "mpc": ba.MPC_Agent(S, A),
            "pid": ba.PIDControllerAgent(S, A, kp=0.02, ki=0.001, kd=0.01, setpoint=setpoint),
        }
        out[f"{key}_obs"] = obs
        out[f"{key}_pid_setpoint"] = setpoint
        out[f"{key}_constant_action"] = np.linspace(-0.3, 0.4, A)
        for aname, ag in agents.items():
            acts = np.stack([np.asarray(ag.act(o)) for o in obs])
            out[f"{key}_{aname}_actions"] = acts
            out[f"{key}_{aname}_dtype"] = np.array(str(acts.dtype))
        rnd = ba.RandomAgent(S, A, action_low=-1.0, action_high=1.0)
        with NoiseTap(17) as tap:
            acts = np.stack([np.asarray(rnd.act(o)) for o in obs[:64]])
            draws = tap.take()
        out[f"{key}_random_actions"] = acts
        out[f"{key}_random_draws"] = draws.reshape(64, A)
        print(key, {k: str(v.dtype) for k, v in out.items() if k.startswith(key) and k.endswith("actions")})
    np.savez_compressed(os.path.join(OUT, "baseline_agents.npz"), **out)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "agents":
    gen_agents()


LAW_ROWS = 640     # (observation, action) pairs kept per (env, quality)


def gen_laws():
    """The action LAW of env.get_dataset(quality) (chemical_reactor.py:364-393, power_grid.py:216-233,
    robot_assembly.py:266-290) on >= 512 observations of the reference's own data-collection run, with the law's own
    random draws patched to zero: np.random.normal / uniform called FROM get_dataset return 0, np.random.random / rand
    called from it return 0.0 (so the epsilon-mixtures take their feedback branch).  The env's reset and process noise
    (drawn inside step / reset, other frames) stays random, so the observations spread.  Stored per (env, quality): the
    float32 observations of the dataset and the actions the reference stored for them (for ChemicalReactor the clipped
    action, for PowerGrid / RobotAssembly the action as computed -- the reference's own conventions)."""
    utils = load_reference()
    real = {k: getattr(np.random, k) for k in ("normal", "uniform", "random", "rand")}

    def from_law():
        return sys._getframe(2).f_code.co_name == "get_dataset"

    def normal(loc=0.0, scale=1.0, size=None):
        if from_law():
            return 0.0 if size is None else np.zeros(size)
        return real["normal"](loc, scale, size)

    def uniform(low=0.0, high=1.0, size=None):
        if from_law():
            return 0.0 if size is None else np.zeros(size)
        return real["uniform"](low, high, size)

    def random(*a):
        return 0.0 if from_law() else real["random"](*a)

    def rand(*a):
        return 0.0 if from_law() else real["rand"](*a)

    out = {}
    np.random.normal, np.random.uniform, np.random.random, np.random.rand = normal, uniform, random, rand
    try:
        for key, name in ENVS.items():
            for q in ("expert", "medium", "mixed", "random"):
                env = utils.make(name)
                np.random.seed(777)
                d = env.get_dataset(q)
                n = min(LAW_ROWS, len(d["actions"]))
                # spread over the run rather than its head (PowerGrid / RobotAssembly episodes are a few steps long)
                idx = np.linspace(0, len(d["actions"]) - 1, n).astype(np.int64)
                out[f"{key}_{q}_obs"] = d["observations"][idx].astype(f32)
                out[f"{key}_{q}_act"] = d["actions"][idx].astype(f32)
                print(key, q, n, "of", len(d["actions"]), "|a| max", float(np.abs(d["actions"]).max()))
    finally:
        np.random.normal, np.random.uniform, np.random.random, np.random.rand = (real[k] for k in ("normal", "uniform", "random", "rand"))
    np.savez_compressed(os.path.join(OUT, "behaviour_laws.npz"), **out)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "laws":
    gen_laws()


def gen_info():
    """ChemicalReactorEnv._get_safety_info (chemical_reactor.py:307-322): the reset-time info dict and the info dict of
    every step along a run under np.random.seed(4321) with a fixed float32 action sequence: safety_metrics sub-dict at
    reset (the five state values), constraint_values (temp_margin, pressure_margin, level_in_bounds) at reset and per
    step, step / violations / total_violations / critical_shutdown.  Two episodes (the second exercises reset's
    bookkeeping: violations back to 0, total_violations kept)."""
    utils = load_reference()
    env = utils.make("ChemicalReactor-v0")
    rng = np.random.Generator(np.random.PCG64(99))
    rows = {k: [] for k in ("obs", "action", "temp_margin", "pressure_margin", "level_in_bounds", "step", "violations",
                            "total_violations", "critical_shutdown", "is_reset", "sm_reset")}
    dtypes = {}
    np.random.seed(4321)
    for ep in range(2):
        obs, info = env.reset()
        cv, sm = info["constraint_values"], info["safety_metrics"]
        dtypes["temp_margin"] = type(cv["temp_margin"]).__name__
        dtypes["level_in_bounds"] = type(cv["level_in_bounds"]).__name__
        rows["obs"].append(obs.copy()); rows["action"].append(np.zeros(3, dtype=f32)); rows["is_reset"].append(1)
        rows["sm_reset"].append([sm["temperature"], sm["pressure"], sm["level"], sm["emergency_stop"], sm["alarm_status"]])
        for k in ("temp_margin", "pressure_margin"):
            rows[k].append(float(cv[k]))
        rows["level_in_bounds"].append(int(bool(cv["level_in_bounds"])))
        rows["step"].append(info["step"]); rows["violations"].append(info["violations"])
        rows["total_violations"].append(info["total_violations"]); rows["critical_shutdown"].append(0)
        for t in range(400):
            a = np.clip(rng.normal(0.0, 0.8, 3), -1, 1).astype(f32) + (f32(0.9) if ep == 1 else f32(0))    # episode 2 is driven hot
            obs, reward, term, trunc, info = env.step(a)
            cv = info["constraint_values"]
            rows["obs"].append(obs.copy()); rows["action"].append(a); rows["is_reset"].append(0)
            rows["sm_reset"].append([0, 0, 0, 0, 0])
            for k in ("temp_margin", "pressure_margin"):
                rows[k].append(float(cv[k]))
            rows["level_in_bounds"].append(int(bool(cv["level_in_bounds"])))
            rows["step"].append(info["step"]); rows["violations"].append(info["violations"])
            rows["total_violations"].append(info["total_violations"]); rows["critical_shutdown"].append(int(bool(info["critical_shutdown"])))
            if term or trunc:
                break
    out = {"obs": np.asarray(rows["obs"], dtype=f32), "action": np.asarray(rows["action"], dtype=f32),
           "temp_margin": np.asarray(rows["temp_margin"], dtype=np.float64),
           "pressure_margin": np.asarray(rows["pressure_margin"], dtype=np.float64),
           "level_in_bounds": np.asarray(rows["level_in_bounds"], dtype=np.uint8),
           "step": np.asarray(rows["step"], dtype=np.int32), "violations": np.asarray(rows["violations"], dtype=np.int32),
           "total_violations": np.asarray(rows["total_violations"], dtype=np.int32),
           "critical_shutdown": np.asarray(rows["critical_shutdown"], dtype=np.uint8),
           "is_reset": np.asarray(rows["is_reset"], dtype=np.uint8), "sm_reset": np.asarray(rows["sm_reset"], dtype=f32),
           "margin_type": np.array(dtypes["temp_margin"]), "bounds_type": np.array(dtypes["level_in_bounds"])}
    np.savez_compressed(os.path.join(OUT, "cr_info.npz"), **out)
    print("cr_info rows", len(out["step"]), "resets", int(out["is_reset"].sum()), "shutdown steps", int(out["critical_shutdown"].sum()),
          "types", dtypes, "total_violations at end", int(out["total_violations"][-1]))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "info":
    gen_info()


STATS_EPISODES = {"cr": 24000, "pg": 160000, "ra": 160000}


def _stats_env(key):
    """One env's share of gen_stats (a process of its own: the reference's draws are module-global np.random state)."""
    utils = load_reference()
    name = ENVS[key]
    env = utils.make(name)
    A = env.action_dim
    n = STATS_EPISODES[key]
    np.random.seed(20250 + ord(key[0]))                         # the reference's own generator, its own call order
    arng = np.random.Generator(np.random.PCG64(7 + ord(key[1])))  # action_space.sample(): gymnasium draws from its own stream
    length = np.zeros(n, np.int32); ret = np.zeros(n); viol = np.zeros(n, np.int32); crit = np.zeros(n, np.int32)
    cbits = np.zeros((n, 3), np.int32); cause = np.zeros(n, np.uint8)
    for e in range(n):
        env.reset()
        R, T, ncrit = 0.0, 0, 0
        while True:
            a = arng.uniform(-1.0, 1.0, A).astype(f32)
            bits = constraint_bits(env, env.state, a)          # the reference's own check_fn on the pre-step state
            obs, reward, term, trunc, info = env.step(a)
            sm = info["safety_metrics"]
            assert sm.violation_count == 3 - sum(bits)
            R += float(reward); T += 1; ncrit += sm.critical_violations
            for k in range(3):
                cbits[e, k] += 1 - bits[k]
            if term or trunc:
                break
        length[e], ret[e], viol[e], crit[e] = T, R, info["violations"], ncrit
        cause[e] = (1 if term else 0) | (2 if trunc else 0) | (4 if info["critical_shutdown"] else 0)
    print(key, n, "episodes: length mean %.2f median %d max %d | viol/ep %.3f crit/ep %.4f | ret mean %.2f | term %.4f trunc %.4f shutdown %.4f"
          % (length.mean(), np.median(length), length.max(), viol.mean(), crit.mean(), ret.mean(),
             (cause & 1).astype(bool).mean(), (cause & 2).astype(bool).mean(), (cause & 4).astype(bool).mean()), flush=True)
    # compact dtypes: lengths <= 1000, counts <= 3000, returns to float32 (the statistics are means over thousands of episodes)
    assert length.max() <= 65535 and viol.max() <= 65535 and cbits.max() <= 65535 and crit.max() <= 255
    return {f"{key}_length": length.astype(np.uint16), f"{key}_ret": ret.astype(np.float32), f"{key}_viol": viol.astype(np.uint16),
            f"{key}_crit": crit.astype(np.uint8), f"{key}_cbits": cbits.astype(np.uint16), f"{key}_cause": cause}


def gen_stats():
    """The workload the driver times, run by the REFERENCE: performance_benchmark.py:106-133's loop (uniform float32
    actions in [-1, 1] = action_space.sample(), reset when terminated or truncated) with the reference's own
    np.random draws (chemical_reactor.py:93-103,149,159, power_grid.py:98-108,136-144, robot_assembly.py:118-122;
    MT19937 seeded per env).  Stored PER EPISODE, so a test can form any statistic and its sampling error:
      length, ret (sum of rewards), viol (info['violations'] at the end = sum of per-step violation counts,
      base.py:179-183), crit (sum of per-step critical_violations), cbits [3] (steps on which constraint k failed),
      cause (bit 0 terminated, bit 1 truncated, bit 2 critical shutdown: the last step's flags, base.py:186-196).
    Nothing here depends on this build: the fast-mode generator of the device path has to reproduce these
    DISTRIBUTIONS (tests/test_reference_stats.py, tests/test_gpu_reference_stats.py, bench.py's parity block).
    Round 5 grew the sample fourfold (24 000 / 160 000 / 160 000 episodes: standard errors halve); one process per env."""
    import multiprocessing as mp
    out = {"numpy": np.array(np.__version__)}
    with mp.get_context("fork").Pool(3) as pool:
        for part in pool.map(_stats_env, list(ENVS)):
            out.update(part)
    np.savez_compressed(os.path.join(OUT, "reference_stats.npz"), **out)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "stats":
    gen_stats()
