"""The two Advanced envs.  PARITY UNPINNED: upstream these classes cannot
be instantiated and their step() reads an attribute nothing sets, so no reference output exists.
What is checked instead:
  not-gpu  the C oracle against an independent vectorised NumPy-float32 restatement of the same
           source text (a different code shape written separately; NEP-50 weak scalars make the
           NumPy arithmetic float32 exactly like JAX with x64 off), plus behavioural facts read
           straight off the source;
  gpu      the HIP kernels against the oracle, bit for bit (tests/test_gpu_parity.py style).
"""
import math

import numpy as np
import pytest
import torch

from conftest import rel_err

f32 = np.float32


def np_acr_step(s, a, step_pre, max_steps, dt=f32(0.1)):
    """advanced_chemical_reactor.py:195-450, rows of float32."""
    s = s.astype(f32); a = a.astype(f32)
    T, Tj, cA, cB, cC, cD = s[:, 0], s[:, 1], s[:, 3], s[:, 4], s[:, 5], s[:, 6]
    Ff, Fp, Fc, hc, mix, Tw = s[:, 7], s[:, 8], s[:, 9], s[:, 10], s[:, 11], s[:, 12:16]
    es = a[:, 5] > 0.5
    feed = np.where(es, f32(0), a[:, 0]); cool = np.where(es, f32(0.01), a[:, 1]); rpm = np.where(es, f32(0), a[:, 2])
    nFf = Ff + 0.1 * (feed - Ff)
    nFc = Fc + 0.2 * (cool - Fc)
    k = 1e8 * np.exp(-8.314e4 / (8.314 * T))
    rr = k * cA * cB * mix
    V = 1.0
    dA = (nFf * 5.0 - Fp * cA) / V - rr
    dB = (nFf * 3.0 - Fp * cB) / V - rr
    dC = -Fp * cC / V + rr
    dD = -Fp * cD / V + rr
    Qgen = 5e4 * rr * V
    Aj = 4.0 * math.pi * (1.0 / (4 / 3 * math.pi)) ** (2 / 3)
    Qj = hc * Aj * (T - Tj)
    Qw = np.zeros_like(T)
    for i in range(4):
        Qw = Qw + (50.0 * (Aj / 4) / 0.01) * (T - Tw[:, i])
    Qf = nFf * 1000.0 * 4180.0 * (a[:, 3] - T)
    dTr = (Qgen - Qj - Qw + Qf) / (1000.0 * 1.0 * 4180.0)
    dTj = (Qj - nFc * 1000.0 * 4180.0 * (Tj - 293.15)) / (100.0 * 4180.0)
    nTw = np.stack([Tw[:, i] + dt * (((50.0 / 0.01) * (T - Tw[:, i]) - 10.0 * (Tw[:, i] - 293.15)) / (50.0 * 500.0))
                    for i in range(4)], axis=1)
    moles = (cA + cB + cC + cD) * V
    vp = 1000.0 * np.exp(20.0 - 5000.0 / T)
    nP = 8.314 * T * moles / V + vp + 1e5
    relief = a[:, 4] / 100.0 * (nP - 3e6 * 0.8)
    nP = np.where(nP > 3e6 * 0.8, nP - relief, nP)
    nmix = np.tanh(rpm / 1000.0) * 0.9 + 0.1
    Re = rpm * 0.1 * 1000.0 / 0.001
    with np.errstate(divide="ignore"):
        nhc = 0.023 * np.where(Re > 0, Re ** f32(0.8), f32(0)) * 0.6 / 0.1
    nFp = 0.001 * (1.0 + 0.5 * ((nP - 1e5) / 1e5))
    nA, nB = np.maximum(0.0, cA + dt * dA), np.maximum(0.0, cB + dt * dB)
    nC, nD = np.maximum(0.0, cC + dt * dC), np.maximum(0.0, cD + dt * dD)
    nT, nTj = T + dt * dTr, Tj + dt * dTj
    tau = V / np.maximum(nFp, 1e-6)
    conv = (2.0 - nA) / 2.0
    mT = (673.15 - nT) / 673.15 * 100.0
    mP = (5e6 - nP) / 5e6 * 100.0
    nxt = np.stack([nT, nTj, nP, nA, nB, nC, nD, nFf, nFp, nFc, nhc, nmix, nTw[:, 0], nTw[:, 1], nTw[:, 2], nTw[:, 3],
                    tau, conv, mT, mP], axis=1).astype(f32)
    rew = (100.0 * (nC / 5.0 + conv) + (mT + mP) / 2.0
           + 50.0 * ((1.0 - np.abs(nT - 373.15) / 100.0) + (1.0 - np.abs(nP - 3e5) / 1e5))
           + -(np.abs(a[:, 0]) + np.abs(a[:, 1]) + np.abs(a[:, 2]) + np.abs(a[:, 3]) + np.abs(a[:, 4])) * 10.0
           + np.where(es, f32(-1000.0), f32(0.0)))
    mask = (nT > 673.15) * 1 + (nP > 5e6) * 2 + (mT < 10.0) * 4 + (mP < 10.0) * 8
    term = (nT > 673.15) | (nP > 5e6) | (nC > 8.0)
    return nxt, rew.astype(f32), term, np.asarray(step_pre) >= max_steps, mask, es


def np_apg_step(s, a, step_pre, max_steps, dt=f32(0.1)):
    """advanced_power_grid.py:228-537."""
    s = s.astype(f32); a = a.astype(f32)
    H = np.array([5.0, 4.0, 3.5, 4.5], dtype=f32); D = np.array([1.0, 0.8, 0.9, 1.1], dtype=f32)
    Pmax = np.array([50, 40, 35, 45], dtype=f32); Pmin = np.array([10, 8, 7, 9], dtype=f32)
    ramp = np.array([2.0, 1.8, 1.5, 2.2], dtype=f32); bl0 = np.array([25, 20, 30, 18], dtype=f32)
    al = np.array([1.5, 1.2, 1.8, 1.3], dtype=f32); Kf = np.array([1.0, 0.8, 1.2, 0.9], dtype=f32)
    V, th, f, Pg = s[:, 0:8], s[:, 8:16], s[:, 16:20], s[:, 20:24]
    em = a[:, 7] > 0.5
    sp = np.where(em[:, None], a[:, 0:4] * 0.7, a[:, 0:4])
    shed = np.where(em, np.minimum(a[:, 6] + 10.0, 30.0), a[:, 6])
    df = (sp / 100.0 - Pg / 100.0 - D * (f - 50.0)) / (2 * H)
    nf = f + dt * df
    fsum = np.zeros(len(s), dtype=f32)
    for i in range(4):
        fsum = fsum + nf[:, i] * H[i]
    fsys = fsum / f32(17.0)
    nPg = np.clip(Pg + np.clip(sp - Pg, -(ramp * dt), ramp * dt), Pmin, Pmax)
    fdev = (fsys - 50.0) / 50.0
    bl = np.tile(bl0, (len(s), 1)); bl[:, 0] = np.maximum(bl[:, 0] - shed, 0.0)
    nL = bl * (V[:, 0:4] / 1.0) ** al * (1.0 + Kf * fdev[:, None])
    inj = np.concatenate([nPg / 100.0, -nL / 100.0], axis=1)
    nV = V + 0.01 * inj
    nV[:, 0] = a[:, 4]; nV[:, 1] = a[:, 5]
    nV = np.clip(nV, 0.8, 1.2)
    nth = th + 0.05 * inj
    flow = nV[:, 0:4] * nV[:, 4:8] / 0.1 * np.sin(nth[:, 0:4] - nth[:, 4:8]) * 100.0
    dv = np.abs(nV - 1.0)
    vmean = np.zeros(len(s), dtype=f32)
    for i in range(8):
        vmean = vmean + dv[:, i]
    vmean = vmean / f32(8.0)
    stab = np.maximum(np.minimum(np.minimum(1.0 - dv.max(1), 1.0 - (nth.max(1) - nth.min(1)) / math.pi),
                                 1.0 - np.abs(nf - 50.0).max(1) / 0.5), 0.0)
    nxt = np.concatenate([nV, nth, nf, nPg, nL, flow], axis=1).astype(f32)
    ferr = np.abs(fsys - 50.0)
    tg = nPg[:, 0] + nPg[:, 1] + nPg[:, 2] + nPg[:, 3]; tl = nL[:, 0] + nL[:, 1] + nL[:, 2] + nL[:, 3]
    cost = nPg[:, 0] ** 2 + nPg[:, 1] ** 2 + nPg[:, 2] ** 2 + nPg[:, 3] ** 2
    ctl = np.abs(a[:, 0]) + np.abs(a[:, 1]) + np.abs(a[:, 2]) + np.abs(a[:, 3]) + np.abs(a[:, 4]) + np.abs(a[:, 5])
    rew = (100.0 * np.exp(-ferr / 0.1) + 50.0 * np.exp(-vmean / 0.05) + 30.0 * np.exp(-np.abs(tg - tl) / 10.0)
           + -(0.01 * cost) + -ctl * 1.0 + -a[:, 6] * 50.0 + -a[:, 7] * 200.0)
    vviol = (dv > 0.05).any(1)
    mask = (ferr > 0.5) * 1 + vviol * 2 + ((nPg < Pmin) | (nPg > Pmax)).any(1) * 4
    term = (ferr > 0.5) | vviol | (stab < 0.1)
    return nxt, rew.astype(f32), term, np.asarray(step_pre) >= max_steps, mask, em


def _sample(key, n, seed, oracle):
    """States reached by random-action rollouts + broad perturbations; actions inside and outside the Box."""
    rng = np.random.default_rng(seed)
    sp = oracle.spec(key)
    st, _, _, _ = oracle.rollout(key, n, int(rng.integers(1, 30)), seed=seed, flavor=oracle.MATH_LIBM)
    st = st * (1.0 + rng.normal(0, 0.02, st.shape)).astype(f32)
    lo = np.array({"acr": [0, 0, 0, 273.15, 0, 0], "apg": [10, 8, 7, 9, .95, .95, 0, 0]}[key], dtype=f32)
    hi = np.array({"acr": [.01, .01, 3000, 473.15, 100, 1], "apg": [50, 40, 35, 45, 1.05, 1.05, 20, 1]}[key], dtype=f32)
    act = (lo + (hi - lo) * rng.uniform(-0.1, 1.1, (n, sp.action_dim))).astype(f32)
    if key == "acr":
        st[: n // 8, 0] = rng.uniform(600, 700, n // 8)           # near the temperature limit
        st[n // 8: n // 4, 3:7] *= 8                               # high inventory -> pressure relief branch
        act[::7, 2] = 0.0                                          # Re = 0 -> 0 ** 0.8
    step_pre = rng.integers(0, sp.max_episode_steps + 2, n).astype(np.int32)
    return st.astype(f32), act, step_pre


@pytest.mark.parametrize("key", ["acr", "apg"])
@pytest.mark.parametrize("flavor", [0, 1])
def test_oracle_matches_independent_numpy_restatement(oracle, key, flavor):
    n = 3000
    st, act, step_pre = _sample(key, n, 17, oracle)
    sp = oracle.spec(key)
    o = oracle.step(key, st, act, None, step_pre, flavor=flavor)
    fn = np_acr_step if key == "acr" else np_apg_step
    nxt, rew, term, trunc, mask, shut = fn(st, act, step_pre, sp.max_episode_steps)
    ok = np.isfinite(nxt).all(1) & np.isfinite(o["state_next"]).all(1)
    assert ok.mean() > 0.95
    assert rel_err(o["state_next"][ok], nxt[ok], floor=1e-4).max() <= 2e-5
    # the reward is a sum of exponentials as steep as exp(-|f - 50| / 0.1) (advanced_power_grid.py:436-482): one float32
    # ulp in the exponent's argument is up to ~1e-5 relative in that term, on top of the different exp implementations
    assert rel_err(o["reward"][ok], rew[ok], floor=1.0).max() <= 5e-5
    # discrete outputs agree except where a threshold sits inside that rounding band
    for name, got, want in (("terminated", o["terminated"], term), ("truncated", o["truncated"], trunc),
                            ("mask", o["viol_mask"], mask), ("shutdown", o["shutdown"], shut)):
        assert (np.asarray(got)[ok] != np.asarray(want)[ok]).mean() < 2e-3, name
    assert np.array_equal(o["viol"], [bin(int(m)).count("1") for m in o["viol_mask"]]) and not o["crit"].any()


def test_reset_states_and_source_facts(oracle):
    s = oracle.reset("acr", np.zeros((1, 1)))[0]
    assert np.allclose(s, [323.15, 313.15, 2e5, 2.0, 1.5, 0.1, 0.1, 0.001, 0.001, 0.005, 300.0, 0.8,
                           323.15, 323.15, 323.15, 323.15, 1000.0, 0.05, 50.0, 60.0])       # :163-191
    g = oracle.reset("apg", np.zeros((1, 1)))[0]
    assert np.allclose(g[:8], 1.0) and np.allclose(g[16:20], 50.0) and np.allclose(g[20:24], [30, 25, 20, 28])   # :186-224
    a = np.array([[0.001, 0.005, 500.0, 323.15, 0.0, 0.0]], dtype=f32)
    base = oracle.step("acr", s[None], a, None, [0])
    a2 = a.copy(); a2[0, 5] = 1.0                                   # emergency shutdown
    shut = oracle.step("acr", s[None], a2, None, [0])
    assert shut["shutdown"][0] == 1 and base["shutdown"][0] == 0
    assert shut["reward"][0] < base["reward"][0] - 900              # -1000 penalty, :394
    assert shut["state_next"][0][11] == pytest.approx(0.1)          # rpm forced to 0 -> mixing 0.1, :223,300
    assert shut["state_next"][0][10] == 0.0                         # Re = 0 -> h = 0, :301-303
    # truncation uses episode_step BEFORE its increment (:351): step index 1000 truncates, 999 does not
    assert oracle.step("acr", s[None], a, None, [999])["truncated"][0] == 0
    assert oracle.step("acr", s[None], a, None, [1000])["truncated"][0] == 1
    # power grid: voltage regulators overwrite buses 0/1 (:379-380); 0.94 p.u. ends the episode (:495)
    ag = np.array([[30, 25, 20, 28, 0.94, 1.0, 0, 0]], dtype=f32)
    r = oracle.step("apg", g[None], ag, None, [0])
    assert r["state_next"][0][0] == f32(0.94) and r["terminated"][0] == 1 and (r["viol_mask"][0] & 2)
    ag[0, 4] = 1.0
    r = oracle.step("apg", g[None], ag, None, [0])
    assert r["terminated"][0] == 0 and r["viol_mask"][0] == 0


# ---------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("key,name", [("acr", "AdvancedChemicalReactor-v0"), ("apg", "AdvancedPowerGrid-v0")])
def test_gpu_bit_identical_to_oracle(oracle, key, name):
    import neorl_industrial_gym_amd as ni
    n = 4000
    st, act, step_pre = _sample(key, n, 23, oracle)
    env = ni.make_batched(name, n, autoreset=False)
    env.set_state(st, current_step=np.minimum(step_pre, 16000))
    _, rew, term, trunc, info = env.step(act, layout="aos")
    o = oracle.step(key, st, act, None, np.minimum(step_pre, 16000), flavor=oracle.MATH_POLY)
    got = env.get_state().cpu().numpy()
    fin = np.isfinite(o["state_next"]).all(1)
    assert np.array_equal(got.view(np.uint32)[fin], o["state_next"].view(np.uint32)[fin])
    assert np.array_equal(rew.cpu().numpy().view(np.uint32)[fin], o["reward"].astype(f32).view(np.uint32)[fin])
    assert np.array_equal(term.cpu().numpy()[fin], o["terminated"][fin] != 0)
    assert np.array_equal(trunc.cpu().numpy()[fin], o["truncated"][fin] != 0)
    assert np.array_equal(info.violation_count.cpu().numpy()[fin], o["viol"][fin])
    assert np.array_equal(info.critical_shutdown.cpu().numpy()[fin], o["shutdown"][fin] != 0)
    cv = info.constraint_violated.cpu().numpy()
    mask = sum(cv[k].astype(int) << k for k in range(cv.shape[0]))
    assert np.array_equal(mask[fin], o["viol_mask"][fin])
    env.close()
    # free-running fast mode with auto-reset (step kernel and fused rollout) vs the oracle
    B, T = 8192, 60
    a = ni.make_batched(name, B, autoreset=True, tally=True)
    b = ni.make_batched(name, B, autoreset=True, tally=True)
    ring = torch.zeros(T, a.action_dim, a.ld, dtype=torch.float32, device=a.device)
    for t in range(T):
        a.fill_actions(t + 1, ring[t])
    rw = torch.zeros(T, a.ld, dtype=torch.float32, device=a.device); fl = torch.zeros(T, a.ld, dtype=torch.int32, device=a.device)
    a.reset(); b.reset()
    a.rollout(T, ring, rw, fl)
    for t in range(T):
        b.step(ring[t][:, :B], layout="soa")
    s_ref, c_ref, tot, _ = oracle.rollout(key, B, T, flavor=oracle.MATH_POLY, nthreads=8)
    assert np.array_equal(a.get_state().cpu().numpy().view(np.uint32), s_ref.view(np.uint32))
    assert torch.equal(a.state_soa.view(torch.int32), b.state_soa.view(torch.int32)) and torch.equal(a.ctr, b.ctr)
    assert np.array_equal(a.current_step.cpu().numpy(), c_ref)
    L = ni._lib
    nv = ((fl[:, :B] >> L.FLAG_NVIOL_SHIFT) & 3) + ((fl[:, :B] >> 13) & 1) * 4
    assert int(nv.sum().item()) == tot.violations and int(a.tally[L.T_EPISODES].sum().item()) == tot.episodes
    a.close(); b.close()


@pytest.mark.gpu
def test_single_env_surface_advanced():
    import neorl_industrial_gym_amd as ni
    env = ni.make("AdvancedChemicalReactor-v0")
    obs, info = env.reset()
    assert obs.shape == (20,) and obs[0] == f32(323.15)
    assert env.action_space.high[2] == 3000.0 and len(env.safety_constraints) == 4
    obs, r, te, tr, info = env.step(np.array([0.001, 0.005, 500.0, 323.15, 0.0, 0.0], dtype=f32))
    assert isinstance(r, float) and not te and not tr
    assert set(info) >= {"conversion", "residence_time", "safety_metrics", "emergency_shutdown", "violation_types"}
    assert info["safety_metrics"].total_constraints == 4 and info["violation_types"] == []
    obs, r2, te, tr, info = env.step(np.array([0.001, 0.005, 500.0, 323.15, 0.0, 1.0], dtype=f32))
    assert info["emergency_shutdown"] and info["critical_shutdown"] and r2 < r - 900
    env.close()
    g = ni.make("AdvancedPowerGrid-v0")
    obs, _ = g.reset()
    obs, r, te, tr, info = g.step(np.array([30, 25, 20, 28, 0.94, 1.0, 0, 0], dtype=f32))
    assert te and info["violation_types"] == ["voltage_deviation"] and abs(info["system_frequency"] - 50.0) < 0.1
    g.close()
