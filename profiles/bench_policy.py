#!/usr/bin/env python3
"""Closed-loop throughput with a device-resident policy (GPU box): env-steps/s for
  mlp-torch   the reference actor shape (S->256->256->A) via torch GEMMs + one step kernel per step
  affine      an on-device affine policy inside the fused rollout kernel
usage: python profiles/bench_policy.py [--batch 65536] [--steps 200]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import neorl_industrial_gym_amd as ni

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--env", default="ChemicalReactor-v0")
ap.add_argument("--mlp-launches", type=int, default=1,
                help="timed launches of the fused MFMA actor kernel (each --mlp-steps env.step per lane): >= 50 gives the "
                     "profiler's kernel stats something to average (VERDICT r02 weak #8)")
ap.add_argument("--mlp-steps", type=int, default=0, help="env.step per MFMA-actor launch (default: --steps)")
ap.add_argument("--only-mlp", action="store_true", help="skip the torch-GEMM and affine-policy measurements")
args = ap.parse_args()
B, T = args.batch, args.steps
env = ni.make_batched(args.env, B, autoreset=True, tally=True)
S, A = env.state_dim, env.action_dim
rng = np.random.default_rng(0)
ws = [(rng.normal(0, 0.02 / np.sqrt(S), (S, 256)), np.zeros(256)), (rng.normal(0, 1 / 16, (256, 256)), np.zeros(256)),
      (rng.normal(0, 1 / 16, (256, A)), np.zeros(A))]
pol = ni.MLPPolicy(ws)
out = {}
env.reset()
if not args.only_mlp:
    for _ in range(20):
        env.step(pol.predict_device(env.obs), layout="aos")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(T):
        env.step(pol.predict_device(env.obs), layout="aos")
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out["mlp_torch"] = {"env_steps_per_s": B * T / dt, "us_per_step": dt / T * 1e6}
env.set_mlp_policy([(w.astype(np.float32), b.astype(np.float32)) for w, b in ws])
Tm, Nm = (args.mlp_steps or T), max(1, args.mlp_launches)
env.rollout_mlp(10)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(Nm):
    env.rollout_mlp(Tm)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
flops = 2.0 * (S * 256 + 256 * 256 + 256 * A)
out["mlp_mfma_fused"] = {"env_steps_per_s": B * Tm * Nm / dt, "us_per_step": dt / (Tm * Nm) * 1e6, "launches": Nm, "steps_per_launch": Tm,
                         "us_per_launch": dt / Nm * 1e6, "flop_per_env_step": flops, "actor_TFLOPs": B * Tm * Nm * flops / dt / 1e12}
if args.only_mlp:
    print(json.dumps(out))
    sys.exit(0)
env.set_policy(ni.behaviour_policy(args.env, "expert"))
env.rollout_policy(50)
torch.cuda.synchronize(); t0 = time.perf_counter()
env.rollout_policy(T)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
out["affine_fused"] = {"env_steps_per_s": B * T / dt, "us_per_step": dt / T * 1e6}
# the same closed loop writing the D4RL transition stream (observation, action, reward, flag word per step)
Tf = min(T, 100)
obs = torch.empty(Tf, B, S, dtype=torch.float32, device=env.device)
act = torch.empty(Tf, A, env.ld, dtype=torch.float32, device=env.device)
rew = torch.empty(Tf, env.ld, dtype=torch.float32, device=env.device)
fl = torch.empty(Tf, env.ld, dtype=torch.int32, device=env.device)
env.rollout_policy(Tf, rew, fl, obs, act)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(4):
    env.rollout_policy(Tf, rew, fl, obs, act)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
out["affine_fused_transitions"] = {"env_steps_per_s": 4 * B * Tf / dt, "us_per_step": dt / (4 * Tf) * 1e6,
                                   "GBps_written": 4 * B * Tf * (4 * S + 4 * A + 8) / dt / 1e9}
print(json.dumps(out))
