#!/usr/bin/env python3
"""The reference's own harness shape (performance_benchmark.py:106-133) on the single-env drop-in class:
1 env, N steps, action_space.sample(), reset on done.  Launch/PCIe-latency bound by construction."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import neorl_industrial_gym_amd as ni

for name in ("ChemicalReactor-v0", "PowerGrid-v0", "RobotAssembly-v0"):
    env = ni.make(name)
    obs, _ = env.reset()
    n = 3000
    for _ in range(200):
        obs, r, te, tr, info = env.step(env.action_space.sample())
        if te or tr:
            env.reset()
    t0 = time.perf_counter()
    for _ in range(n):
        obs, r, te, tr, info = env.step(env.action_space.sample())
        if te or tr:
            obs, _ = env.reset()
    dt = time.perf_counter() - t0
    print(f"{name}: {n / dt:,.0f} steps/s  ({dt / n * 1e6:.1f} us/step)")
    env.close()
