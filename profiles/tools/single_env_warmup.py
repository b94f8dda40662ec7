#!/usr/bin/env python3
"""Why is the single-env class 60 us per step in a fresh process and 21 us later?  Blocks of 500 steps on ONE env object, then on a
second env object made after the first is closed, then after a burst of heavy GPU work."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import neorl_industrial_gym_amd as ni


def blocks(env, n, tag):
    a = np.zeros(3, dtype=np.float32)
    for b in range(n):
        t0 = time.perf_counter()
        for _ in range(500):
            obs, r, te, tr, info = env.step(a)
            if te or tr:
                env.reset()
        print("%s block %d: %.1f us per step" % (tag, b, (time.perf_counter() - t0) / 500 * 1e6), flush=True)


env = ni.make("ChemicalReactor-v0")
env.reset()
blocks(env, 6, "first env")
env.close()
env = ni.make("ChemicalReactor-v0")
env.reset()
blocks(env, 3, "second env")
x = torch.empty(1 << 28, device="cuda")
for _ in range(50):
    x.fill_(1.0)
torch.cuda.synchronize()
blocks(env, 3, "after 50 fills of 1 GiB")
time.sleep(2.0)
blocks(env, 3, "after 2 s idle")
env.close()
