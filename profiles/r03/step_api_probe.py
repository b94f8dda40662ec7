#!/usr/bin/env python3
"""Step API (one step_kernel launch per env.step, 250 per hipGraph replay) at the headline size, by handle options:
what do the episode tally and the in-kernel auto-reset cost per launch?  (profiles/ubench/step_floor.hip has the
floors: an empty graph node, the memory shape without arithmetic.)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import neorl_industrial_gym_amd as ni

B, P, R = 65536, 250, 64
out = {}
ENVS = [e for e in ("ChemicalReactor-v0", "PowerGrid-v0", "RobotAssembly-v0") if len(sys.argv) < 2 or e.startswith(sys.argv[1])]
for env_id in ENVS:
    # (autoreset=0 is no baseline for the cost of auto-reset: finished lanes stay frozen, after a few hundred steps the launch is idle)
    for tally, autoreset in ((True, True), (False, True), (False, False)):
        env = ni.make_batched(env_id, B, autoreset=autoreset, tally=tally)
        ring = torch.empty(R, env.action_dim, env.ld, dtype=torch.float32, device=env.device)
        for s in range(R):
            env.fill_actions(1000 + s, ring[s])
        env.reset()
        plan = env.make_plan(P, ring, env.reward, env.flags)
        for _ in range(4):
            plan.launch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 16
        for _ in range(n):
            plan.launch()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / (n * P) * 1e6
        out[f"{env_id} tally={int(tally)} autoreset={int(autoreset)}"] = round(us, 3)
        plan.close(); env.close()
print(json.dumps(out, indent=1))
