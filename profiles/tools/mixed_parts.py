#!/usr/bin/env python3
"""Where the mixed launch's time goes: mixed_rollout_kernel<1> (reward + flags) over ONE env type at a time, the whole
batch (default 1 048 576 lanes) in that env -- the env's body under the mixed kernel's register allocation with the chip
full.  The 7-env launch holds 1/7 of its lanes in each env, so sum / 7 is what it would take if the bodies packed
perfectly; the measured 7-env launch is printed beside it.
    python profiles/tools/mixed_parts.py [lanes] [steps]          (GPU box, repo root)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import neorl_industrial_gym_amd as ni
from bench import MIXED7

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
P = int(sys.argv[2]) if len(sys.argv) > 2 else 250
dev = "cuda:0"


def run(counts, reps=6):
    mix = ni.MixedBatchedEnv(counts, device=dev, seed=0x5EED, autoreset=True, tally=True, fused=True)
    ring = torch.zeros(16, mix.A_max, mix.ld, dtype=torch.float32, device=dev)
    for s in range(16):
        mix.fill_actions(1000 + s, ring[s])
    rew = torch.empty(P, mix.ld, dtype=torch.float32, device=dev)
    fl = torch.empty(P, mix.ld, dtype=torch.int32, device=dev)
    mix.reset()
    for _ in range(2):
        mix.rollout(P, ring, rew, fl, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        mix.rollout(P, ring, rew, fl, None)
    e1.record()
    torch.cuda.synchronize()
    mix.close()
    del ring, rew, fl
    return e0.elapsed_time(e1) / reps


out = {}
for name, S, A, _ in MIXED7:
    out[name] = run([(name, B)])
    print("%-22s %8.3f ms per %d steps at %d lanes" % (name, out[name], P, B), flush=True)
per = (B // 7) // 256 * 256
mixed = run([(name, per if i else B - 6 * per) for i, (name, _, _, _) in enumerate(MIXED7)])
tot = sum(out.values())
print("sum / 7 = %.3f ms; the 7-env launch %.3f ms" % (tot / 7, mixed))
for k, v in out.items():
    print("   share %-22s %.1f %%" % (k, 100 * v / tot))
print(json.dumps({"lanes": B, "steps": P, "single_env_ms": out, "sum_over_7_ms": tot / 7, "mixed_ms": mixed}))
