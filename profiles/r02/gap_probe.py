"""Is the back-to-back rollout slower per launch than an isolated one?  (clock / write-drain effects)
usage: python profiles/r02/gap_probe.py   -- prints per-launch HIP-event durations, back to back and with idle gaps"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ni = importlib.import_module("neorl-industrial-gym_amd")

B, P, R = 65536, 250, 250
dev = torch.device("cuda:0")


def run(outputs, gap_s, n=60):
    env = ni.make_batched("ChemicalReactor-v0", B, device=dev, seed=1, autoreset=True, tally=True)
    ring = torch.empty(R, env.action_dim, env.ld, dtype=torch.float32, device=dev)
    for s in range(R):
        env.fill_actions(1000 + s, ring[s])
    env.reset()
    rew = fl = traj = None
    if outputs != "none":
        rew = torch.empty(P, env.ld, dtype=torch.float32, device=dev)
        fl = torch.empty(P, env.ld, dtype=torch.int32, device=dev)
        if outputs == "full":
            traj = torch.empty(P, B, env.state_dim, dtype=torch.float32, device=dev)
    for _ in range(10):
        env.rollout(P, ring, rew, fl, traj)
    torch.cuda.synchronize()
    evs = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.rollout(P, ring, rew, fl, traj); e1.record()
        evs.append((e0, e1))
        if gap_s:
            torch.cuda.synchronize(); time.sleep(gap_s)
    torch.cuda.synchronize()
    d = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    env.close()
    return d[len(d) // 2], d[0], d[-1]


for outputs in ("full", "none"):
    for gap in (0.0, 0.0005, 0.005):
        med, lo, hi = run(outputs, gap)
        print(f"split<={os.environ.get('NIG_SPLIT_BLOCKS', '256')} outputs={outputs} gap={gap*1e3:.1f} ms: per-launch us median {med:.1f} min {lo:.1f} max {hi:.1f}", flush=True)
