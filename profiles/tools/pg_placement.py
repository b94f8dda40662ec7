#!/usr/bin/env python3
"""Does WHERE the outputs land in HBM move PowerGrid's full-output launch?  (Round 5: the same build ran 2.02-2.27 ms per 250 steps from
one process to the next on one box, profiles/r05/pg_ab_*.txt.)  One process, one handle, one action ring; the trajectory /
reward / flag buffers are views into one big allocation at different byte offsets, each timed over several launches, twice.
    python profiles/tools/pg_placement.py [lanes]          (GPU box, repo root)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import neorl_industrial_gym_amd as ni

B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
P, dev = 250, "cuda:0"
env = ni.make_batched("PowerGrid-v0", B, device=dev, autoreset=True, tally=True)
S, A, ld = env.state_dim, env.action_dim, env.ld
ring = torch.empty(34, A, ld, dtype=torch.float32, device=dev)
for s in range(34):
    env.fill_actions(1000 + s, ring[s])
env.reset()
traj_bytes, row_bytes = P * B * S * 4, P * ld * 4
slack = 1 << 30
big = torch.empty(traj_bytes + 2 * row_bytes + 3 * slack, dtype=torch.uint8, device=dev)
base = big.data_ptr()
print("allocation at 0x%x (mod 2 MiB = 0x%x, mod 1 GiB = 0x%x)" % (base, base % (2 << 20), base % (1 << 30)), flush=True)


def views(off_traj, off_rf):
    t = big[off_traj:off_traj + traj_bytes].view(torch.float32).view(P, B, S)
    o = traj_bytes + slack + off_rf
    r = big[o:o + row_bytes].view(torch.float32).view(P, ld)
    o2 = o + row_bytes + slack // 2
    f = big[o2:o2 + row_bytes].view(torch.int32).view(P, ld)
    return t, r, f


def time_one(t, r, f, n=6):
    for _ in range(2):
        env.rollout(P, ring, r, f, t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        env.rollout(P, ring, r, f, t)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for _ in range(40):                                # settle
    env.rollout(P, ring, *views(0, 0)[1:], views(0, 0)[0])
torch.cuda.synchronize()
out = {}
offsets = [0, 256, 1024, 4096, 16384, 65536, 1 << 20, (2 << 20) + 4096, 33 << 20, (1 << 29) + 12288]
for rep in range(2):
    for off in offsets:
        us = time_one(*views(off, 0))
        out.setdefault(off, []).append(us)
        print("trajectory at +%-10d  %.1f us per launch" % (off, us), flush=True)
print(json.dumps({str(k): v for k, v in out.items()}))
