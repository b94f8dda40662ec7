#!/usr/bin/env python3
"""Soak of the cooperating-wave kernels under the BOUNDED-WAIT library variant (libnig_ringlimit.so, csrc/nig_ring.hpp): every
ring-protocol kernel family runs back to back for a fixed wall time at its BASELINE-sized batch; with bounded waits a protocol
slip is an error code naming the ring instead of a hang, and the episode tallies must keep growing.  VERDICT r04 weak #10: the
rings are formally racy under the LLVM memory model and production waits are unbounded by design -- this is the long run behind
the suite's short ones (tests/test_gpu_ring_limit.py).
    NIG_LIB_PATH=neorl-industrial-gym_amd/libnig_ringlimit.so python profiles/tools/ring_soak.py [seconds per family]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import neorl_industrial_gym_amd as ni

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 45.0
L = ni._lib.lib()
bounded = L.nig_tune(ni._lib.TUNE_DIAG_RING_FAULT, 0) == 0          # only the bounded-wait variant knows this key
ni.tune(split_blocks=-1, wide_min_blocks=-1)
P = 250
out = {"bounded_waits": bounded, "seconds_per_family": SECONDS, "families": {}}
cases = [("ChemicalReactor-v0", 65536, "open", "split_rollout_kernel (three-wave, open loop)"),
         ("RobotAssembly-v0", 65536, "open", "split_rollout_kernel<RobotAssembly> (three ring slots)"),
         ("PowerGrid-v0", 65536, "open", "rollout_pg_pair_kernel (producer + stepping wave)"),
         ("ChemicalReactor-v0", 65536, "policy", "split_policy_kernel (closed loop, transition stream)"),
         ("RobotAssembly-v0", 65536, "policy", "split_policy_kernel<RobotAssembly> (observation rows in the P -> I slot)"),
         ("PowerGrid-v0", 65536, "policy", "rollout_pg_pair_policy_kernel (register-resident stepper + stream)")]
for name, B, kind, what in cases:
    env = ni.make_batched(name, B, autoreset=True, tally=True)
    env.reset()
    rew = torch.empty(P, env.ld, dtype=torch.float32, device=env.device)
    fl = torch.empty(P, env.ld, dtype=torch.int32, device=env.device)
    obs = torch.empty(P, B, env.state_dim, dtype=torch.float32, device=env.device)
    if kind == "open":
        ring = torch.empty(64, env.action_dim, env.ld, dtype=torch.float32, device=env.device)
        for s in range(64):
            env.fill_actions(100 + s, ring[s])
        run = lambda: env.rollout(P, ring, rew, fl, obs)
    else:
        env.set_policy(ni.behaviour_policy(name, "medium"))
        act = torch.empty(P, env.action_dim, env.ld, dtype=torch.float32, device=env.device)
        run = lambda: env.rollout_policy(P, rew, fl, obs, act)
    t0, launches, last_eps = time.perf_counter(), 0, 0
    while time.perf_counter() - t0 < SECONDS:
        for _ in range(64):
            run()                                          # (the bounded-wait build synchronises and raises on a ring time-out)
        torch.cuda.synchronize()
        launches += 64
    eps = int(env.reduce_tally()[ni._lib.T_EPISODES].item())
    assert eps > last_eps
    out["families"][what] = {"env": name, "lanes": B, "launches": launches, "env_steps": launches * P * B, "episodes": eps}
    print("%-75s %7d launches  %.3e env-steps  %d episodes: no ring time-out" % (what, launches, launches * P * B, eps), flush=True)
    env.close()
print(json.dumps(out))
