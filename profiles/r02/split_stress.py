"""One-off stress of the three-wave rollout on the GPU box: very long launches, single-step launches at the full
batch, episodes of one step (a cooperative reset of all 64 lanes of every wave in every step).  Prints timings; run
under `timeout`."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ni = importlib.import_module("neorl-industrial-gym_amd")
dev = torch.device("cuda:0")
for name, B, T, ms, reps in (("long launch", 65536, 20000, None, 1), ("single-step launches", 65536, 1, None, 300),
                              ("reset every step", 65536, 500, 1, 2), ("three-step episodes", 65536, 500, 3, 2)):
    env = ni.make_batched("ChemicalReactor-v0", B, device=dev, autoreset=True, tally=True, max_episode_steps=ms)
    ring = torch.empty(8, env.action_dim, env.ld, dtype=torch.float32, device=dev)
    for s in range(8):
        env.fill_actions(5 + s, ring[s])
    env.reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        env.rollout(T, ring)
        env.counter = env.counter | 1 if T == 1 else env.counter     # single-step launches: keep them on the odd (paired) start
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name}: B={B} T={T} x{reps} max_steps={ms}: {dt*1e3:.1f} ms, {B*T*reps/dt:.3e} env-steps/s, episodes {int(env.tally[0].sum().item())}", flush=True)
    env.close()
