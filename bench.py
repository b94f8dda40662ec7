#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the IndustrialEnv.step() hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by torch.distributed.run, one rank per GPU (RANK/LOCAL_RANK/WORLD_SIZE).
A "step" = one IndustrialEnv.step() of every lane of the batch.  Workload at N=1 =
BASELINE.json configs[1]: ChemicalReactor-v0, 65536 parallel envs, uniform random float32
actions from a pre-filled on-device ring, process noise and auto-reset drawn in-kernel from the
counter-based generator (synthetic data).  Every rank runs the same per-GPU batch
(weak scaling); lanes are keyed by global index; the only collective is the all-gather of the
11-double episode tally after the timed region.

Modes (same arithmetic, bit-identical results -- tests/test_gpu_parity.py):
  rollout (default)  fused rollout kernel: --plan-steps env.step per launch, state in registers;
                     EVERY step's return values are still materialised in HBM (observation
                     trajectory, reward, flag word), i.e. the information of the step API.
  graph / eager      step API: one step-kernel launch per env.step (state round-trips HBM),
                     replayed from a hipGraph / launched one by one.
The headline `value` is the selected mode; the default run also times the step API for a
shorter stretch and reports it under "step_api".

One JSON line on rank 0:  metric/value/unit/... + "roofline" + "cpu_baseline" (+ "parity").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS = {"cr": "ChemicalReactor-v0", "pg": "PowerGrid-v0", "ra": "RobotAssembly-v0"}
DIMS = {"cr": (12, 3), "pg": (32, 8), "ra": (24, 7)}
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


def alg_bytes_per_step(S, A):
    """SURVEY.md 8(d), step-API mode: state R + state/obs W + action R + reward W + flags W +
    counter R+W = 8*S + 4*A + 16 bytes per env-step (episode accumulators not counted)."""
    return 8 * S + 4 * A + 16


def alg_bytes_rollout(S, A, outputs):
    """SURVEY.md 8(d), fused-rollout mode (reported separately from the step-API figure): per
    env-step the action is read (4A) and the requested return values are written: observation
    4S, reward 4, flag word 4.  State/counters move once per LAUNCH (amortised, not counted)."""
    return 4 * A + {"full": 4 * S + 8, "min": 8, "none": 0}[outputs]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--env", default="cr", choices=list(ENVS) + ["mixed"],
                    help="mixed = all env types in one padded SoA batch (BASELINE config 4), rollout mode, min outputs")
    ap.add_argument("--batch", type=int, default=0, help="lanes per GPU (default: BASELINE config of --env)")
    ap.add_argument("--mode", default="rollout", choices=["graph", "eager", "rollout"],
                    help="graph/eager: one step kernel per env.step (step-API); rollout: fused multi-step kernel")
    ap.add_argument("--plan-steps", type=int, default=250,
                    help="env.step per launch (rollout mode) / per hipGraph replay (graph mode)")
    ap.add_argument("--ring", type=int, default=64, help="slots of the pre-filled action ring")
    ap.add_argument("--outputs", default="full", choices=["full", "min", "none"],
                    help="rollout mode: full = obs trajectory + reward + flags per step; min = reward + flags; none")
    ap.add_argument("--traj", default="aos", choices=["aos", "soa"], help="observation trajectory layout: [T,B,S] or [T,S,ld]")
    ap.add_argument("--mixed-set", default="readme", choices=["readme", "survey"],
                    help="--env mixed: the README's seven envs, or SURVEY 8(d).4's seven (the three reference envs + the two "
                         "Advanced candidates + two README-only plants)")
    ap.add_argument("--no-step-api", action="store_true", help="skip the secondary step-API measurement")
    ap.add_argument("--calibrate", action="store_true", help="also run known-size dword copies (PMC calibration)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=4.0, help="wall budget of the CPU baseline sample")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # NIG_BENCH_REHEARSE=1: every rank on cuda:0 with the gloo backend (a 1-GPU box cannot host two
    # RCCL ranks); used only to rehearse the N>1 control flow, never for a reported number.
    rehearse = os.environ.get("NIG_BENCH_REHEARSE") == "1"
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)   # RCCL over xGMI
    comm_dev = torch.device("cpu") if rehearse else device

    import neorl_industrial_gym_amd as ni

    if args.env == "mixed":
        return bench_mixed(args, ni, torch, dist, device, comm_dev, world, rank)
    key = args.env
    B = args.batch or {"cr": 65536, "pg": 262144, "ra": 262144}[key]
    seed = 0x5EED
    env = ni.make_batched(ENVS[key], B, device=device, seed=seed, env_index0=rank * B, autoreset=True, tally=True)
    S, A = env.state_dim, env.action_dim

    # ---- parity probe (rank 0): the workload's first steps against the CPU oracle, bit for bit
    parity = None
    if rank == 0 and not args.no_parity:
        from oracle import oracle as O
        L0 = ni._lib
        Tp = 256
        penv = ni.make_batched(ENVS[key], B, device=device, seed=seed, env_index0=0, autoreset=True)
        pring = torch.empty(Tp, A, penv.ld, dtype=torch.float32, device=device)
        for t in range(Tp):
            penv.fill_actions(t + 1, pring[t])         # slot k = the generator's action stream at t = k + 1
        pfl = torch.zeros(Tp, penv.ld, dtype=torch.int32, device=device)
        penv.reset()
        prw = torch.zeros(Tp, penv.ld, dtype=torch.float32, device=device)
        penv.rollout(Tp, pring, prw, pfl)
        viol = ((pfl[:, :B] >> L0.FLAG_NVIOL_SHIFT) & 3).sum()
        crit = ((pfl[:, :B] >> L0.FLAG_NCRIT_SHIFT) & 3).sum()
        nres = ((pfl[:, :B] & L0.FLAG_DID_RESET) != 0).sum()
        st, sc, tot, _ = O.rollout(key, B, Tp, seed=seed, flavor=O.MATH_POLY, nthreads=min(os.cpu_count() or 1, 32))
        same = bool(np.array_equal(penv.get_state().cpu().numpy().view(np.uint32), st.view(np.uint32)))
        parity = {"lanes": B, "steps": Tp, "violations_gpu": int(viol.item()), "violations_cpu": int(tot.violations),
                  "critical_gpu": int(crit.item()), "critical_cpu": int(tot.critical),
                  "episodes_gpu": int(nres.item()), "episodes_cpu": int(tot.episodes), "state_bits_equal": same}
        del pring, pfl, prw
        penv.close()
        del penv

    # ---- workload: action ring resident in HBM before the timed region
    R = args.ring
    ring = torch.empty(R, A, env.ld, dtype=torch.float32, device=device)
    for s in range(R):
        env.fill_actions(1000 + s, ring[s])
    env.reset()
    P = max(1, min(args.plan_steps, args.steps)) if args.mode in ("graph", "rollout") else 1
    plan = None
    launches = [0]
    traj = rew_t = fl_t = None
    if args.mode == "rollout" and args.outputs != "none":
        rew_t = torch.empty(P, env.ld, dtype=torch.float32, device=device)
        fl_t = torch.empty(P, env.ld, dtype=torch.int32, device=device)
        if args.outputs == "full":
            traj = (torch.empty(P, B, S, dtype=torch.float32, device=device) if args.traj == "aos"   # row-major [T,B,S]
                    else torch.empty(P, S, env.ld, dtype=torch.float32, device=device))

    def run_rollout(n):
        full, rem = divmod(n, P)
        for _ in range(full):
            env.rollout(P, ring, rew_t, fl_t, traj)
        if rem:
            env.rollout(rem, ring, rew_t, fl_t, traj)
        launches[0] += full + (1 if rem else 0)

    def run_step_api(n, use_plan):
        full, rem = (divmod(n, P) if use_plan is not None else (0, n))
        for _ in range(full):
            use_plan.launch()
        for k in range(rem):
            env.step_raw(ring[k % R], env.ld, reward=True, flags=True)
        launches[0] += n

    if args.mode == "graph":
        plan = env.make_plan(P, ring, env.reward, env.flags)

    def run(n):
        if args.mode == "rollout":
            run_rollout(n)
        else:
            run_step_api(n, plan)

    def barrier():
        if world > 1:
            dist.barrier()

    run(args.warmup)
    torch.cuda.synchronize()
    launches[0] = 0
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    run(args.steps)
    ev1.record()
    torch.cuda.synchronize(); barrier()
    t1 = time.perf_counter()
    wall = t1 - t0
    dev_ms = ev0.elapsed_time(ev1)
    tmax = torch.tensor([wall, dev_ms], dtype=torch.float64, device=comm_dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    wall, dev_ms = float(tmax[0].item()), float(tmax[1].item())

    # ---- secondary measurement: the step API (one kernel launch per env.step, hipGraph replay)
    step_api = None
    if args.mode == "rollout" and not args.no_step_api:
        K2 = max(P, min(args.steps, 4000) // P * P)
        plan2 = env.make_plan(P, ring, env.reward, env.flags)
        run_step_api(P, plan2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier(); torch.cuda.synchronize()
        w0 = time.perf_counter(); e0.record()
        run_step_api(K2, plan2)
        e1.record(); torch.cuda.synchronize(); barrier()
        w2 = time.perf_counter() - w0
        tm2 = torch.tensor([w2, e0.elapsed_time(e1)], dtype=torch.float64, device=comm_dev)
        if world > 1:
            dist.all_reduce(tm2, op=dist.ReduceOp.MAX)
        step_api = (K2, float(tm2[0].item()), float(tm2[1].item()))
        plan2.close()

    if args.calibrate:   # known-size dword-per-lane copies for the PMC byte calibration (profiles/)
        cal = torch.empty(S, env.ld, dtype=torch.float32, device=device)
        for _ in range(20):
            ni._lib.check(env._L.nig_get_state(env._h, cal.data_ptr(), env.ld, None, env._stream()))
        torch.cuda.synchronize()

    # ---- the path's one exchange: final tally reduction (after the timed region)
    from neorl_industrial_gym_amd.parallel import all_reduce_partial
    total = all_reduce_partial(env.reduce_tally().to(comm_dev)).cpu().numpy()
    L = ni._lib

    if rank == 0:
        steps_total = args.steps * B * world
        value = steps_total / wall
        rollout_mode = args.mode == "rollout"
        bytes_step = alg_bytes_rollout(S, A, args.outputs) if rollout_mode else alg_bytes_per_step(S, A)
        n_launch = max(launches[0], 1) if not rollout_mode else (args.steps // P + (1 if args.steps % P else 0))
        steps_per_launch = args.steps / n_launch
        launch_us = dev_ms * 1e3 / n_launch                     # HIP events over the timed region / launches
        # algorithmic bytes per launch = SURVEY 8(d) per-env-step figure x env-steps one launch processes
        achieved = bytes_step * B * steps_per_launch / (launch_us * 1e-6) / 1e9    # GB/s
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(f"{key}_{B}_{args.mode}_{args.outputs if rollout_mode else 'step'}", {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "env-steps/sec (whole node) + safety-violation-count parity, ChemicalReactor-v0"
                      if key == "cr" else f"env-steps/sec (whole node), {ENVS[key]}",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{ENVS[key]}, batch={B} parallel envs per GPU, "
                                   + (f"fused rollout kernel ({P} env.step per launch, state in registers, "
                                      f"per-step outputs: {args.outputs})"
                                      if args.mode == "rollout" else
                                      f"step-API (one fused step kernel per env.step), {args.mode} launch"
                                      + (f" ({P} steps per hipGraph replay)" if plan is not None else "")),
                       "batch_per_gpu": B, "global_batch": B * world, "action_ring": R,
                       "autoreset": True, "episode_tally": True,
                       "parallelism": f"env-shard x{world} (no data-path collective)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": ("rollout_kernel<%s>" if args.mode == "rollout" else "step_kernel<%s,false>")
                                   % {"cr": "ChemicalReactor", "pg": "PowerGrid", "ra": "RobotAssembly"}[key],
                         "alg_bytes_per_env_step": bytes_step, "env_steps_per_launch": B * steps_per_launch,
                         "alg_bytes_per_launch": bytes_step * B * steps_per_launch, "launch_us": launch_us,
                         "bytes_model": ("fused-rollout figure (SURVEY 8d): action read + requested per-step outputs"
                                         if rollout_mode else "step-API figure (SURVEY 8d): 8S+4A+16")},
            "tally": {"episodes": int(total[L.T_EPISODES]), "violations": int(total[L.T_VIOL]),
                      "critical": int(total[L.T_CRIT]), "return_mean": float(total[L.T_RET_SUM] / max(total[L.T_EPISODES], 1))},
        }
        if step_api is not None:
            K2, w2, d2 = step_api
            b2 = alg_bytes_per_step(S, A)
            out["step_api"] = {"value": K2 * B * world / w2, "unit": "env-steps/s", "steps": K2,
                               "launch_us": d2 * 1e3 / K2, "alg_bytes_per_env_step": b2,
                               "achieved_GBps": b2 * B / (d2 * 1e-3 / K2) / 1e9,
                               "frac_of_hbm_peak": b2 * B / (d2 * 1e-3 / K2) / 1e9 / HBM_PEAK_GBS,
                               "note": "one step_kernel launch per env.step, hipGraph replay"}
        if parity is not None:
            out["parity"] = parity
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle as O
            cores = max(1, min(os.cpu_count() or 1, 64))
            try:
                cores = min(cores, len(os.sched_getaffinity(0)))
            except Exception:
                pass
            # calibrate, then a bounded sample of the SAME workload (same lanes, seeds, policy, auto-reset)
            c0 = time.perf_counter()
            O.rollout(key, B, 4, seed=seed, flavor=O.MATH_LIBM, nthreads=cores)
            per_step = (time.perf_counter() - c0) / 4
            Tc = int(max(8, min(2000, args.cpu_seconds / max(per_step, 1e-6))))
            c0 = time.perf_counter()
            _, _, tot, _ = O.rollout(key, B, Tc, seed=seed, flavor=O.MATH_LIBM, nthreads=cores)
            cw = time.perf_counter() - c0
            out["cpu_baseline"] = {"value": tot.steps / cw, "unit": "env-steps/s", "cores": cores, "kind": "port",
                                   "sample": f"{B} lanes x {Tc} steps of the same workload (oracle/nig_oracle.c, "
                                             f"OpenMP over lanes, libm math), {cw:.2f} s wall"}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


MIXED7 = [("ChemicalReactor-v0", 12, 3, "reference"), ("RobotAssembly-v0", 24, 7, "reference"),
          ("HVACControl-v0", 18, 5, "build-specified"), ("WaterTreatment-v0", 15, 4, "build-specified"),
          ("SteelAnnealing-v0", 20, 6, "build-specified"), ("PowerGrid-v0", 32, 8, "reference"),
          ("SupplyChain-v0", 28, 10, "build-specified")]


MIXED7_SURVEY = [("ChemicalReactor-v0", 12, 3, "reference"), ("PowerGrid-v0", 32, 8, "reference"),
                 ("RobotAssembly-v0", 24, 7, "reference"),
                 ("AdvancedChemicalReactor-v0", 20, 6, "restated from source text (not instantiable upstream)"),
                 ("AdvancedPowerGrid-v0", 32, 8, "restated from source text (not instantiable upstream)"),
                 ("HVACControl-v0", 18, 5, "build-specified"), ("WaterTreatment-v0", 15, 4, "build-specified")]


def bench_mixed(args, ni, torch, dist, device, comm_dev, world, rank):
    """BASELINE config 4: the README's seven environments (README.md:24-32) in ONE padded SoA batch of
    --batch lanes (default 1 048 576), contiguous 256-aligned segments of equal size, one rollout
    kernel per segment on its own stream.  Three of the seven exist upstream and are parity-checked
    against the reference; four are README-only there and run build-specified plants (flagged per env).
    Reports total and per-env throughput; the HBM figure is the lane-weighted fused-rollout byte count
    with reward+flags outputs."""
    B = args.batch or 1048576
    per = (B // 7) // 256 * 256
    envset = MIXED7 if args.mixed_set == "readme" else MIXED7_SURVEY
    counts = [(name, per if i else B - 6 * per) for i, (name, _, _, _) in enumerate(envset)]
    dims = {name: (S, A) for name, S, A, _ in envset}
    origin = {name: o for name, _, _, o in envset}
    mix = ni.MixedBatchedEnv(counts, device=device, seed=0x5EED, autoreset=True, tally=True, env_index0=rank * B)
    R, P = min(args.ring, 16), max(1, min(args.plan_steps, args.steps))
    ring = torch.zeros(R, mix.A_max, mix.ld, dtype=torch.float32, device=device)
    for s in range(R):
        mix.fill_actions(1000 + s, ring[s])
    rew = torch.empty(P, mix.ld, dtype=torch.float32, device=device)
    fl = torch.empty(P, mix.ld, dtype=torch.int32, device=device)
    mix.reset()

    def run(n):
        full, rem = divmod(n, P)
        for _ in range(full):
            mix.rollout(P, ring, rew, fl)
        if rem:
            mix.rollout(rem, ring, rew, fl)

    run(args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); ev0.record()
    run(args.steps)
    ev1.record(); torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    tm = torch.tensor([wall, ev0.elapsed_time(ev1)], dtype=torch.float64, device=comm_dev)
    if world > 1:
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    wall, dev_ms = float(tm[0]), float(tm[1])
    # per-env rate measured separately on its own segment size (same kernels)
    per_env = {}
    for (name, n), seg in zip(counts, mix.envs):
        o = mix.offsets[mix.envs.index(seg)]
        reps = max(1, args.steps // P // 4)
        torch.cuda.synchronize(); c0 = time.perf_counter()
        for _ in range(reps):
            seg.rollout(P, ring[:, :seg.action_dim, o:o + n], rew[:, o:o + n], fl[:, o:o + n])
        torch.cuda.synchronize()
        per_env[name] = {"lanes": n, "state_dim": dims[name][0], "action_dim": dims[name][1],
                         "env_steps_per_s": reps * P * n / (time.perf_counter() - c0),
                         "dynamics": origin[name], "reference_parity": origin[name] == "reference"}
    if rank == 0:
        bytes_launch = sum((4 * v["action_dim"] + 8) * v["lanes"] for v in per_env.values()) * P
        n_launch = args.steps // P + (1 if args.steps % P else 0)
        launch_us = dev_ms * 1e3 / n_launch
        achieved = bytes_launch / (launch_us * 1e-6) / 1e9
        print(json.dumps({
            "metric": "env-steps/sec (whole node), all 7 envs mixed-batch", "value": args.steps * B * world / wall,
            "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"mixed padded-SoA batch of {B} lanes per GPU (S_max={mix.S_max}, A_max={mix.A_max}): "
                                   + ", ".join(f"{n} x {e}" for e, n in counts) + f"; fused rollout, {P} env.step per launch, "
                                   "reward+flags outputs", "batch_per_gpu": B, "segments": counts},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "rollout_kernel<*,1> x7 (concurrent streams)",
                         "alg_bytes_per_launch": bytes_launch, "launch_us": launch_us,
                         "bytes_model": "fused-rollout figure: action read + reward + flag word per env-step, lane-weighted"},
            "per_env": per_env}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
