#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the IndustrialEnv.step() hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1: one rank per GPU (RANK/LOCAL_RANK/WORLD_SIZE from torch.distributed.run).  Started WITHOUT a launcher,
  `bench.py --gpus N` starts its own N ranks (a child `python -m torch.distributed.run ... bench.py <same args>`,
  before this process has imported torch or touched the GPU), relays rank 0's JSON line, checks that the line
  says n_gpus == N with N ranks in the tally exchange, and exits with the children's status.  A world that
  differs from --gpus is an error (non-zero exit), never a silent 1-GPU run.

One bench "step" = ONE PASS of the hot path over the batch = one fused launch that runs
--plan-steps (default 250) consecutive IndustrialEnv.step() calls of every lane (the shape of
the reference's own measurement loop, performance_benchmark.py:106-133: act -> step -> reset on
done).  K timed launches, W untimed; `ms_per_step` is per launch; `config.env_steps_per_step`
= plan_steps x batch; `value` = K x plan_steps x batch x N / wall.  Before the W warm-up launches the same
workload runs untimed for --settle seconds (default 0.6): after idle the GPU needs ~0.2 s of sustained load
to reach its steady state (profiles/r02/runlength_probe.txt), and the metric is sustained throughput.

Workload at N=1 = BASELINE.json configs[1]: ChemicalReactor-v0, 65536 parallel envs, uniform
random float32 actions from a pre-filled on-device ring (--ring slots, default 250 = one launch's steps), process noise and auto-reset drawn
in-kernel from the counter-based generator (synthetic data).  Every rank runs the same
per-GPU batch (weak scaling); lanes are keyed by global index; the only collective is the
all-gather of the episode tally after the timed region.

Modes (same arithmetic, bit-identical results -- tests/test_gpu_parity.py):
  rollout (default)  fused rollout kernel, state in registers; EVERY env.step's return values are
                     still materialised in HBM (observation trajectory, reward, flag word).
  graph / eager      step API: one step-kernel launch per env.step (state round-trips HBM),
                     --plan-steps of them replayed from one hipGraph / launched one by one.

One JSON line on rank 0: metric/value/... + "roofline" + "cpu_baseline" (+ "parity", "step_api",
"powergrid" = BASELINE configs[2] per GPU, i.e. configs[4] when N = 8, "robotassembly" = the third reference env at the
same batch, "mixed" = configs[3], "single_env" = configs[0]); every timed record carries `rank_times.clock` (the shader clock
held over the timed launches, in-kernel counters) and `rank_times.dpm` (sclk / mclk / power while its settle launches ran).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS = {"cr": "ChemicalReactor-v0", "pg": "PowerGrid-v0", "ra": "RobotAssembly-v0"}
KERNEL_ENV = {"cr": "ChemicalReactor", "pg": "PowerGrid", "ra": "RobotAssembly"}
BASELINE_BATCH = {"cr": 65536, "pg": 262144, "ra": 262144}
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
REFERENCE_PYTHON_1CORE = 21409  # reference NumPy path, ChemicalReactor-v0, 1 core, build container


def alg_bytes_per_step(S, A):
    """SURVEY.md 8(d), step-API mode: state R + state/obs W + action R + reward W + flags W +
    counter R+W = 8*S + 4*A + 16 bytes per env-step (episode accumulators not counted)."""
    return 8 * S + 4 * A + 16


def alg_bytes_rollout(S, A, outputs):
    """SURVEY.md 8(d), fused-rollout mode (reported separately from the step-API figure): per
    env-step the action is read (4A) and the requested return values are written: observation
    4S, reward 4, flag word 4.  State/counters move once per LAUNCH (amortised, not counted)."""
    return 4 * A + {"full": 4 * S + 8, "min": 8, "none": 0}[outputs]


def measured_traffic(key, B, mode, outputs, P):
    """HBM bytes per env-step from the PMC passes committed under profiles/ (FETCH_SIZE / WRITE_SIZE,
    separate passes, calibrated).  Only returned when the profiled launch had the same number of
    env.step per launch as this run (the per-launch prologue/epilogue is amortised over it)."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        rec = json.load(open(tpath)).get(f"{key}_{B}_{mode}_{outputs if mode == 'rollout' else 'step'}")
    except Exception:
        return None, None
    if not rec or int(rec.get("plan_steps", -1)) != int(P):
        return None, None
    return float(rec["hbm_bytes_per_env_step"]), rec.get("source")


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


class Workload:
    """One env type on this rank's GPU with its action ring and output buffers resident in HBM."""

    def __init__(self, ni, torch, key, B, device, rank, mode, P, ring_len, outputs, traj_layout, seed=0x5EED, ring_layout="rows"):
        self.ni, self.torch, self.key, self.B, self.mode, self.P = ni, torch, key, B, mode, P
        self.outputs = outputs
        # "aos": the action ring as [R, B, A] (row-major slots: a policy's batched output; nig_rollout's ld_act == 0) -- rollout mode only
        self.ring_aos = ring_layout == "aos" and mode == "rollout"
        env = ni.make_batched(ENVS[key], B, device=device, seed=seed, env_index0=rank * B, autoreset=True, tally=True)
        self.env = env
        self.S, self.A = env.state_dim, env.action_dim
        self.R = ring_len
        self.ring = torch.empty(ring_len, self.A, env.ld, dtype=torch.float32, device=device)
        for s in range(ring_len):
            env.fill_actions(1000 + s, self.ring[s])
        if self.ring_aos:
            self.ring = self.ring[:, :, :B].permute(0, 2, 1).contiguous()
        env.reset()
        self.rew = self.fl = self.traj = None
        if mode == "rollout" and outputs != "none":
            # NIG_DIAG_OVERWRITE=1 (diagnostic, never a headline): every step writes the SAME rows (row stride 0), so the
            # stores are issued but stay in cache -- separates "bound by issuing the stores" from "bound by draining them"
            rows = 1 if os.environ.get("NIG_DIAG_OVERWRITE") else P
            self.rew = torch.empty(rows, env.ld, dtype=torch.float32, device=device).expand(P, env.ld)
            self.fl = torch.empty(rows, env.ld, dtype=torch.int32, device=device).expand(P, env.ld)
            if outputs == "full":
                self.traj = (torch.empty(rows, B, self.S, dtype=torch.float32, device=device).expand(P, B, self.S) if traj_layout == "aos"   # row-major [T,B,S]
                             else torch.empty(rows, self.S, env.ld, dtype=torch.float32, device=device).expand(P, self.S, env.ld))
        self.plan = env.make_plan(P, self.ring, env.reward, env.flags) if mode == "graph" else None
        self.rings, self._turn = [self.ring], 0

    def use_big_ring(self, min_bytes):
        """Action rings whose union exceeds `min_bytes`: a launch reads slots 0..P-1 of ITS ring only (step k reads slot
        k mod R), so consecutive launches cycle through several P-slot rings -- the bytes re-read between two uses of
        a slot exceed the 256 MB Infinity Cache (the default 250-slot ring, 196 MB for the headline, sits inside it)."""
        slot_bytes = self.A * self.env.ld * 4
        n = max(2, -(-int(min_bytes) // (self.P * slot_bytes)))
        big = self.torch.empty(n, self.P, self.A, self.env.ld, dtype=self.torch.float32, device=self.ring.device)
        for j in range(n):
            for s_ in range(self.P):
                self.env.fill_actions(5000 + j * self.P + s_, big[j, s_])
        self.rings = [(big[j][:, :, :self.B].permute(0, 2, 1).contiguous() if self.ring_aos else big[j]) for j in range(n)]
        return n * self.P * slot_bytes

    def launch(self):
        """One bench step: P env.step of every lane."""
        if self.mode == "rollout":
            ring = self.rings[self._turn]
            self._turn = (self._turn + 1) % len(self.rings)
            self.env.rollout(self.P, ring, self.rew, self.fl, self.traj)
        elif self.mode == "graph":
            self.plan.launch()
        else:
            for k in range(self.P):
                self.env.step_raw(self.ring[k % self.R], self.env.ld, reward=True, flags=True)

    def kernels_per_launch(self):
        return 1 if self.mode == "rollout" else self.P

    def close(self):
        if self.plan is not None:
            self.plan.close()
        self.env.close()


class ClockProbe:
    """The clock the chip HELD over a run of launches (box-to-box spread of 164-176 us on an unchanged
    kernel could not be attributed): two nig_clock_stamp launches around the run, each writing s_memtime (shader-clock counter)
    and s_memrealtime (constant 100 MHz) per XCD; shader clock = d memtime / d memrealtime x 100 MHz.  The stamps are tiny
    kernels OUTSIDE the timed interval (before its opening barrier, after its clock has stopped)."""

    def __init__(self, ni, torch, device):
        self.L, self.torch = ni._lib.lib(), torch
        self.buf = torch.zeros(2, 4096, dtype=torch.int64, device=device)        # NIG_CLOCK_STAMP_WORDS per stamp

    def stamp(self, i):
        if i == 0:
            self.buf.zero_()
        self.L.nig_clock_stamp(self.torch.cuda.current_stream().cuda_stream, self.buf[i].data_ptr())

    def read(self):
        import numpy as np
        v = self.buf.cpu().numpy().astype(np.uint64).reshape(2, 2048, 2)
        both = (v[0, :, 0] != 0) & (v[1, :, 0] != 0) & (v[1, :, 1] > v[0, :, 1]) & (v[1, :, 0] > v[0, :, 0])
        if not both.any():
            return {"error": "no compute unit stamped twice"}
        dt = (v[1, both, 0] - v[0, both, 0]).astype(np.float64)
        dr = (v[1, both, 1] - v[0, both, 1]).astype(np.float64)
        mhz = np.sort(dt / dr * 100.0)
        return {"shader_clock_mhz": float(np.median(mhz)), "shader_clock_mhz_p05": float(mhz[int(0.05 * (mhz.size - 1))]),
                "shader_clock_mhz_p95": float(mhz[int(0.95 * (mhz.size - 1))]), "compute_units": int(mhz.size),
                "span_ms": float(np.median(dr) / 1e5),
                "how": "(s_memtime1 - s_memtime0) / (s_memrealtime1 - s_memrealtime0) x 100 MHz per compute unit stamped both times, "
                       "median; the stamps are outside the timed interval"}


_PROBE = None            # set by main() on a GPU run
_DPM_DIR = None


def gpu_sysfs_dir(torch, idx):
    """sysfs directory of HIP device `idx` (its PCI function), for the DPM state files rocm-smi reads."""
    import glob
    try:
        p = torch.cuda.get_device_properties(idx)
        d = "/sys/bus/pci/devices/%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        if os.path.exists(d + "/pp_dpm_sclk"):
            return d
    except Exception:
        pass
    try:
        amd = [c for c in sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
               if open(c + "/vendor").read().strip() == "0x1002" and os.path.exists(c + "/pp_dpm_sclk")]
        return amd[idx] if idx < len(amd) else None
    except Exception:
        return None


def dpm_sample(d):
    """What `rocm-smi --showclocks --showpower` would print at this moment, read from the files it reads (no child process
    while launches are in flight): the starred level of pp_dpm_sclk / pp_dpm_mclk, hwmon power, gpu_busy_percent."""
    import glob
    if not d:
        return {"error": "no sysfs directory for this device"}
    out = {}

    def starred(name):
        try:
            for line in open(f"{d}/{name}"):
                if "*" in line:
                    return int("".join(ch for ch in line.split(":")[1] if ch.isdigit()))
        except Exception:
            return None
        return None
    out["sclk_mhz"], out["mclk_mhz"] = starred("pp_dpm_sclk"), starred("pp_dpm_mclk")
    for name in ("power1_average", "power1_input"):
        f = glob.glob(f"{d}/hwmon/hwmon*/{name}")
        if f:
            try:
                out["power_w"] = int(open(f[0]).read()) / 1e6
                break
            except Exception:
                pass
    try:
        out["gpu_busy_percent"] = int(open(f"{d}/gpu_busy_percent").read())
    except Exception:
        pass
    out["how"] = "sysfs pp_dpm_sclk / pp_dpm_mclk (starred level), hwmon power, read while the settle phase's launches were in flight"
    return out


def grouped(dist, world=1):
    """True when this process is a member of a torch.distributed group: every N > 1 run, and the one-rank "nccl" group of
    NIG_BENCH_FORCE_PG=1 (the RCCL code path of this file -- init_process_group("nccl"), barrier,
    all_gather of device tensors, all_gather_object, destroy_process_group -- executed on the one GPU a build box has)."""
    if world > 1:
        return True
    try:
        return bool(dist is not None and dist.is_available() and dist.is_initialized())
    except AttributeError:               # (a test double without the query functions)
        return False


def settle(torch, wl, seconds, stats=None):
    """Untimed: run the workload back to back for `seconds` before the W warm-up launches.  After an idle phase
    (process start, buffer setup) this GPU takes ~0.1-0.2 s of sustained load to reach its steady state: the first
    20 launches of the headline rollout take 235 us each, launches 1000+ take 175 us (profiles/r02/runlength_probe.txt).
    The metric is sustained throughput, so the ramp is kept out of the timed region.  Returns the launches run."""
    n = 0
    if seconds > 0:
        # at most ~2 000 kernel dispatches between two synchronisations: a hipGraph replay is 251 of them, and 32 replays
        # in flight (8 032 dispatches) is what the one hung `rocprofv3 --pmc` pass of round 2 had queued behind the
        # profiler's per-dispatch counter packets (profiles/README.md); 10 replays (2 510) had always completed
        per = max(1, min(32, 2000 // max(1, wl.kernels_per_launch() + 1)))
        t_end = time.perf_counter() + seconds
        while time.perf_counter() < t_end:
            for _ in range(per):
                wl.launch()
            if stats is not None and _DPM_DIR:      # launches in flight: the DPM state of THIS load (the last sample is kept)
                stats["dpm"] = dpm_sample(_DPM_DIR)
            torch.cuda.synchronize()
            n += per
    return n


def timed(torch, dist, world, comm_dev, wl, K, W, settle_s=0.0, stats=None):
    """[settle_s of untimed load,] W untimed + exactly K timed launches, bracketed by barrier + synchronize on both
    sides; HIP events on the launch stream around the same region.  Returns (wall s, device ms): max over ranks.

    The timed interval is THE SAME SET OF OPERATIONS at every world size: K launches + the stream synchronisation
    behind them.  The clock stops after the synchronisation and BEFORE the closing barrier (the barrier still brackets
    the region -- nobody leaves it while a rank is still running -- but a collective inside the interval would be a
    cost the N = 1 run never pays: with the driver's --steps 20 the region is a few ms and one RCCL barrier 0.05-0.2 ms,
    VERDICT r03 weak #5).  Stragglers are covered by the MAX over ranks below.  `stats` (dict, optional) receives every
    rank's own wall / device time: wall_min / wall_median / wall_max (s), launch_us per rank (HIP events)."""
    gpu = torch.cuda.is_available()          # False only in the CPU control-flow rehearsal (NIG_BENCH_REHEARSE=cpu)
    sync = torch.cuda.synchronize if gpu else (lambda: None)
    n_settle = settle(torch, wl, settle_s, stats) if gpu else 0
    for _ in range(W):
        wl.launch()
    sync()
    if stats is not None:      # launches of this workload so far, in order: profiles/phase_stats.py splits a kernel trace by them
        stats["phases"] = [{"name": "settle", "launches": n_settle}, {"name": "warmup", "launches": W}, {"name": "timed", "launches": K}]
    ev0 = ev1 = None
    probe = _PROBE if (gpu and stats is not None) else None
    if gpu:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if probe:
        probe.stamp(0)
    if grouped(dist, world):
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    if gpu:
        ev0.record()
    for _ in range(K):
        wl.launch()
    if gpu:
        ev1.record()
    sync()
    wall = time.perf_counter() - t0          # <- the timed interval ends here, at every world size
    if probe:
        probe.stamp(1)
        sync()
        stats["clock"] = probe.read()
    if grouped(dist, world):
        dist.barrier()
    mine = torch.tensor([wall, ev0.elapsed_time(ev1) if gpu else wall * 1e3], dtype=torch.float64, device=comm_dev)
    if grouped(dist, world):
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        every = torch.stack(every).cpu()
    else:
        every = mine.cpu().reshape(1, 2)
    walls = sorted(float(x) for x in every[:, 0].tolist())
    if stats is not None:
        n_k = K * wl.kernels_per_launch()
        stats.update({"ranks": world, "wall_min_s": walls[0], "wall_median_s": walls[len(walls) // 2], "wall_max_s": walls[-1],
                      "wall_s_per_rank": [float(x) for x in every[:, 0].tolist()],
                      "launch_us_per_rank": [float(x) * 1e3 / n_k for x in every[:, 1].tolist()],
                      "timed_interval": "K launches + stream synchronize; barrier before and after, both outside the interval"})
    return walls[-1], float(every[:, 1].max().item())


def timed_events(torch, wl, K):
    """K launches between two HIP events on the launch stream; no barrier, no warm-up.  Returns us per launch."""
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(K):
        wl.launch()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) * 1e3 / K


def honest_brackets(torch, wl, K, roof, settle_s, idle_gap=1.0, phases=None):
    """What the sustained headline figure does not show, measured on the same workload object:
    * write_only_frac   -- the output bytes alone (4S+8 per env-step; action reads can be served by the Infinity Cache,
                           the trajectory stores cannot) / launch time / peak: the floor of the HBM-side fraction;
    * cold_first_launches -- the same K launches re-timed right after `idle_gap` seconds of idle, no settle, no warm-up;
    * ring_gt_mall      -- sustained, with action rings of > 2 x 256 MB cycled launch by launch, so every action
                           read comes from HBM."""
    env_steps = wl.B * wl.P
    out = {"write_only_frac": (4 * wl.S + 8) * env_steps / (roof["launch_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
           if wl.outputs == "full" else None}       # (== roofline.frac when the action ring fits the Infinity Cache)
    torch.cuda.synchronize()
    time.sleep(idle_gap)
    us = timed_events(torch, wl, K)
    frac_of = lambda bytes_step, us_: bytes_step * env_steps / (us_ * 1e-6) / 1e9 / HBM_PEAK_GBS
    out["cold_first_launches"] = {"launches": K, "idle_gap_s": idle_gap, "launch_us": us,
                                  "frac": frac_of(roof["hbm_side_bytes_per_env_step"], us),
                                  "frac_algorithmic": frac_of(roof["alg_bytes_per_env_step"], us)}
    ring_bytes = wl.use_big_ring(2 * 256 * 2**20 + 1)
    n_settle = settle(torch, wl, settle_s)
    us = timed_events(torch, wl, K)
    if phases is not None:
        phases += [{"name": "cold_first_launches", "launches": K}, {"name": "ring_gt_mall_settle", "launches": n_settle},
                   {"name": "ring_gt_mall", "launches": K}]
    out["ring_gt_mall"] = {"ring_bytes": ring_bytes, "rings": len(wl.rings), "launch_us": us,
                           "frac": frac_of(hbm_side_bytes(wl)[0], us),       # every byte from HBM: == the algorithmic figure
                           "frac_algorithmic": frac_of(roof["alg_bytes_per_env_step"], us)}
    out["stream_probe"] = stream_probe(torch, wl.env.device)
    return out


def stream_probe(torch, device, nbytes=2 * 2**30, reps=8):
    """Context for `frac` (peak = the guide's 8 TB/s): what plain streaming kernels reach on THIS box at THIS moment over buffers
    eight times the Infinity Cache -- torch's fill (write only: the shape of the headline's trajectory stores) and a
    device-to-device copy (read + write, both directions counted).  Library kernels on torch's stream, torch events."""
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
    b = torch.empty_like(a)
    out = {"bytes": nbytes, "reps": reps}
    for name, op, moved in (("fill_GBps", lambda: a.fill_(1.0), nbytes), ("copy_GBps", lambda: b.copy_(a), 2 * nbytes)):
        op(); op()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            op()
        e1.record()
        torch.cuda.synchronize()
        out[name] = moved * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del a, b
    return out


def rollout_kernel_name(wl):
    """The kernel nig_rollout launches for this workload (csrc/nig_kernels.hpp launch_rollout_form): ChemicalReactor and
    RobotAssembly batches of whole 256-lane blocks, up to nig_tune(NIG_TUNE_SPLIT_BLOCKS) of them, run in the three-wave
    form (ChemicalReactor also larger batches, in rounds of that many blocks)."""
    out = {"none": 0, "min": 1, "full": 3}[wl.outputs]
    blocks, per_round = wl.B // 256, wl.ni.tune()["split_blocks"]
    last = blocks % per_round if per_round else 0
    # (+ a one-wave launch for a ragged last block; beyond one round only for launches that write the observation trajectory, round 5)
    if wl.key == "cr" and blocks > 0 and per_round and (blocks <= per_round or (out >= 2 and (last == 0 or 4 * last >= 3 * per_round))):
        return "split_rollout_kernel<ChemicalReactor,%d,4>" % out
    if wl.key == "ra" and blocks > 0 and per_round and blocks <= per_round:     # RobotAssembly: the three-wave form for one round only
        return "split_rollout_kernel<RobotAssembly,%d,4>" % out
    # PowerGrid: the LDS-resident body (csrc/nig_pg_lds.hpp) -- whole 512-lane blocks of a batch of at least
    # nig_tune(NIG_TUNE_WIDE_MIN_BLOCKS) of them in the wide form, whole 256-lane blocks otherwise; a ragged tail (and
    # everything, with the knob at 2^30 or more) on the register-resident kernel
    wide_min = wl.ni.tune()["wide_min_blocks"]
    if wl.key == "pg" and wide_min < (1 << 30) and wl.B >= 256:
        if wl.B // 512 >= max(1, wide_min):
            return "rollout_wide_kernel<PowerGrid,%d,512>" % out
        if per_round and blocks <= per_round:     # at most one 256-lane block per compute unit: stepping + producer wave per 64 lanes
            return "rollout_pg_pair_kernel<%d>" % out
        return "rollout_wide_kernel<PowerGrid,%d,256>" % out
    return "rollout_kernel<%s,%d>" % (KERNEL_ENV[wl.key], out)


def policy_kernel_name(ni, key, B, policy_kind="affine", stream_obs=True):
    """The kernel nig_rollout_policy launches for the whole 256-lane blocks of an auto-reset handle without frozen lanes
    (csrc/nig_kernels.hpp launch_policy) -- since round 5 WHATEVER transition-stream outputs the call asks for: ChemicalReactor
    in the three-wave closed-loop form (a second round only with the observation stream, never more), RobotAssembly in its BIG layout for a single round (the observation
    rows ride in the producer -> integrator slot), PowerGrid's affine policies in the paired form with the register-resident
    stepper (which writes the observation stream through the reset image); everything else on rollout_policy_kernel."""
    blocks, per_round = B // 256, ni.tune()["split_blocks"]
    last = blocks % per_round if per_round else 0
    if key == "cr" and blocks > 0 and per_round and (blocks <= per_round or (stream_obs and blocks <= 2 * per_round and (last == 0 or 4 * last >= 3 * per_round))):
        return "split_policy_kernel<ChemicalReactor,4>"
    if key == "ra" and blocks > 0 and per_round and blocks <= per_round:
        return "split_policy_kernel<RobotAssembly,4>"
    if key == "pg" and policy_kind == "affine" and blocks > 0 and per_round and blocks <= per_round:
        return "rollout_pg_pair_policy_kernel<PolicyArgs> (pg_policy_reg_body)"
    return "rollout_policy_kernel<%s>" % KERNEL_ENV[key]


MALL_BYTES = 256 * 2**20      # Infinity Cache (MI355X_MICROARCH.md): FETCH_SIZE counts its hits, HBM does not serve them


def hbm_side_bytes(wl):
    """Bytes per env-step that cannot be served by the 256 MB Infinity Cache, and the reasoning:
    a byte stream counts when its footprint between two uses exceeds the cache.
    rollout mode -- per-step outputs: written once per launch, P x B x (4S + 8) bytes, never re-read (917 MB for the
    headline: HBM); action reads: slot k mod R of the ring(s), re-read every launch -- HBM only if the rings' union
    exceeds the cache (the default 250-slot ring is 196 MB: Infinity-Cache hits, measured: same launch time as a 50 MB ring).
    step API -- state, counters, actions, reward, flags are all re-used every launch: they count when their union (the
    launch's working set) exceeds the cache, and not at all when it fits (65 536 lanes: 4 MB -- that launch is
    latency-bound and moves nothing to HBM in steady state)."""
    S, A, ld = wl.S, wl.A, wl.env.ld
    if wl.mode == "rollout":
        out_b = {"full": 4 * S + 8, "min": 8, "none": 0}[wl.outputs]
        out_foot = out_b * wl.P * wl.B
        ring_foot = sum(int(r.numel()) * 4 for r in wl.rings)
        b = (out_b if out_foot > MALL_BYTES else 0) + (4 * A if ring_foot > MALL_BYTES else 0)
        return b, {"outputs_footprint_bytes": out_foot, "action_ring_footprint_bytes": ring_foot, "cache_bytes": MALL_BYTES}
    work = (4 * S + 4 + 8) * ld + wl.R * A * ld * 4 + 8 * ld + (8 + 8 * 13) * ld     # state, counters, ring, reward + flags, tally rows
    return (alg_bytes_per_step(S, A) if work > MALL_BYTES else 0), {"working_set_bytes": work, "cache_bytes": MALL_BYTES}


def roofline_of(wl, K, dev_ms):
    """`frac` is the HBM-SIDE fraction (hbm_side_bytes: what the Infinity Cache cannot serve) -- the figure that cannot
    flatter; `frac_algorithmic` / `achieved` keep SURVEY 8(d)'s algorithmic bytes (every action read counted, wherever it
    is served from), the number earlier rounds reported as `frac`."""
    rollout = wl.mode == "rollout"
    bytes_step = alg_bytes_rollout(wl.S, wl.A, wl.outputs) if rollout else alg_bytes_per_step(wl.S, wl.A)
    n_kernels = K * wl.kernels_per_launch()
    kernel_us = dev_ms * 1e3 / n_kernels                        # HIP events over the timed region / kernel launches in it
    env_steps_per_kernel = wl.B * (wl.P if rollout else 1)
    alg = bytes_step * env_steps_per_kernel
    achieved = alg / (kernel_us * 1e-6) / 1e9
    hbm_b, why = hbm_side_bytes(wl)
    hbm_achieved = hbm_b * env_steps_per_kernel / (kernel_us * 1e-6) / 1e9
    per_step, src = measured_traffic(wl.key, wl.B, wl.mode, wl.outputs, wl.P)
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_achieved / HBM_PEAK_GBS,
            "frac_algorithmic": achieved / HBM_PEAK_GBS, "hbm_side_GBps": hbm_achieved, "hbm_side_bytes_per_env_step": hbm_b,
            "hbm_side_model": why,
            "traffic": None if per_step is None else per_step * env_steps_per_kernel,
            "traffic_bytes_per_env_step": per_step, "traffic_source": src,
            "traffic_measured_in_this_run": False,      # looked up from the committed PMC passes of the same kernel and launch shape (profiles/traffic.json)
            "kernel": (rollout_kernel_name(wl) if rollout else "step_kernel<%s,false>" % KERNEL_ENV[wl.key]),
            "alg_bytes_per_env_step": bytes_step, "env_steps_per_launch": env_steps_per_kernel,
            "alg_bytes_per_launch": alg, "launch_us": kernel_us, "launches_timed": n_kernels,
            "bytes_model": ("fused-rollout figure (SURVEY 8d): action read + requested per-step outputs"
                            if rollout else "step-API figure (SURVEY 8d): 8S+4A+16")
                           + "; achieved / frac_algorithmic use it, frac counts only the bytes the Infinity Cache cannot serve"}


def gathered_tally(torch, dist, world, comm_dev, wl):
    """The path's one exchange: all-gather of every rank's partial tally + fixed-order combine.
    Self-check: every rank contributed and the combined counts are the sums of the per-rank ones."""
    return gathered_partial(world, comm_dev, wl.ni, wl.env.reduce_tally())


def gathered_partial(world, comm_dev, ni, partial):
    from neorl_industrial_gym_amd.parallel import all_gather_partials, combine_partials
    L = ni._lib
    parts = all_gather_partials(partial.to(comm_dev)).cpu()
    total = combine_partials(parts).numpy()
    per_rank_eps = [int(x) for x in parts[:, L.T_EPISODES].tolist()]
    ok = (parts.shape[0] == world and all(e > 0 for e in per_rank_eps)
          and int(total[L.T_EPISODES]) == sum(per_rank_eps)
          and int(total[L.T_VIOL]) == sum(int(x) for x in parts[:, L.T_VIOL].tolist()))
    if not ok and not os.environ.get("NIG_DIAG_NO_TALLY_CHECK"):     # (the switch: diagnostic library variants whose results are garbage on purpose)
        raise SystemExit(f"tally self-check failed: ranks={parts.shape[0]}/{world} episodes per rank={per_rank_eps} "
                         f"combined={int(total[L.T_EPISODES])}")
    return total, {"ranks": int(parts.shape[0]), "episodes_per_rank": per_rank_eps, "ok": True}


def parity_probe_gpu(ni, torch, key, B, device, seed=0x5EED, Tp=256):
    """The workload's first Tp steps on the device; compared with the CPU oracle, bit for bit, by parity_probe_cpu."""
    L0 = ni._lib
    penv = ni.make_batched(ENVS[key], B, device=device, seed=seed, env_index0=0, autoreset=True)
    A = penv.action_dim
    pring = torch.empty(Tp, A, penv.ld, dtype=torch.float32, device=device)
    for t in range(Tp):
        penv.fill_actions(t + 1, pring[t])         # slot k = the generator's action stream at t = k + 1
    pfl = torch.zeros(Tp, penv.ld, dtype=torch.int32, device=device)
    prw = torch.zeros(Tp, penv.ld, dtype=torch.float32, device=device)
    penv.reset()
    penv.rollout(Tp, pring, prw, pfl)
    out = {"seed": seed, "steps": Tp,
           "violations": int(((pfl[:, :B] >> L0.FLAG_NVIOL_SHIFT) & 3).sum().item()),
           "critical": int(((pfl[:, :B] >> L0.FLAG_NCRIT_SHIFT) & 3).sum().item()),
           "episodes": int(((pfl[:, :B] & L0.FLAG_DID_RESET) != 0).sum().item()),
           "state": penv.get_state().cpu().numpy()}
    penv.close()
    return out


def parity_probe_cpu(g, key, B):
    import numpy as np
    from oracle import oracle as O
    st, sc, tot, _ = O.rollout(key, B, g["steps"], seed=g["seed"], flavor=O.MATH_POLY, nthreads=min(os.cpu_count() or 1, 32))
    same = bool(np.array_equal(g["state"].view(np.uint32), st.view(np.uint32)))
    return {"lanes": B, "steps": g["steps"], "violations_gpu": g["violations"], "violations_cpu": int(tot.violations),
            "critical_gpu": g["critical"], "critical_cpu": int(tot.critical),
            "episodes_gpu": g["episodes"], "episodes_cpu": int(tot.episodes), "state_bits_equal": same}


def fast_mode_statistics(ni, key, total, B=0, episodes_per_lane=0, device=None):
    """The metric's second half -- "safety-violation-count parity" -- in the mode this line times: the
    workload's episode statistics beside the REFERENCE's for the same loop (tests/golden/reference_stats.npz:
    performance_benchmark.py:106-133 under uniform float32 actions, run by the reference with its own np.random draws; a
    committed fixture written by `oracle/gen_golden.py stats`).  Two build-side samples:
      sample -- ni.uniform_action_statistics: the same fused kernel, generator and auto-reset, a FRESH action every step, the
                first `episodes_per_lane` episodes of every lane (unbiased); run here, untimed, after the timed regions.  This
                is the sample tests/test_gpu_reference_stats.py asserts within 4 standard errors (with >= 1e6 episodes).
      timed_workload -- the device tally of the timed run itself: every episode that FINISHED inside it, under the
                pre-filled action ring of SURVEY 8(d): a lane's actions repeat every --ring steps, which is not the reference's
                loop -- with 64 slots ChemicalReactor's violations per episode came out 3.3 % low (-11 sigma), with the default 250
                (one launch's steps; still Infinity-Cache resident, same launch time) 0.7 % (-2.5 sigma); RobotAssembly, whose
                episodes run up to 1 000 steps on a random walk of the joints, cycles four 250-slot rings.  Reported for completeness.
    deviation_sigma = (build - reference) / the reference's standard error."""
    import numpy as np
    L = ni._lib
    try:
        d = np.load(os.path.join(ROOT, "tests", "golden", "reference_stats.npz"))
        ref = {k: d[f"{key}_{k}"].astype(np.float64) for k in ("length", "viol", "crit", "ret")}
    except Exception as e:
        return {"error": repr(e)}
    n_ref = int(ref["length"].size)
    name = {"length": "episode_length_mean", "viol": "violations_per_episode", "crit": "critical_violations_per_episode", "ret": "return_mean"}

    def table(got, n):
        out = {"episodes": int(n)}
        for k, nm in name.items():
            se = float(ref[k].std() / np.sqrt(n_ref)) or float(np.sqrt(3.0 / n_ref))
            out[nm] = got[k]
            out[nm + "_deviation_sigma"] = (got[k] - float(ref[k].mean())) / se
        return out
    out = {"reference": dict({"episodes": n_ref, "source": "tests/golden/reference_stats.npz (the reference's own code and draws)"},
                             **{nm: float(ref[k].mean()) for k, nm in name.items()},
                             **{nm + "_standard_error": float(ref[k].std() / np.sqrt(n_ref)) for k, nm in name.items()})}
    n = float(total[L.T_EPISODES])
    if n > 0:
        out["timed_workload"] = table({"length": float(total[L.T_LEN_SUM]) / n, "viol": float(total[L.T_VIOL]) / n,
                                       "crit": float(total[L.T_CRIT]) / n, "ret": float(total[L.T_RET_SUM]) / n}, n)
    if episodes_per_lane > 0:
        s = ni.uniform_action_statistics(ENVS[key], B, episodes_per_lane, device=device, outputs="full")
        m = s["episodes"]
        out["sample"] = dict(table({"length": s["steps"] / m, "viol": s["viol"] / m, "crit": s["crit"] / m, "ret": s["ret"] / m}, m),
                             lanes=B, episodes_per_lane=episodes_per_lane, launches=s["launches"])
        out["violations_per_episode_gpu"], out["violations_per_episode_reference"] = s["viol"] / m, float(ref["viol"].mean())
        out["within_4_sigma"] = all(abs(v) <= 4.0 for k_, v in out["sample"].items() if k_.endswith("_deviation_sigma"))
    return out


def cpu_baseline(key, B, seconds, seed=0x5EED):
    """oracle/nig_oracle.c (the parity-checked C restatement, libm math) timed on this host: all cores
    and one thread, each on a bounded sample of the SAME workload (same lanes, seeds, policy, auto-reset)."""
    from oracle import oracle as O
    cores = max(1, min(os.cpu_count() or 1, 64))
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except Exception:
        pass

    def sample(nthreads, lanes, budget):
        c0 = time.perf_counter()
        O.rollout(key, lanes, 4, seed=seed, flavor=O.MATH_LIBM, nthreads=nthreads)
        per_step = (time.perf_counter() - c0) / 4
        Tc = int(max(8, min(2000, budget / max(per_step, 1e-6))))
        c0 = time.perf_counter()
        _, _, tot, _ = O.rollout(key, lanes, Tc, seed=seed, flavor=O.MATH_LIBM, nthreads=nthreads)
        cw = time.perf_counter() - c0
        return tot.steps / cw, Tc, cw

    v_all, T_all, w_all = sample(cores, B, seconds)
    lanes1 = min(B, 4096)
    v_one, T_one, w_one = sample(1, lanes1, min(seconds, 3.0))
    return {"value": v_all, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{B} lanes x {T_all} steps of the same workload (oracle/nig_oracle.c, OpenMP over lanes, "
                      f"libm math), {w_all:.2f} s wall",
            "threads_1": {"value": v_one, "unit": "env-steps/s",
                          "sample": f"{lanes1} lanes x {T_one} steps, 1 thread, {w_one:.2f} s wall"},
            "cpu_model": cpu_model(),
            "reference_python_1core": {"value": REFERENCE_PYTHON_1CORE, "unit": "env-steps/s",
                                       "note": "the reference's own NumPy step loop, ChemicalReactor-v0, 1 core, measured "
                                               "in the build container (SURVEY.md section 6); the reference cannot travel "
                                               "to the GPU box, so this is a stated constant, not a live measurement"}}


def gpu_numa_nodes():
    """NUMA node of every GPU, in KFD enumeration order (== HIP device order when no *_VISIBLE_DEVICES variable remaps
    it), read from sysfs: no HIP call, so it can run before anything touches the GPU.  None when it cannot be told."""
    if any(os.environ.get(v) for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")):
        return None
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        nodes = []
        for n in sorted(os.listdir(base), key=int):
            props = dict(l.split()[:2] for l in open(f"{base}/{n}/properties") if len(l.split()) >= 2)
            if int(props.get("simd_count", "0")) <= 0:
                continue                                   # a CPU node
            dom, loc = int(props.get("domain", "0")), int(props["location_id"])
            bdf = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7}"
            nodes.append(int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read()))
        return nodes or None
    except Exception:
        return None


def _cpulist(text):
    out = []
    for part in text.strip().split(","):
        if part:
            a, _, b = part.partition("-")
            out.extend(range(int(a), int(b or a) + 1))
    return out


def numa_cpulists(allowed):
    """The allowed CPUs of every NUMA node of this host, in node order (sysfs); nodes without an allowed CPU dropped."""
    nodes = sorted(int(d[4:]) for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit())
    per = [[c for c in _cpulist(open(f"/sys/devices/system/node/node{n}/cpulist").read()) if c in allowed] for n in nodes]
    return [c for c in per if c]


def pin_rank(local_rank, local_world):
    """One rank per GPU: keep this rank's host threads (launch loop, RCCL proxy) on the CPUs of ITS GPU's NUMA node,
    shared evenly with the other ranks of that node; without topology information, an even contiguous share of the
    CPUs the process may use (sockets are numbered contiguously, GPUs 0..N/2-1 hang off socket 0 on the usual
    two-socket node).  Called before torch is imported.  Never fatal: returns what it did for the result line."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
        if local_world <= 1 or len(allowed) < 2 * local_world:
            return {"pinned": False, "reason": "single rank or too few CPUs", "cpus": len(allowed)}
        numa = gpu_numa_nodes()
        how, mine = "even-split", None
        if numa and len(numa) >= local_world and numa[local_rank] >= 0:
            node = numa[local_rank]
            cpus = [c for c in _cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read()) if c in set(allowed)]
            peers = [r for r in range(local_world) if numa[r] == node]
            if len(cpus) >= len(peers):
                share = len(cpus) // len(peers)
                i = peers.index(local_rank)
                mine, how = cpus[i * share:(i + 1) * share], f"numa node {node}"
        if not mine:
            # no GPU -> NUMA map: ranks in order over the NUMA nodes in order (GPUs 0..N/2-1 hang off the first socket on the
            # usual two-socket node), each rank an even share of ITS node's CPUs -- a node's cpulist holds its cores and their
            # SMT siblings, which a plain split of the CPU numbers would hand to different sockets' ranks
            try:
                per = numa_cpulists(set(allowed))
                if len(per) > 1 and local_world % len(per) == 0:
                    rpn = local_world // len(per)                   # ranks per node
                    cpus = per[local_rank // rpn]
                    share = len(cpus) // rpn
                    if share >= 1:
                        i = local_rank % rpn
                        mine, how = cpus[i * share:(i + 1) * share], f"numa node {local_rank // rpn} of {len(per)} (ranks in node order)"
            except Exception:
                mine = None
        if not mine:
            share = len(allowed) // local_world
            mine = allowed[local_rank * share:(local_rank + 1) * share]
        os.sched_setaffinity(0, mine)
        return {"pinned": True, "how": how, "cpus": len(mine), "first_cpu": mine[0], "last_cpu": mine[-1]}
    except Exception as e:                                   # affinity is an optimisation, never a reason to fail
        return {"pinned": False, "reason": repr(e)}


def scale_record(world, rehearse, affinity, dist=None):
    """What the N-rank figure is and is not: said in the line itself, not only in README."""
    rec = {"world": world, "measured_on_hardware": bool(world > 1 and not rehearse), "rank_affinity": affinity,
           "process_group": ({"backend": dist.get_backend(), "world": dist.get_world_size(),
                              "forced_one_rank": os.environ.get("NIG_BENCH_FORCE_PG") == "1" and world == 1}
                             if grouped(dist) else None),
           "exchange": "all-gather of 13 float64 partial sums per rank after the timed region + fixed-order combine "
                       "(no data-path collective)"}
    if world == 1 and rec["process_group"]:
        rec["note"] = ("single-GPU line through a ONE-RANK process group (NIG_BENCH_FORCE_PG=1): the barrier, the all-gathers of the rank "
                       "times and of the tally partials and the object gather ran through the group's backend -- the code path of an N > 1 "
                       "run, executed; no scaling information")
    elif world == 1:
        rec["note"] = ("single-GPU line: no scaling information.  No 8-GPU node has been available to this build in any round: "
                       "the N > 1 path (rank launch, sharding by global lane index, RCCL tally exchange, nig_reduce_metrics) is "
                       "covered by 2-rank gloo tests on CPU and a 2-ranks-on-one-GPU rehearsal only; nig_reduce_metrics has only "
                       "ever seen a 1-rank RCCL communicator")
    elif rehearse:
        rec["note"] = "REHEARSAL (NIG_BENCH_REHEARSE): ranks share one device or none, gloo exchange -- not a scaling measurement"
    return rec


def measure_single_env(ni, n_steps=1000, warm=100, settle_s=0.25):
    """BASELINE configs[0] as written: ChemicalReactor-v0, batch = 1 env, a 1000-step rollout with reset on done -- the
    reference's own harness loop (performance_benchmark.py:106-133) on the single-env drop-in class (ni.make): one
    step-kernel launch + one host round trip per env.step through pinned staging.  Launch / PCIe latency bound by
    construction; it is the plumbing configuration, timed so the line carries every BASELINE config.
    Settle: like the fused records (`--settle`), the loop first runs untimed -- `settle_s` seconds of steps.  A fresh process
    on an idle GPU takes ~115 us per launch + round trip for its first ~50 ms and ~23 us from then on
    (profiles/r05/single_env_warmup.txt); 100 warm-up steps end inside that ramp, which is what made this record read
    16-18 k steps/s on some boxes and 45 k on others for the same build."""
    import numpy as np
    env = ni.make("ChemicalReactor-v0")
    rng = np.random.default_rng(0)
    env.reset()
    settled, t_s = 0, time.perf_counter()
    zero = np.zeros(3, dtype=np.float32)
    while time.perf_counter() - t_s < settle_s:
        _, _, te, tr, _ = env.step(zero)
        settled += 1
        if te or tr:
            env.reset()
    acts = rng.uniform(-1.0, 1.0, size=(n_steps + warm, 3)).astype(np.float32)    # action_space.sample()-equivalent
    env.reset()
    viol = eps = 0
    t0 = 0.0
    for i in range(n_steps + warm):
        if i == warm:
            t0 = time.perf_counter()
            viol = eps = 0
        obs, r, te, tr, info = env.step(acts[i])
        viol += info["safety_metrics"].violation_count
        if te or tr:
            eps += 1
            env.reset()
    dt = time.perf_counter() - t0
    env.close()
    return {"workload": f"ChemicalReactor-v0, batch=1 env, {n_steps}-step rollout, reset on done (BASELINE configs[0]; "
                        "single-env drop-in class, host buffers, one launch per env.step)",
            "value": n_steps / dt, "unit": "env-steps/s", "us_per_step": dt / n_steps * 1e6, "steps": n_steps,
            "episodes": eps, "violations": viol, "settle_steps": settled, "warmup_steps": warm,
            "reference_python_1core": REFERENCE_PYTHON_1CORE,
            "note": "PCIe/launch-latency bound (includes the host round trip of every step; never the headline `value`)"}


def spawn_ranks(n, argv):
    """`bench.py --gpus N` without a launcher: start the N ranks as a CHILD process group (torch.distributed.run, one
    rank per GPU, rendezvous on 127.0.0.1), relay rank 0's JSON line and verify it.  This process has imported neither
    torch nor the package and has made no HIP call (a process that has touched the GPU must not start another program
    in its place); the library is built here first if stale, so the ranks do not race to compile it."""
    import importlib.util
    import socket
    import subprocess
    preload = os.environ.get("LD_PRELOAD", "") + os.environ.get("ROCP_TOOL_LIBRARIES", "")
    if "rocprof" in preload:
        # a profiler preload has initialised the GPU in THIS process before main() ran: starting another program from it is
        # the exec-from-a-GPU-process pattern this pool forbids.  Profile one rank of a launcher-started job instead.
        raise SystemExit("bench.py: --gpus N > 1 under a profiler preload: start the ranks with torch.distributed.run and "
                         "profile inside it (or profile a --gpus 1 run); bench.py will not spawn from a GPU-initialised process")
    spec = importlib.util.spec_from_file_location("_nig_build", os.path.join(ROOT, "neorl-industrial-gym_amd", "_build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.ensure()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:                        # only rank 0 prints the result line; anything else is passed through
        if out.startswith("{") and '"metric"' in out:
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if rc != 0:
        raise SystemExit(f"bench.py: the {n}-rank child job exited with status {rc}")
    if line is None:
        raise SystemExit("bench.py: the child job printed no result line")
    rec = json.loads(line)
    if rec.get("n_gpus") != n or rec.get("ranks") != n or len(rec.get("episodes_per_rank", [])) != n:
        raise SystemExit(f"bench.py: asked for {n} ranks, the line reports n_gpus={rec.get('n_gpus')} ranks={rec.get('ranks')}")
    print(line)
    return 0


class RehearsalWorkload:
    """NIG_BENCH_REHEARSE=cpu (tests only, a box without a GPU): stands in for the device workload so that the N-rank
    CONTROL FLOW -- spawn, rendezvous, barriers, max-over-ranks timing, the tally all-gather + fixed-order combine and
    its self-check, the result line -- runs over gloo.  It computes nothing: `launch` is a no-op and the partial tally
    is a made-up, rank-dependent vector; the line it produces carries value null and says so."""

    def __init__(self, ni, torch, rank):
        self.ni, self.torch, self.rank = ni, torch, rank
        self.env = self

    def launch(self):
        pass

    def kernels_per_launch(self):
        return 1

    def reduce_tally(self):
        L = self.ni._lib
        p = self.torch.zeros(L.T_ROWS, dtype=self.torch.float64)
        p[L.T_EPISODES] = 100 + self.rank
        p[L.T_VIOL] = 7 * (self.rank + 1)
        p[L.T_CRIT] = self.rank
        p[L.T_RET_SUM] = -50.0 * (100 + self.rank)
        p[L.T_RET_MIN], p[L.T_RET_MAX] = -60.0 - self.rank, -40.0 + self.rank
        return p

    def close(self):
        pass


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed launches (one launch = --plan-steps env.step of every lane)")
    ap.add_argument("--warmup", type=int, default=20, help="untimed launches")
    ap.add_argument("--settle", type=float, default=0.6,
                    help="seconds of untimed back-to-back load before the warm-up launches (clock / memory-side ramp after idle)")
    ap.add_argument("--env", default="cr", choices=list(ENVS) + ["mixed"],
                    help="mixed = all env types in one padded SoA batch (BASELINE config 4), rollout mode, min outputs")
    ap.add_argument("--batch", type=int, default=0, help="lanes per GPU (default: BASELINE config of --env)")
    ap.add_argument("--mode", default="rollout", choices=["graph", "eager", "rollout"],
                    help="graph/eager: one step kernel per env.step (step-API); rollout: fused multi-step kernel")
    ap.add_argument("--plan-steps", type=int, default=250,
                    help="env.step per launch (rollout mode) / per hipGraph replay (graph mode)")
    ap.add_argument("--ring", type=int, default=250,
                    help="slots of the pre-filled action ring (default = --plan-steps' default: a lane's actions repeat every --ring steps, "
                         "and the headline's 196 MB ring still lives in the 256 MB Infinity Cache -- same launch time as 64 slots, "
                         "profiles/r05/headline_ring_length.txt -- while the timed workload's violations per episode move from 3.3 % "
                         "to 0.7 % under the reference's)")
    ap.add_argument("--ring-layout", default="rows", choices=["rows", "aos"],
                    help="action ring as [R, A, ld] rows (default) or row-major [R, B, A] slots (nig_rollout's ld_act == 0: read natively "
                         "by PowerGrid's wide form, transposed by the library per call for every other kernel form)")
    ap.add_argument("--outputs", default="full", choices=["full", "min", "none"],
                    help="rollout mode: full = obs trajectory + reward + flags per step; min = reward + flags; none")
    ap.add_argument("--traj", default="aos", choices=["aos", "soa"], help="observation trajectory layout: [T,B,S] or [T,S,ld]")
    ap.add_argument("--mixed-set", default="readme", choices=["readme", "survey"],
                    help="--env mixed: the README's seven envs, or SURVEY 8(d).4's seven (the three reference envs + the two "
                         "Advanced candidates + two README-only plants)")
    ap.add_argument("--mixed-outputs", default="full", choices=["full", "min"],
                    help="--env mixed / the mixed sub-record: full = per-step observation rows [T][S_max][ld] + reward + flags; min = reward + flags")
    ap.add_argument("--mixed-launch", default="fused", choices=["fused", "streams"],
                    help="--env mixed: one kernel launch over all segments (nig_create_mixed) or one launch per segment on its own stream")
    ap.add_argument("--no-step-api", action="store_true", help="skip the secondary step-API measurement")
    ap.add_argument("--no-powergrid", action="store_true", help="skip the secondary PowerGrid (BASELINE configs[2]/[4]) measurement")
    ap.add_argument("--no-robotassembly", action="store_true", help="skip the secondary RobotAssembly measurement (the third reference env, 262144 lanes)")
    ap.add_argument("--no-mixed", action="store_true", help="skip the secondary mixed-batch (BASELINE configs[3]) measurement")
    ap.add_argument("--no-single-env", action="store_true", help="skip the BASELINE configs[0] record (1 env, 1000 steps, host loop)")
    ap.add_argument("--no-brackets", action="store_true",
                    help="skip the headline's brackets (cold first launches, action rings larger than the Infinity Cache)")
    ap.add_argument("--calibrate", action="store_true", help="also run known-size dword copies (PMC calibration)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=4.0, help="wall budget of the all-core CPU baseline sample")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: this process becomes the launcher (nothing below runs here; no torch, no HIP call so far)
        return spawn_ranks(args.gpus, sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus {args.gpus} ...` "
                         f"(it launches its own ranks) or `python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} "
                         f"--master-addr 127.0.0.1 --master-port P bench.py --gpus {args.gpus} ...`")

    # one rank per GPU: host threads onto the GPU's NUMA node, before torch (and its thread pools) are imported
    my_affinity = pin_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))

    import torch
    import torch.distributed as dist

    # the package (and with it libnig.so, built here if stale) is loaded BEFORE anything touches the GPU
    import neorl_industrial_gym_amd as ni

    # NIG_BENCH_REHEARSE=1: every rank on cuda:0 with the gloo backend (a 1-GPU box cannot host two RCCL ranks);
    # NIG_BENCH_REHEARSE=cpu: no GPU at all, control flow only (RehearsalWorkload).  Neither ever gives a reported number.
    mode_r = os.environ.get("NIG_BENCH_REHEARSE", "")
    rehearse = mode_r in ("1", "cpu")
    if mode_r == "cpu":
        device = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
        dev_index = 0 if rehearse else local_rank
        torch.cuda.set_device(dev_index)
        device = torch.device("cuda", dev_index)
    # NIG_BENCH_FORCE_PG=1: a process group even at world == 1 -- a one-rank "nccl" (RCCL) group is legal on one GPU and takes
    # every collective branch below; it yields no scaling number (scale.measured_on_hardware stays false)
    force_pg = os.environ.get("NIG_BENCH_FORCE_PG") == "1"
    if world > 1 or force_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:          # (FORCE_PG without a launcher)
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)   # RCCL over xGMI
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")
        if force_pg:
            from neorl_industrial_gym_amd import parallel as _par
            _par.ALWAYS_COLLECTIVE = True             # the tally all-gather goes through the backend even with one rank
    comm_dev = torch.device("cpu") if rehearse else device
    if mode_r != "cpu":
        global _PROBE, _DPM_DIR
        _PROBE, _DPM_DIR = ClockProbe(ni, torch, device), gpu_sysfs_dir(torch, device.index or 0)
    affinity = [my_affinity]
    if grouped(dist):
        affinity = [None] * world
        dist.all_gather_object(affinity, my_affinity)
    if mode_r == "cpu":
        return rehearse_cpu(args, ni, torch, dist, comm_dev, world, rank, affinity)

    if args.env == "mixed":
        return bench_mixed(args, ni, torch, dist, device, comm_dev, world, rank)
    key = args.env
    B = args.batch or BASELINE_BATCH[key]
    P, K, W = max(1, args.plan_steps), max(1, args.steps), max(0, args.warmup)

    # Order of the GPU work: the secondary measurements run FIRST and the headline LAST, right behind them, so the
    # headline's W warm-up launches start on a chip that is already at its sustained clocks (a cold start costs the
    # first ~10 ms of launches ~15 %; with the driver's --steps 20 the whole timed region is 5 ms).  The CPU halves
    # of the parity probe and the CPU baseline run after all GPU timing.
    parity_gpu = None
    if rank == 0 and not args.no_parity:
        parity_gpu = parity_probe_gpu(ni, torch, key, B, device)

    # ---- BASELINE configs[0]: one env, 1000 steps, reset on done (the plumbing configuration; ~50 ms)
    single_env = None
    if rank == 0 and key == "cr" and not args.no_single_env:
        single_env = measure_single_env(ni)

    # ---- secondary: the step API (one kernel launch per env.step, hipGraph replay of P of them)
    step_api = None
    if args.mode == "rollout" and not args.no_step_api:
        w2 = Workload(ni, torch, key, B, device, rank, "graph", P, min(args.ring, 64), args.outputs, args.traj)     # (a 64-slot ring: the step API's working set stays what it was)
        K2 = max(2, min(K, 8))
        sw, sd = timed(torch, dist, world, comm_dev, w2, K2, 1)
        r2 = roofline_of(w2, K2, sd)
        step_api = {"value": K2 * P * B * world / sw, "unit": "env-steps/s", "steps": K2, "launch_us": r2["launch_us"],
                    "alg_bytes_per_env_step": r2["alg_bytes_per_env_step"], "achieved_GBps": r2["achieved"],
                    "frac_of_hbm_peak": r2["frac_algorithmic"], "hbm_side_frac": r2["frac"], "hbm_side_model": r2["hbm_side_model"],
                    "traffic": r2["traffic"],
                    "note": f"one step_kernel launch per env.step, {P} per hipGraph replay"}
        w2.close()
        del w2

    # ---- secondary: PowerGrid-v0, 262144 lanes per GPU (BASELINE configs[2]; configs[4] = this at N = 8)
    powergrid = None
    if key == "cr" and args.mode == "rollout" and not args.no_powergrid:
        Bp = BASELINE_BATCH["pg"]
        # action ring of 34 slots = 285 MB: larger than the 256 MB Infinity Cache, so every action read of this record comes
        # from HBM and all 168 B per env-step are HBM-side bytes (roofline_of; a 16-slot ring sat inside the cache)
        # ... as ROW-MAJOR slots [R][B][A] -- the layout of a policy's batched output, nig_rollout's ld_act == 0 -- which PowerGrid's
        # wide form reads natively: a lane's eight actions are two 16-byte loads, 2 KiB contiguous per wave, where [A][ld] rows are
        # eight 256-byte segments 1 MB apart; same bytes, same values (tests/test_gpu_action_layout.py), 1 940-1 947 vs 1 995-2 183 us
        # on one box (profiles/r05/pg_ring_layout_ab.txt).  The rows form is timed right behind it for the record (`rows_ring`).
        w3 = Workload(ni, torch, "pg", Bp, device, rank, "rollout", P, 34, args.outputs, args.traj, ring_layout="aos")
        K3 = max(2, min(K, 8))
        pg_times = {}
        pw, pd = timed(torch, dist, world, comm_dev, w3, K3, 2, args.settle, stats=pg_times)
        r3 = roofline_of(w3, K3, pd)
        r3["action_ring_layout"] = "row-major [R][B][A] slots (ld_act == 0), read natively by rollout_wide_kernel"
        ptotal, pcheck = gathered_tally(torch, dist, world, comm_dev, w3)
        L = ni._lib
        powergrid = {"workload": f"PowerGrid-v0, batch={Bp} per GPU x {world} GPU(s) = {Bp * world} lanes, fused rollout, "
                                 f"{P} env.step per launch, outputs: {args.outputs}",
                     "value": K3 * P * Bp * world / pw, "unit": "env-steps/s", "steps": K3, "ms_per_step": pw * 1e3 / K3,
                     "roofline": r3, "rank_times": pg_times,
                     "tally": {"episodes": int(ptotal[L.T_EPISODES]), "violations": int(ptotal[L.T_VIOL]),
                               "critical": int(ptotal[L.T_CRIT])}, "tally_check": pcheck,
                     "fast_mode_statistics": fast_mode_statistics(ni, "pg", ptotal, Bp, 0 if args.no_parity else 1, device) if rank == 0 else None}
        w3.close()
        del w3
        w3r = Workload(ni, torch, "pg", Bp, device, rank, "rollout", P, 34, args.outputs, args.traj)          # the same record with [A][ld] rows
        rw_, rd3 = timed(torch, dist, world, comm_dev, w3r, K3, 2, min(args.settle, 0.3))
        rr = roofline_of(w3r, K3, rd3)
        powergrid["rows_ring"] = {"ms_per_step": rw_ * 1e3 / K3, "launch_us": rr["launch_us"], "frac": rr["frac"],
                                  "action_ring_layout": "[R][A][ld] rows"}
        w3r.close()
        del w3r

    # ---- secondary: RobotAssembly-v0, 262144 lanes per GPU: the third reference env (robot_assembly.py:139-188, SURVEY rows
    # a15-a19) has no BASELINE config of its own; timed at PowerGrid's batch so the line carries all three (VERDICT r04 next #6b)
    robotassembly = None
    if key == "cr" and args.mode == "rollout" and not args.no_robotassembly:
        Br = BASELINE_BATCH["ra"]
        # 40 ring slots = 294 MB: larger than the Infinity Cache, every action read is an HBM read (as for PowerGrid above)
        w4 = Workload(ni, torch, "ra", Br, device, rank, "rollout", P, 40, args.outputs, args.traj)
        # RobotAssembly episodes run up to 1 000 steps and its violation counts depend on the joints' RANDOM WALK: under a
        # 40-slot ring (a lane's actions repeating every 40 steps) the timed workload's violations per episode came out 36 %
        # under the reference's (16.3 vs 25.6).  Four rings of P slots cycled launch by launch (7.3 GB, HBM-resident like the
        # 40-slot ring was) repeat only after 1 000 steps = the longest possible episode
        w4.use_big_ring(4 * P * w4.A * w4.env.ld * 4 - 1)
        K4 = max(2, min(K, 8))
        ra_times = {}
        rw, rd_ = timed(torch, dist, world, comm_dev, w4, K4, 2, args.settle, stats=ra_times)
        r4 = roofline_of(w4, K4, rd_)
        rtotal, rcheck = gathered_tally(torch, dist, world, comm_dev, w4)
        L = ni._lib
        robotassembly = {"workload": f"RobotAssembly-v0, batch={Br} per GPU x {world} GPU(s) = {Br * world} lanes, fused rollout, "
                                     f"{P} env.step per launch, outputs: {args.outputs}",
                         "value": K4 * P * Br * world / rw, "unit": "env-steps/s", "steps": K4, "ms_per_step": rw * 1e3 / K4,
                         "roofline": r4, "rank_times": ra_times,
                         "tally": {"episodes": int(rtotal[L.T_EPISODES]), "violations": int(rtotal[L.T_VIOL]),
                                   "critical": int(rtotal[L.T_CRIT])}, "tally_check": rcheck,
                         "fast_mode_statistics": fast_mode_statistics(ni, "ra", rtotal, Br, 0 if args.no_parity else 1, device) if rank == 0 else None}
        w4.close()
        del w4

    # ---- secondary: all 7 envs mixed-batch, 1 048 576 lanes per GPU (BASELINE configs[3]), ONE fused launch
    mixed = None
    if key == "cr" and args.mode == "rollout" and not args.no_mixed:
        mixed = measure_mixed(args, ni, torch, dist, device, comm_dev, world, rank, 1048576, max(2, min(K, 8)), 2,
                              args.settle, per_env_rates=False)

    # ---- headline: workload resident in HBM before the timed region
    wl = Workload(ni, torch, key, B, device, rank, args.mode, P, args.ring, args.outputs, args.traj, ring_layout=args.ring_layout)
    rank_times = {}
    wall, dev_ms = timed(torch, dist, world, comm_dev, wl, K, W, args.settle, stats=rank_times)
    roof = roofline_of(wl, K, dev_ms)
    total, tally_check = gathered_tally(torch, dist, world, comm_dev, wl)
    if args.mode == "rollout" and not args.no_brackets:
        roof.update(honest_brackets(torch, wl, K, roof, args.settle, phases=rank_times.get("phases")))

    if args.calibrate:   # known-size dword-per-lane copies for the PMC byte calibration (profiles/)
        cal = torch.empty(wl.S, wl.env.ld, dtype=torch.float32, device=device)
        for _ in range(20):
            ni._lib.check(wl.env._L.nig_get_state(wl.env._h, cal.data_ptr(), wl.env.ld, None, wl.env._stream()))
        torch.cuda.synchronize()
    S, A = wl.S, wl.A
    wl.close()
    del wl
    parity = parity_probe_cpu(parity_gpu, key, B) if parity_gpu is not None else None
    fast_stats = None
    if rank == 0 and key in ENVS and args.mode == "rollout":     # untimed, after every timed region: a few launches with fresh actions
        fast_stats = fast_mode_statistics(ni, key, total, B, 0 if args.no_parity else (2 if key == "cr" else 1), device)

    if rank == 0:
        L = ni._lib
        out = {
            "metric": "env-steps/sec (whole node) + safety-violation-count parity, ChemicalReactor-v0"
                      if key == "cr" else f"env-steps/sec (whole node), {ENVS[key]}",
            "value": K * P * B * world / wall, "unit": "env-steps/s", "n_gpus": world, "ranks": tally_check["ranks"],
            "episodes_per_rank": tally_check["episodes_per_rank"], "steps": K, "warmup": W,
            "ms_per_step": wall * 1e3 / K, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{ENVS[key]}, batch={B} parallel envs per GPU; one bench step = "
                                   + (f"one fused rollout launch of {P} env.step per lane (state in registers, "
                                      f"per-step outputs: {args.outputs})"
                                      if args.mode == "rollout" else
                                      f"{P} step-API launches (one fused step kernel per env.step), {args.mode}"),
                       "batch_per_gpu": B, "global_batch": B * world, "plan_steps": P, "env_steps_per_step": P * B * world,
                       "action_ring": args.ring, "autoreset": True, "episode_tally": True,
                       "settle_seconds": args.settle,
                       "parallelism": f"env-shard x{world} (no data-path collective)"},
            "roofline": roof,
            "rank_times": rank_times,
            "scale": scale_record(world, rehearse, affinity, dist),
            "tally": {"episodes": int(total[L.T_EPISODES]), "violations": int(total[L.T_VIOL]),
                      "critical": int(total[L.T_CRIT]), "return_mean": float(total[L.T_RET_SUM] / max(total[L.T_EPISODES], 1))},
            "tally_check": tally_check,
        }
        if single_env is not None:
            out["single_env"] = single_env
        if step_api is not None:
            out["step_api"] = step_api
        if powergrid is not None:
            out["powergrid"] = powergrid
        if robotassembly is not None:
            out["robotassembly"] = robotassembly
        if mixed is not None:
            out["mixed"] = mixed
        if parity is not None or key in ENVS:
            out["parity"] = dict(parity or {}, fast_mode_statistics=fast_stats)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(key, B, args.cpu_seconds)
        print(json.dumps(out))
    if grouped(dist):
        dist.destroy_process_group()


def rehearse_cpu(args, ni, torch, dist, comm_dev, world, rank, affinity):
    """Control flow of an N-rank run on a box without a GPU (tests/test_host_logic.py).  No measurement: value null."""
    wl = RehearsalWorkload(ni, torch, rank)
    K, W = max(1, args.steps), max(0, args.warmup)
    rank_times = {}
    wall, _ = timed(torch, dist, world, comm_dev, wl, K, W, stats=rank_times)
    total, check = gathered_tally(torch, dist, world, comm_dev, wl)
    if rank == 0:
        L = ni._lib
        print(json.dumps({"metric": "env-steps/sec (whole node) + safety-violation-count parity, ChemicalReactor-v0",
                          "value": None, "unit": "env-steps/s", "n_gpus": world, "ranks": check["ranks"],
                          "episodes_per_rank": check["episodes_per_rank"], "steps": K, "warmup": W,
                          "ms_per_step": wall * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f32", "data": "none", "rehearsal": "cpu control flow only: no device workload ran",
                          "rank_times": rank_times, "scale": scale_record(world, True, affinity, dist),
                          "tally": {"episodes": int(total[L.T_EPISODES]), "violations": int(total[L.T_VIOL]),
                                    "critical": int(total[L.T_CRIT])}, "tally_check": check}))
    if grouped(dist):
        dist.destroy_process_group()
    return 0


MIXED7 = [("ChemicalReactor-v0", 12, 3, "reference"), ("RobotAssembly-v0", 24, 7, "reference"),
          ("HVACControl-v0", 18, 5, "build-specified"), ("WaterTreatment-v0", 15, 4, "build-specified"),
          ("SteelAnnealing-v0", 20, 6, "build-specified"), ("PowerGrid-v0", 32, 8, "reference"),
          ("SupplyChain-v0", 28, 10, "build-specified")]


MIXED7_SURVEY = [("ChemicalReactor-v0", 12, 3, "reference"), ("PowerGrid-v0", 32, 8, "reference"),
                 ("RobotAssembly-v0", 24, 7, "reference"),
                 ("AdvancedChemicalReactor-v0", 20, 6, "restated from source text (not instantiable upstream)"),
                 ("AdvancedPowerGrid-v0", 32, 8, "restated from source text (not instantiable upstream)"),
                 ("HVACControl-v0", 18, 5, "build-specified"), ("WaterTreatment-v0", 15, 4, "build-specified")]


def measure_mixed(args, ni, torch, dist, device, comm_dev, world, rank, B, K, W, settle_s, per_env_rates=True):
    """BASELINE config 4: the README's seven environments (README.md:24-32) in ONE padded SoA batch of B lanes per
    GPU, contiguous 256-aligned segments of equal size.  --mixed-launch fused: ONE kernel launch over all segments
    (block -> env table, heaviest envs first); streams: one rollout kernel per segment on its own stream (round-1
    form).  Three of the seven exist upstream and are parity-checked against the reference; four are README-only
    there and run build-specified plants (flagged per env).  --mixed-outputs full: every step's observation rows go to
    a padded [T][S_max][ld] trajectory (what get_dataset stores); min: reward + flag word only.  The byte model is the
    lane-weighted fused-rollout figure of the outputs requested."""
    per = (B // 7) // 256 * 256
    envset = MIXED7 if args.mixed_set == "readme" else MIXED7_SURVEY
    counts = [(name, per if i else B - 6 * per) for i, (name, _, _, _) in enumerate(envset)]
    dims = {name: (S, A) for name, S, A, _ in envset}
    origin = {name: o for name, _, _, o in envset}
    fused = args.mixed_launch == "fused"
    full = args.mixed_outputs == "full"
    mix = ni.MixedBatchedEnv(counts, device=device, seed=0x5EED, autoreset=True, tally=True, env_index0=rank * B,
                             fused=fused)
    P = max(1, args.plan_steps)
    R = min(args.ring, 16)
    ring = torch.zeros(R, mix.A_max, mix.ld, dtype=torch.float32, device=device)
    for s in range(R):
        mix.fill_actions(1000 + s, ring[s])
    rew = torch.empty(P, mix.ld, dtype=torch.float32, device=device)
    fl = torch.empty(P, mix.ld, dtype=torch.int32, device=device)
    obs = None
    if full:
        try:    # 33.5 GB at the default size (250 steps x 32 rows x 1 048 576 lanes): fall back to reward + flags if it does not fit
            obs = torch.empty(P, mix.S_max, mix.ld, dtype=torch.float32, device=device)
        except torch.cuda.OutOfMemoryError:
            full = False
    mix.reset()

    class _W:
        def launch(self):
            mix.rollout(P, ring, rew, fl, obs)

        def kernels_per_launch(self):
            return 1 if fused else len(mix.envs)
    wall, dev_ms = timed(torch, dist, world, comm_dev, _W(), K, W, settle_s)
    # per-env rate measured separately on its own segment size (same per-env kernels, stand-alone launch)
    per_env = {}
    for (name, n), seg, o in zip(counts, mix.envs, mix.offsets):
        per_env[name] = {"lanes": n, "state_dim": dims[name][0], "action_dim": dims[name][1],
                         "dynamics": origin[name], "reference_parity": origin[name] == "reference"}
        if per_env_rates:
            reps = max(1, K // 4)
            torch.cuda.synchronize(); c0 = time.perf_counter()
            for _ in range(reps):
                seg.rollout(P, ring[:, :seg.action_dim, o:o + n], rew[:, o:o + n], fl[:, o:o + n],
                            None if obs is None else obs[:, :seg.state_dim, o:o + n])
            torch.cuda.synchronize()
            per_env[name]["env_steps_per_s"] = reps * P * n / (time.perf_counter() - c0)
    from neorl_industrial_gym_amd.parallel import combine_partials
    _, tally_check = gathered_partial(world, comm_dev, ni, combine_partials(torch.stack(mix.reduce_tally())))
    bytes_launch = sum((4 * v["action_dim"] + 8 + (4 * v["state_dim"] if full else 0)) * v["lanes"] for v in per_env.values()) * P
    launch_us = dev_ms * 1e3 / K
    achieved = bytes_launch / (launch_us * 1e-6) / 1e9
    out_mode = 2 if full else 1
    per_step, tsrc = measured_traffic("mixed", B, "rollout", "full" if full else "min", P) if (fused and args.mixed_set == "readme") else (None, None)
    # HBM-side bytes (roofline_of): the per-step outputs always exceed the Infinity Cache at this size; the action ring
    # (R slots x A_max rows x ld) does when it is larger than the cache
    out_bytes = sum((8 + (4 * v["state_dim"] if full else 0)) * v["lanes"] for v in per_env.values()) * P
    ring_foot = int(ring.numel()) * 4
    hbm_bytes = (out_bytes if out_bytes > MALL_BYTES else 0) + ((bytes_launch - out_bytes) if ring_foot > MALL_BYTES else 0)
    rec = {"value": K * P * B * world / wall, "unit": "env-steps/s", "steps": K, "warmup": W, "ms_per_step": wall * 1e3 / K,
           "config": {"workload": f"mixed padded-SoA batch of {B} lanes per GPU (S_max={mix.S_max}, A_max={mix.A_max}): "
                                  + ", ".join(f"{n} x {e}" for e, n in counts) + f"; one bench step = {P} env.step per lane, "
                                  + ("ONE fused launch over all segments" if fused else "one launch per segment on 7 streams")
                                  + (", outputs: observation rows [T][S_max][ld] + reward + flags" if full else ", reward+flags outputs"),
                      "batch_per_gpu": B, "plan_steps": P, "env_steps_per_step": P * B * world, "segments": counts,
                      "launch": args.mixed_launch, "outputs": "full" if full else "min"},
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": hbm_bytes / (launch_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "frac_algorithmic": achieved / HBM_PEAK_GBS,
                        "hbm_side_bytes_per_launch": hbm_bytes,
                        "hbm_side_model": {"outputs_footprint_bytes": out_bytes, "action_ring_footprint_bytes": ring_foot, "cache_bytes": MALL_BYTES},
                        "traffic": None if per_step is None else per_step * B * P, "traffic_bytes_per_env_step": per_step,
                        "traffic_source": tsrc,
                        "kernel": ("mixed_rollout_kernel<%d>" % out_mode) if fused else "rollout_kernel<*,%d> x7 (concurrent streams)" % out_mode,
                        "alg_bytes_per_launch": bytes_launch, "launch_us": launch_us,
                        "bytes_model": "fused-rollout figure: action read + requested per-step outputs per env-step, lane-weighted"},
           "tally_check": tally_check, "per_env": per_env}
    mix.close()
    return rec


def bench_mixed(args, ni, torch, dist, device, comm_dev, world, rank):
    """--env mixed: BASELINE config 4 as the headline of its own line."""
    B = args.batch or 1048576
    K, W = max(1, args.steps), max(0, args.warmup)
    rec = measure_mixed(args, ni, torch, dist, device, comm_dev, world, rank, B, K, W, args.settle)
    if args.calibrate:   # known-size dword-per-lane copies for the PMC byte calibration (profiles/pmc_to_traffic.py): 20 x 12 rows x 65 536 lanes
        cenv = ni.make_batched("ChemicalReactor-v0", 65536, device=device)
        cal = torch.empty(12, cenv.ld, dtype=torch.float32, device=device)
        for _ in range(20):
            ni._lib.check(cenv._L.nig_get_state(cenv._h, cal.data_ptr(), cenv.ld, None, cenv._stream()))
        torch.cuda.synchronize()
        cenv.close()
    if rank == 0:
        print(json.dumps({
            "metric": "env-steps/sec (whole node), all 7 envs mixed-batch", "value": rec["value"],
            "unit": "env-steps/s", "n_gpus": world, "ranks": rec["tally_check"]["ranks"],
            "episodes_per_rank": rec["tally_check"]["episodes_per_rank"], "steps": K, "warmup": W,
            "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "config": rec["config"], "roofline": rec["roofline"],
            "per_env": rec["per_env"]}))
    if grouped(dist):
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
