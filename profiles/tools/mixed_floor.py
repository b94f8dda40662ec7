#!/usr/bin/env python3
"""The mixed launch (BASELINE configs[3], reward + flags) against its bodies' own issue rates.

Run under rocprofv3 --pmc (SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE): `run` launches
mixed_rollout_kernel<1> REPS times over ONE env type at a time (the whole batch in that env: the body under the mixed kernel's
register allocation, the chip full) and then REPS times over the 7-env batch, and prints the dispatch plan; `table` reads the
counter csv, cuts the kernel's dispatches by that plan and prints per body: vector instructions per wave-step, chip cycles
per launch, cycles per vector instruction per SIMD -- and for the 7-env launch the same figures beside the lane-weighted
prediction from the bodies (what the launch would take if every body issued at its own stand-alone rate).
    rocprofv3 --pmc ... -- python3 profiles/tools/mixed_floor.py run [lanes] [steps]
    python3 profiles/tools/mixed_floor.py table <counter_collection.csv> [lanes] [steps]"""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
REPS, WARM = 4, 1
NAMES = ["ChemicalReactor-v0", "RobotAssembly-v0", "HVACControl-v0", "WaterTreatment-v0", "SteelAnnealing-v0", "PowerGrid-v0", "SupplyChain-v0"]


def run(B, P):
    import torch
    import neorl_industrial_gym_amd as ni
    dev = "cuda:0"

    def go(counts):
        mix = ni.MixedBatchedEnv(counts, device=dev, seed=0x5EED, autoreset=True, tally=True, fused=True)
        ring = torch.zeros(16, mix.A_max, mix.ld, dtype=torch.float32, device=dev)
        for s in range(16):
            mix.fill_actions(1000 + s, ring[s])
        rew = torch.empty(P, mix.ld, dtype=torch.float32, device=dev)
        fl = torch.empty(P, mix.ld, dtype=torch.int32, device=dev)
        mix.reset()
        for _ in range(WARM + REPS):
            mix.rollout(P, ring, rew, fl, None)
            torch.cuda.synchronize()
        mix.close()
    for name in NAMES:
        go([(name, B)])
    per = (B // 7) // 256 * 256
    go([(name, per if i else B - 6 * per) for i, name in enumerate(NAMES)])
    # the same bodies in their STAND-ALONE kernels (nig_rollout of a one-env handle: the form the library picks for B lanes, its own
    # register allocation and occupancy), reward + flags outputs: the rate the mixed kernel's bodies are measured against
    for name in NAMES:
        env = ni.make_batched(name, B, device=dev, seed=0x5EED, autoreset=True, tally=True)
        ring = torch.zeros(16, env.action_dim, env.ld, dtype=torch.float32, device=dev)
        for s in range(16):
            env.fill_actions(1000 + s, ring[s])
        rew = torch.empty(P, env.ld, dtype=torch.float32, device=dev)
        fl = torch.empty(P, env.ld, dtype=torch.int32, device=dev)
        env.reset()
        for _ in range(WARM + REPS):
            env.rollout(P, ring, rew, fl)
            torch.cuda.synchronize()
        env.close()
    print(json.dumps({"plan": NAMES + ["7-env"], "launches_each": WARM + REPS, "warm": WARM, "lanes": B, "steps": P}))


def table(path, B, P):
    rows, alone = defaultdict(dict), defaultdict(lambda: defaultdict(dict))
    for r in csv.DictReader(open(path)):
        if "mixed_rollout_kernel<1>" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
        elif any(x in r["Kernel_Name"] for x in ("rollout_kernel<", "rollout_wide_kernel<", "split_rollout_kernel<")) and ", 1," in r["Kernel_Name"]:
            alone[r["Kernel_Name"]][int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(rows)
    each = WARM + REPS
    assert len(ids) == each * (len(NAMES) + 1), (len(ids), each)
    out = {}
    for k, name in enumerate(NAMES + ["7-env"]):
        sel = [rows[i] for i in ids[k * each + WARM:(k + 1) * each]]
        med = lambda c: sorted(x[c] for x in sel)[len(sel) // 2]
        waves, valu, gui = med("SQ_WAVES"), med("SQ_INSTS_VALU"), med("GRBM_GUI_ACTIVE") / 8.0      # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        out[name] = {"valu_per_wave_step": valu / (waves * P), "chip_cycles": gui, "cycles_per_valu_per_simd": gui * 1024.0 / valu,
                     "valu_total": valu, "waves": waves}
    print("%-22s %12s %14s %12s" % ("body", "VALU/wave-step", "chip cycles", "cyc/VALU/SIMD"))
    for name, v in out.items():
        print("%-22s %12.1f %14.4g %12.3f" % (name, v["valu_per_wave_step"], v["chip_cycles"], v["cycles_per_valu_per_simd"]))
    # prediction for the 7-env launch: every body's instructions (its share of the lanes) at its own stand-alone rate
    per = (B // 7) // 256 * 256
    share = {name: (per if i else B - 6 * per) / B for i, name in enumerate(NAMES)}
    pred_cycles = sum(out[n]["chip_cycles"] * share[n] for n in NAMES)
    got = out["7-env"]["chip_cycles"]
    print("7-env launch: %.4g chip cycles measured, %.4g predicted from the bodies at their own rates (%.1f %% over)" % (got, pred_cycles, 100 * (got / pred_cycles - 1)))
    print("\nstand-alone kernels (nig_rollout of a one-env handle, the same lanes and outputs):")
    sa = {}
    for kname, disp in alone.items():
        ids2 = sorted(disp)[WARM:]                   # (a kernel that ran for several handles: every launch counts, warm-ups of the first dropped)
        sel = [disp[i] for i in ids2 if "SQ_INSTS_VALU" in disp[i]]
        if not sel:
            continue
        med = lambda c: sorted(x[c] for x in sel)[len(sel) // 2]
        waves, valu, gui = med("SQ_WAVES"), med("SQ_INSTS_VALU"), med("GRBM_GUI_ACTIVE") / 8.0
        short = kname.replace("void nig::", "").split("(")[0]
        sa[short] = {"valu_per_wave_step": valu / (waves * P), "chip_cycles": gui, "cycles_per_valu_per_simd": gui * 1024.0 / valu, "waves": waves}
        print("%-70s %10.1f %12.4g %10.3f  (%d waves)" % (short[:70], valu / (waves * P), gui, gui * 1024.0 / valu, waves))
    print(json.dumps({"bodies": out, "predicted_cycles": pred_cycles, "measured_cycles": got, "stand_alone": sa}))


if __name__ == "__main__":
    B = int(sys.argv[3 if sys.argv[1] == "table" else 2]) if len(sys.argv) > (3 if sys.argv[1] == "table" else 2) else 1048576
    P = 250
    if sys.argv[1] == "run":
        run(B, P)
    else:
        table(sys.argv[2], B, P)
