#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE csvs into HBM bytes per launch and per env-step.

Calibration (MI355X_MICROARCH.md "HBM"): on gfx950 FETCH_SIZE is exact only for some access
widths (it reads 1/2 for 16 B/lane streams) and "other access widths are uncalibrated", so the
byte scale for THIS code's dword-per-lane row accesses is taken from copy_rows_kernel dispatches
of known size in the same run (bench.py --calibrate): scale = known bytes / counter value.
Writes gpurun_out/pmc_<tag>/traffic_<tag>.json; merge the entry into profiles/traffic.json.
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict


ROLLOUT_RX = r"rollout_kernel|rollout_wide_kernel|split_rollout_kernel|rollout_pg_pair_kernel|mixed_rollout_kernel"     # the fused-rollout kernel forms
ENV_NAME = {"cr": "ChemicalReactor", "pg": "PowerGrid", "ra": "RobotAssembly"}
ROUND = os.environ.get("NIG_PROFILE_ROUND", "r04")                           # directory under profiles/ the record is kept in


def load(path):
    per = defaultdict(list)
    for r in csv.DictReader(open(path)):
        per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return per


def main():
    out, tag = sys.argv[1], sys.argv[2]
    args = sys.argv[3:]
    env = args[args.index("--env") + 1] if "--env" in args else "cr"
    S = {"cr": 12, "pg": 32, "ra": 24, "mixed": 12}[env]
    B = int(args[args.index("--batch") + 1]) if "--batch" in args else {"cr": 65536, "pg": 262144, "ra": 262144, "mixed": 1048576}[env]
    calB = 65536 if env == "mixed" else B           # bench_mixed calibrates on a 65 536-lane ChemicalReactor handle
    P = int(args[args.index("--plan-steps") + 1]) if "--plan-steps" in args else 250
    mode = args[args.index("--mode") + 1] if "--mode" in args else "rollout"
    res = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        per = load(os.path.join(out, f"{tag}_{ctr}.csv"))
        cal = [v for k, vs in per.items() if "copy_rows_kernel" in k for v in vs]
        known = S * calB * 4.0                    # bytes read (and written) by one copy_rows dispatch
        cal = sorted(cal)[len(cal) // 2] if cal else None
        scale = (known / cal) if cal else None    # bytes per counter unit for dword-per-lane rows
        for k, vs in per.items():
            # (a default bench run also times its PowerGrid sub-record: only the kernels of THIS env count)
            if env in ENV_NAME and ENV_NAME[env] not in k:
                continue
            if re.search(ROLLOUT_RX + r"|step_kernel", k):
                vs = sorted(vs)
                med = vs[len(vs) // 2]
                res.setdefault(k, {})[ctr] = {"median_counter": med, "calibrated_scale_bytes_per_unit": scale,
                                              "bytes_per_launch": med * scale if scale else None,
                                              "copy_rows_counter": cal, "copy_rows_known_bytes": known, "n": len(vs)}
    summary = {}
    for k, d in res.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d and d["FETCH_SIZE"]["bytes_per_launch"] is not None:
            per_launch = d["FETCH_SIZE"]["bytes_per_launch"] + d["WRITE_SIZE"]["bytes_per_launch"]
            env_steps = B * (P if re.search(ROLLOUT_RX, k) else 1)          # env-steps one kernel launch processes
            summary[k] = {"hbm_bytes_per_env_step": per_launch / env_steps, "plan_steps": P, "mode": mode,
                          "env_steps_per_launch": env_steps, "hbm_bytes_per_launch": per_launch,
                          "fetch_bytes": d["FETCH_SIZE"]["bytes_per_launch"], "write_bytes": d["WRITE_SIZE"]["bytes_per_launch"],
                          "kernel": k, "detail": d}
    path = os.path.join(out, f"traffic_{tag}.json")
    json.dump(summary, open(path, "w"), indent=1)
    print(json.dumps(summary, indent=1))
    # merge into profiles/traffic.json under the key bench.py looks up: <env>_<B>_<mode>_<outputs|step>
    outputs = args[args.index("--outputs") + 1] if "--outputs" in args else "full"
    if env == "mixed":
        outputs = args[args.index("--mixed-outputs") + 1] if "--mixed-outputs" in args else "full"
    want = ("mixed_rollout_kernel" if env == "mixed" else ROLLOUT_RX) if mode == "rollout" else "step_kernel"
    tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "traffic.json")
    try:
        allt = json.load(open(tpath))
    except Exception:
        allt = {}
    group = [k for k in summary if re.search(want, k)]
    if len(group) > 1 and len({summary[k]["detail"]["FETCH_SIZE"]["n"] for k in group}) == 1:
        # one bench step = several kernels of this env launched back to back (RobotAssembly beyond one residency round: the full
        # rounds one-wave + the tail three-wave): their bytes add up to the launch's
        tot = {f: sum(summary[k][f] for k in group) for f in ("hbm_bytes_per_launch", "fetch_bytes", "write_bytes")}
        k0 = group[0]
        summary = {" + ".join(group): dict(summary[k0], kernel=" + ".join(group), hbm_bytes_per_env_step=tot["hbm_bytes_per_launch"] / (B * P),
                                           env_steps_per_launch=B * P, **tot)}
    for k, v in summary.items():
        if re.search(want, k):
            rec = {kk: vv for kk, vv in v.items() if kk != "detail"}
            rec["source"] = f"profiles/{ROUND}/{tag}_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, calibrated on copy_rows_kernel)"
            allt[f"{env}_{B}_{mode}_{outputs if mode == 'rollout' else 'step'}"] = rec
    json.dump(allt, open(tpath, "w"), indent=1)


if __name__ == "__main__":
    main()
