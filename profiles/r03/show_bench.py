#!/usr/bin/env python3
"""Print the figures of a bench.py result line that the round notes quote (usage: show_bench.py file.json)."""
import json
import sys

r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
roof = r["roofline"]
print("headline  %.4g env-steps/s  ms_per_step %.4f  n_gpus %s ranks %s" % (r["value"], r["ms_per_step"], r["n_gpus"], r.get("ranks")))
print("  kernel %s  launch_us %.1f  frac %.3f  traffic %s" % (roof["kernel"], roof["launch_us"], roof["frac"], roof.get("traffic")))
for k in ("write_only_frac", "cold_first_launches", "ring_gt_mall"):
    if k in roof:
        print("  %s: %s" % (k, roof[k]))
for k in ("powergrid", "mixed"):
    if k in r:
        q = r[k]
        print("%s  %.4g env-steps/s  ms_per_step %.3f  frac %.3f  %s" % (k, q["value"], q["ms_per_step"], q["roofline"]["frac"], q["roofline"]["kernel"]))
if "step_api" in r:
    print("step_api  launch_us %.2f  frac %.3f" % (r["step_api"]["launch_us"], r["step_api"]["frac_of_hbm_peak"]))
if "cpu_baseline" in r:
    print("cpu  %.4g env-steps/s on %d cores; 1 thread %.4g" % (r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"], r["cpu_baseline"]["threads_1"]["value"]))
if "parity" in r:
    print("parity", r["parity"])
