#!/bin/bash
# usage: bash profiles/sweep.sh <tag>     (GPU box, repo root) -- the measurement table of DESIGN.md section 5
tag=$1; out=gpurun_out/sweep_$tag.jsonl; mkdir -p gpurun_out; : > $out
run() { echo "# $*" >> $out; timeout -k 10 200 python bench.py --no-cpu-baseline "$@" 2>/dev/null >> $out; }
run --env cr --batch 65536 --outputs full
run --env cr --batch 65536 --outputs min --no-step-api
run --env cr --batch 65536 --outputs none --no-step-api
run --env cr --batch 65536 --outputs full --traj soa --no-step-api
run --env cr --batch 1048576 --outputs full --no-step-api
run --env cr --batch 4194304 --mode graph --plan-steps 20 --steps 400 --warmup 40
run --env pg --batch 262144 --outputs min
run --env pg --batch 262144 --outputs full --no-step-api
run --env ra --batch 262144 --outputs min
run --env ra --batch 262144 --outputs full --no-step-api
run --env mixed --batch 1048576
python - "$out" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('#'): print(l.strip()); continue
    try: d = json.loads(l)
    except Exception: print("  (no json)"); continue
    r = d.get("roofline") or {}
    s = d.get("step_api") or {}
    print("  value %.3e  ms/step %.5f  frac %s  launch_us %s  parity %s | step_api %s launch_us %s frac %s" % (
        d["value"], d["ms_per_step"], r.get("frac"), r.get("launch_us"), (d.get("parity") or {}).get("state_bits_equal"),
        s.get("value"), s.get("launch_us"), s.get("frac_of_hbm_peak")))
PY
