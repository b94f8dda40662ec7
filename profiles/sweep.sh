#!/bin/bash
# usage: bash profiles/sweep.sh <tag>     (GPU box, repo root) -- the measurement table of DESIGN.md section 5
# one bench step = one launch of 250 env.step per lane (bench.py); NIG_NO_AUTOBUILD: build first
tag=$1; out=gpurun_out/sweep_$tag.jsonl; mkdir -p gpurun_out; : > $out
python3 -c 'import __graft_entry__ as g; g.build(force=False)' > /dev/null
export NIG_NO_AUTOBUILD=1
run() { echo "# $*" >> $out; timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity --no-powergrid --no-mixed --no-robotassembly --no-brackets --no-single-env "$@" 2>/dev/null >> $out; }
run --env cr --batch 65536 --outputs full --steps 60 --warmup 10
run --env cr --batch 65536 --outputs min --no-step-api --steps 60 --warmup 10
run --env cr --batch 65536 --outputs none --no-step-api --steps 60 --warmup 10
run --env cr --batch 65536 --outputs full --traj soa --no-step-api --steps 60 --warmup 10
run --env cr --batch 262144 --outputs full --no-step-api --steps 40 --warmup 8
run --env cr --batch 1048576 --outputs full --no-step-api --steps 20 --warmup 4
run --env cr --batch 4194304 --mode graph --plan-steps 20 --steps 20 --warmup 4
NIG_SPLIT_BLOCKS=0 run --env cr --batch 65536 --outputs full --no-step-api --steps 60 --warmup 10      # one-wave form
run --env pg --batch 262144 --outputs min --steps 12 --warmup 3
NIG_WIDE_MIN_BLOCKS=1000000000 run --env pg --batch 262144 --outputs full --no-step-api --steps 12 --warmup 3      # register-resident form
run --env pg --batch 262144 --outputs none --no-step-api --steps 12 --warmup 3
run --env pg --batch 262144 --outputs full --no-step-api --steps 12 --warmup 3
run --env pg --batch 1048576 --outputs full --no-step-api --steps 6 --warmup 2
run --env ra --batch 262144 --outputs min --steps 12 --warmup 3
run --env ra --batch 262144 --outputs full --no-step-api --steps 12 --warmup 3
run --env mixed --batch 1048576 --mixed-outputs min --steps 10 --warmup 2
run --env mixed --batch 1048576 --mixed-outputs full --steps 10 --warmup 2
run --env mixed --batch 1048576 --mixed-outputs min --mixed-launch streams --steps 10 --warmup 2
run --env mixed --batch 1048576 --mixed-outputs min --mixed-set survey --steps 10 --warmup 2
python - "$out" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('#'): print(l.strip()); continue
    try: d = json.loads(l)
    except Exception: print("  (no json)"); continue
    r = d.get("roofline") or {}
    s = d.get("step_api") or {}
    print("  value %.3e  ms/step %.4f  frac %s  launch_us %s | step_api %s launch_us %s frac %s" % (
        d["value"], d["ms_per_step"], "%.3f (alg %.3f)" % (r.get("frac", 0), r.get("frac_algorithmic", r.get("frac", 0))), r.get("launch_us"),
        s.get("value"), s.get("launch_us"), s.get("frac_of_hbm_peak")))
PY
