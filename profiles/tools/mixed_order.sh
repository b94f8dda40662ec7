#!/bin/bash
# EXPERIMENT: launch order of the mixed kernel's segments (env_cost table through NIG_DIAG_MIXED_COST)
export NIG_NO_AUTOBUILD=1
run() { echo -n "$1: "; NIG_DIAG_MIXED_COST=$2 timeout -k 10 100 python bench.py --env mixed --batch 1048576 --mixed-outputs min --steps 30 --warmup 6 --settle 0.4 --no-cpu-baseline --no-step-api --no-parity --no-powergrid --no-mixed --no-brackets 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%.3e  launch_us %.1f' % (d['value'], d['roofline']['launch_us']))"; }
for r in 1 2; do
run "base RA,PG,Supply,Steel,HVAC,CR,Water" "10,44,50,15,20,14,9,17,29"
run "PG first                            " "10,60,50,15,20,14,9,17,29"
run "PG,RA then cheap ascending          " "40,60,50,15,20,30,45,25,20"
run "cheap first (reverse)               " "50,5,4,15,20,40,60,30,10"
run "PG, plants..., RA last              " "10,60,1,15,20,14,9,17,29"
done
