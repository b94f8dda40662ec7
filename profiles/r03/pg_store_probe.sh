#!/bin/bash
# Where do PowerGrid's 0.6 ms of "full outputs" go?  Same kernel, same box: (a) full row-major outputs, (b) the same stores
# issued to ONE reused set of rows (NIG_DIAG_OVERWRITE: they stay in cache -- issue cost without the HBM drain),
# (c) the [T][S][ld] row layout, (d) no outputs.
set -e
out=${1:-gpurun_out/r03_pg_store_probe.txt}
export NIG_NO_AUTOBUILD=1
: > $out
run() { echo "== $1" >> $out; shift; "$@" 2>>$out.err | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print(json.dumps({'launch_us': r['roofline']['launch_us'], 'frac': r['roofline']['frac'], 'kernel': r['roofline']['kernel']}))" >> $out; }
B="python bench.py --env pg --steps 40 --warmup 5 --no-step-api --no-cpu-baseline --no-parity --no-brackets"
run "full aos" $B --outputs full
NIG_DIAG_OVERWRITE=1 run "full aos, overwrite one row set (stores stay in cache)" $B --outputs full
run "full soa" $B --outputs full --traj soa
run "min" $B --outputs min
run "none" $B --outputs none
cat $out
