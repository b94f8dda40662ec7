#!/bin/bash
# usage: bash profiles/run_sq2.sh <tag> [bench args...]   -- LDS-side SQ counters (own pass, no tracing)
set -e
tag=$1; shift
out=gpurun_out/sq2_$tag
mkdir -p $out
export TMPDIR=/tmp
python3 -c 'import __graft_entry__ as g; g.build(force=False)' > /dev/null
export NIG_NO_AUTOBUILD=1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM --output-format csv -d $out -o $tag -- python3 bench.py --no-cpu-baseline --no-parity --no-step-api "$@" > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if any(x in r['Kernel_Name'] for x in ('rollout_kernel', 'split_rollout', 'step_kernel')):
        acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        v = sorted(v); print("   %-22s median %.4g  (n=%d)" % (c, v[len(v)//2], len(v)))
PY
