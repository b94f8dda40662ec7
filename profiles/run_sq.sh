#!/bin/bash
# usage: bash profiles/run_sq.sh <tag> [bench args...]   -- SQ issue/stall counters (own pass, no tracing)
set -e
tag=$1; shift
out=gpurun_out/sq_$tag
mkdir -p $out
export TMPDIR=/tmp
# build first: the profiled process must never spawn the compiler (NIG_NO_AUTOBUILD makes a stale library an error)
python3 -c 'import __graft_entry__ as g; g.build(force=False)' > /dev/null
export NIG_NO_AUTOBUILD=1
# NIG_SQ_COUNTERS overrides the counter set of the pass (e.g. the LDS set: SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD)
ctrs=${NIG_SQ_COUNTERS:-SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY}
rocprofv3 --pmc $ctrs --output-format csv -d $out -o $tag -- python3 bench.py --no-cpu-baseline --no-parity --no-step-api --no-mixed --no-robotassembly --no-brackets --no-single-env --settle 0 "$@" > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if any(x in r['Kernel_Name'] for x in ('rollout_kernel', 'rollout_pair', 'rollout_pg_pair', 'step_kernel', 'split_rollout', 'rollout_wide', 'mixed_rollout', 'rollout_mlp', 'split_policy', 'rollout_policy')):
        acc[r['Kernel_Name'][:80]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        v = sorted(v); print("   %-22s median %.4g  (n=%d)" % (c, v[len(v)//2], len(v)))
PY
