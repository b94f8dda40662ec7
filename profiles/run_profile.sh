#!/bin/bash
# usage: profiles/run_profile.sh <tag> [bench args...]   (run on the GPU box from the repo root)
# kernel-trace + stats only (PMC counters are collected in separate runs, see profiles/run_pmc.sh)
set -e
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
# build first: the profiled process must never spawn the compiler (NIG_NO_AUTOBUILD makes a stale library an error)
python3 -c 'import __graft_entry__ as g; g.build(force=False)' > /dev/null
export NIG_NO_AUTOBUILD=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 bench.py --no-cpu-baseline --no-parity $NIG_PROFILE_EXTRA "$@" > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
f=$(find $out -name "*kernel_stats.csv" | head -1)
[ "$f" -ef "$out/${tag}_kernel_stats.csv" ] || cp "$f" $out/${tag}_kernel_stats.csv
cat $out/${tag}_kernel_stats.csv | head -20
cat $out/bench.json
