#!/bin/bash
# usage: bash profiles/run_pmc.sh <tag> [bench args...]      (GPU box, repo root)
# HBM traffic counters in their OWN passes (never combined with trace domains), as
# MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots" prescribe: FETCH_SIZE and WRITE_SIZE do not
# fit one pass.  --calibrate makes bench.py also run known-size dword-per-lane copies
# (copy_rows_kernel) so the byte scale of this access width can be calibrated.
set -e
tag=$1; shift
out=gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
# build first: the profiled process must never spawn the compiler (NIG_NO_AUTOBUILD makes a stale library an error)
python3 -c 'import __graft_entry__ as g; g.build(force=False)' > /dev/null
export NIG_NO_AUTOBUILD=1
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $out/$ctr -o $tag -- python3 bench.py --no-cpu-baseline --no-parity --no-step-api --no-mixed --no-robotassembly --no-brackets --no-single-env --settle 0 --calibrate "$@" > $out/bench_$ctr.json 2> $out/bench_$ctr.err || { tail -20 $out/bench_$ctr.err; exit 1; }
  f=$(find $out/$ctr -name "*counter_collection.csv" | head -1)
  cp $f $out/${tag}_${ctr}.csv
done
python3 profiles/pmc_to_traffic.py $out $tag "$@"
