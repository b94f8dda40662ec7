#!/bin/bash
# A/B of PowerGrid's rollout forms on ONE box: register-resident rollout_kernel (two waves per SIMD) vs the LDS-resident
# wide form (nig_pg_lds.hpp, four waves per SIMD).  Same library, the form is chosen by NIG_WIDE_MIN_BLOCKS.
set -e
out=${1:-gpurun_out/r03_pg_ab.txt}
export NIG_NO_AUTOBUILD=1
: > $out
for outputs in full min none; do
  for form in register wide; do
    if [ $form = register ]; then export NIG_WIDE_MIN_BLOCKS=1000000000; else unset NIG_WIDE_MIN_BLOCKS; fi
    echo "== pg 262144 outputs=$outputs form=$form" >> $out
    python bench.py --env pg --steps 40 --warmup 5 --outputs $outputs --no-step-api --no-cpu-baseline --no-parity --no-brackets 2>>$out.err | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print(json.dumps({'value': r['value'], 'launch_us': r['roofline']['launch_us'], 'frac': r['roofline']['frac'], 'kernel': r['roofline']['kernel']}))" >> $out
  done
done
cat $out
