#!/bin/bash
# PowerGrid, batches of at most one 256-lane block per compute unit: the paired form (stepping + producer wave per 64 lanes,
# rollout_pg_pair_kernel; the default there) against the one-wave 256-lane form of the same LDS-resident body
# (NIG_SPLIT_BLOCKS=0), same box; and the BASELINE size (262 144 lanes, wide form) before / after the body gained its
# PROD template parameter (variant prepair = the library built before that change), to show the wide form did not move.
export NIG_NO_AUTOBUILD=1
run() { echo -n "$1 [$2 lanes, $3]: "; timeout -k 10 100 python bench.py --env pg --batch $2 --outputs $3 --steps 40 --warmup 8 --settle 0.4 --no-cpu-baseline --no-step-api --no-parity --no-powergrid --no-mixed --no-brackets 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%.3e  launch_us %.1f  frac %.3f  %s' % (d['value'], d['roofline']['launch_us'], d['roofline']['frac'], d['roofline']['kernel']))"; }
for r in 1 2; do
  for b in 16384 32768 65536; do for o in full min; do
    unset NIG_SPLIT_BLOCKS; run paired $b $o
    export NIG_SPLIT_BLOCKS=0; run one-wave $b $o
  done; done
  unset NIG_SPLIT_BLOCKS
  run wide 262144 full
  if [ -f $PWD/neorl-industrial-gym_amd/libnig_prepair.so ]; then NIG_LIB_PATH=$PWD/neorl-industrial-gym_amd/libnig_prepair.so run wide-before 262144 full; fi
done
