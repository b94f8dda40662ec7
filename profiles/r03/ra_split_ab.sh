#!/bin/bash
# RobotAssembly: three-wave form (default at <= one 256-lane block per CU) against the one-wave kernel (NIG_SPLIT_BLOCKS=0),
# same box; and the three-wave form in rounds at 262 144 lanes (variant built with -DNIG_RA_SPLIT_ROUNDS=true).
export NIG_NO_AUTOBUILD=1
run() { echo -n "$1 [$2 lanes, $3]: "; timeout -k 10 100 python bench.py --env ra --batch $2 --outputs $3 --steps 40 --warmup 8 --settle 0.4 --no-cpu-baseline --no-step-api --no-parity --no-powergrid --no-mixed --no-brackets 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%.3e  launch_us %.1f  frac %.3f  %s' % (d['value'], d['roofline']['launch_us'], d['roofline']['frac'], d['roofline']['kernel']))"; }
for r in 1 2; do
  for b in 16384 32768 65536; do for o in full min; do
    unset NIG_SPLIT_BLOCKS; run three-wave $b $o
    export NIG_SPLIT_BLOCKS=0; run one-wave $b $o
  done; done
  unset NIG_SPLIT_BLOCKS
  run one-wave 262144 full
  run one-wave 262144 min
  NIG_SPLIT_BLOCKS=0 run one-wave 131072 full
  if [ -f $PWD/neorl-industrial-gym_amd/libnig_rarounds.so ]; then
    NIG_LIB_PATH=$PWD/neorl-industrial-gym_amd/libnig_rarounds.so run rounds 262144 full
    NIG_LIB_PATH=$PWD/neorl-industrial-gym_amd/libnig_rarounds.so run rounds 131072 full
  fi
done
