#!/bin/bash
# usage: bash profiles/r03/mixed_ab.sh "<variant> ..."   (variants: base = libnig.so, else neorl-industrial-gym_amd/libnig_<variant>.so)
# Same-box A/B of the mixed launch (BASELINE configs[3], 1 048 576 lanes x 250 steps), reward + flags and with observation rows.
export NIG_NO_AUTOBUILD=1
for r in 1 2; do for v in $1; do
  if [ "$v" = base ]; then unset NIG_LIB_PATH; else export NIG_LIB_PATH=$PWD/neorl-industrial-gym_amd/libnig_$v.so; fi
  for o in min full; do echo -n "$v mixed $o: "
    timeout -k 10 150 python bench.py --env mixed --mixed-outputs $o --steps 12 --warmup 3 --settle 0.3 --no-cpu-baseline --no-step-api --no-parity --no-powergrid --no-brackets 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%.3e  launch_us %.1f  frac %.3f' % (d['value'], d['roofline']['launch_us'], d['roofline']['frac']))"
  done
done; done
unset NIG_LIB_PATH
