#!/bin/bash
# usage: bash profiles/ab.sh "<variant> <variant> ..." "<env batch outputs traj [extra bench args]>" ...
# Same-box A/B of kernel library builds (neorl-industrial-gym_amd/libnig_<variant>.so, built by hand):
# box-to-box spread of the fused rollout is ~15 %, larger than most single optimisations.
export NIG_NO_AUTOBUILD=1
variants=$1; shift
cfgs=("$@")
cp neorl-industrial-gym_amd/libnig.so /tmp/libnig_keep.so
for r in 1 2; do for v in $variants; do
  cp neorl-industrial-gym_amd/libnig_$v.so neorl-industrial-gym_amd/libnig.so
  for cfg in "${cfgs[@]}"; do read -r e b o t extra <<< "$cfg"; echo -n "$v $cfg: "
    timeout -k 10 100 python bench.py --env $e --batch $b --outputs $o --traj $t $extra --steps 40 --warmup 8 --settle 0.4 --no-cpu-baseline --no-step-api --no-parity --no-powergrid 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%.3e  launch_us %.1f  frac %.3f' % (d['value'], d['roofline']['launch_us'], d['roofline']['frac']))"
  done
done; done
cp /tmp/libnig_keep.so neorl-industrial-gym_amd/libnig.so
