#!/bin/bash
# usage: bash profiles/ab.sh "<variant> <variant> ..." "<env batch outputs traj [extra bench args]>" ...
# Same-box A/B of kernel library builds (neorl-industrial-gym_amd/libnig_<variant>.so, built by hand):
# box-to-box spread of the fused rollout is ~15 %, larger than most single optimisations.
export NIG_NO_AUTOBUILD=1
variants=$1; shift
cfgs=("$@")
# a variant is loaded through NIG_LIB_PATH: libnig.so itself is never overwritten ("base" = the built libnig.so)
for r in 1 2; do for v in $variants; do
  if [ "$v" = base ]; then unset NIG_LIB_PATH; else export NIG_LIB_PATH=$PWD/neorl-industrial-gym_amd/libnig_$v.so; fi
  for cfg in "${cfgs[@]}"; do read -r e b o t extra <<< "$cfg"; echo -n "$v $cfg: "
    timeout -k 10 100 python bench.py --env $e --batch $b --outputs $o --traj $t $extra --steps 40 --warmup 8 --settle 0.4 --no-cpu-baseline --no-step-api --no-parity --no-powergrid --no-mixed --no-robotassembly --no-brackets 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%.3e  launch_us %.1f  frac %.3f' % (d['value'], d['roofline']['launch_us'], d['roofline']['frac']))"
  done
done; done
unset NIG_LIB_PATH
