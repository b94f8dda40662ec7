#!/bin/bash
# usage: bash profiles/ab.sh "<variant> <variant> ..." "<env batch outputs traj>" ...
# Same-box A/B of kernel library builds (neorl-industrial-gym_amd/libnig_<variant>.so, built by hand):
# box-to-box spread of the fused rollout is ~15 %, larger than most single optimisations.
variants=$1; shift
cfgs=("$@")
for r in 1 2; do for v in $variants; do
  cp neorl-industrial-gym_amd/libnig_$v.so neorl-industrial-gym_amd/libnig.so; touch neorl-industrial-gym_amd/libnig.so
  for cfg in "${cfgs[@]}"; do read -r e b o t <<< "$cfg"; echo -n "$v $cfg: "
    timeout -k 10 100 python bench.py --env $e --batch $b --outputs $o --traj $t --no-cpu-baseline --no-step-api --no-parity 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%.3e' % d['value'])"
  done
done; done
