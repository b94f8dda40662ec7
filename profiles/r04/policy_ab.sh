# closed loop at 65 536 lanes (GPU box, repo root): PowerGrid's paired form (round 4) against the one-wave kernel on the
# same box -- NIG_SPLIT_BLOCKS=0 keeps every batch on rollout_policy_kernel -- and the other two envs for the record
export NIG_NO_AUTOBUILD=1
for env in PowerGrid-v0 ChemicalReactor-v0 RobotAssembly-v0; do
  for sb in 256 0; do
    echo "== $env NIG_SPLIT_BLOCKS=$sb"
    NIG_SPLIT_BLOCKS=$sb timeout -k 10 150 python3 profiles/bench_policy.py --env $env --steps 200 --mlp-steps 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k: round(v['us_per_step'],3) for k,v in d.items() if k.startswith('affine')})"
  done
done
