# usage: bash profiles/r02/policy_ab.sh "<variants>"  -- closed-loop policy rollout (bench_policy.py) per library variant; tests on the last
export NIG_NO_AUTOBUILD=1
mkdir -p gpurun_out
cp neorl-industrial-gym_amd/libnig.so /tmp/libnig_orig.so
trap 'cp /tmp/libnig_orig.so neorl-industrial-gym_amd/libnig.so' EXIT   # (round-2 script, kept as the record of that A/B: it swaps libnig.so in place; new A/Bs load variants through NIG_LIB_PATH, profiles/ab.sh)
last=$(echo $1 | awk '{print $NF}')
cp neorl-industrial-gym_amd/libnig_$last.so neorl-industrial-gym_amd/libnig.so
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_abi_round2.py tests/test_spec_envs.py -m gpu -x -q -k "policy or pid or dataset or evaluate or agents" > gpurun_out/policy_tests.txt 2>&1 || { tail -30 gpurun_out/policy_tests.txt; cp /tmp/libnig_orig.so neorl-industrial-gym_amd/libnig.so; exit 1; }
tail -2 gpurun_out/policy_tests.txt
for r in 1 2; do for v in $1; do cp neorl-industrial-gym_amd/libnig_$v.so neorl-industrial-gym_amd/libnig.so
  for b in 65536 262144; do echo -n "$v batch $b: "; timeout -k 10 200 python profiles/bench_policy.py --steps 100 --batch $b --env ChemicalReactor-v0 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(' '.join('%s %.3e' % (k, v['env_steps_per_s']) for k, v in d.items() if k.startswith('affine') or k.startswith('pid')))"; done
done; done
cp /tmp/libnig_orig.so neorl-industrial-gym_amd/libnig.so
