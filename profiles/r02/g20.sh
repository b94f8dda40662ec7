mkdir -p gpurun_out
export NIG_NO_AUTOBUILD=1
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gpu_tests_20.log 2>&1; tail -6 gpurun_out/r02_gpu_tests_20.log
for v in cur polcoop; do cp neorl-industrial-gym_amd/libnig_$v.so neorl-industrial-gym_amd/libnig.so; for e in PowerGrid-v0 RobotAssembly-v0 ChemicalReactor-v0; do echo -n "$v $e: "; timeout -k 10 120 python profiles/bench_policy.py --steps 60 --env $e --batch 262144 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print({k: '%.3e' % v['env_steps_per_s'] for k, v in d.items() if k.startswith('affine')})"; done; done
cp neorl-industrial-gym_amd/libnig_polcoop.so neorl-industrial-gym_amd/libnig.so
