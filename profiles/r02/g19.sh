mkdir -p gpurun_out
export NIG_NO_AUTOBUILD=1
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gpu_tests_19.log 2>&1; tail -6 gpurun_out/r02_gpu_tests_19.log
for v in cur mlplds cur mlplds; do cp neorl-industrial-gym_amd/libnig_$v.so neorl-industrial-gym_amd/libnig.so; echo -n "$v: "; timeout -k 10 120 python profiles/bench_policy.py --steps 100 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print({k: ('%.3e' % v['env_steps_per_s'], round(v.get('actor_TFLOPs', 0), 1)) for k, v in d.items()})"; done
cp neorl-industrial-gym_amd/libnig_mlplds.so neorl-industrial-gym_amd/libnig.so
timeout -k 10 120 python profiles/bench_policy.py --steps 60 --env PowerGrid-v0 --batch 65536 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('pg', {k: ('%.3e' % v['env_steps_per_s'], round(v.get('actor_TFLOPs', 0), 1)) for k, v in d.items()})"
