mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02_gpu_tests_15.log 2>&1; tail -6 gpurun_out/r02_gpu_tests_15.log
bash profiles/ab.sh "cur racoop racoop2" "ra 262144 full aos" "ra 262144 min aos" "ra 149760 min aos" "ra 262144 full aos --mode graph --plan-steps 50" 2>&1 | tee gpurun_out/r02_ab_ra.log
for v in cur racoop2; do cp neorl-industrial-gym_amd/libnig_$v.so neorl-industrial-gym_amd/libnig.so; NIG_NO_AUTOBUILD=1 python bench.py --env mixed --steps 12 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$v mixed', '%.3e' % d['value'], 'ms/step %.3f' % d['ms_per_step'])"; done
