mkdir -p gpurun_out
python -m pytest tests/test_gpu_mixed.py tests/test_spec_envs.py -m gpu -x -q > gpurun_out/r02_gpu_tests_10.log 2>&1; tail -25 gpurun_out/r02_gpu_tests_10.log
for l in fused streams; do python bench.py --env mixed --mixed-launch $l --steps 12 --warmup 2 2>gpurun_out/r02_mixed_$l.err | tee gpurun_out/r02_mixed_$l.json | python -c "import json,sys; d=json.load(sys.stdin); print('$l', '%.3e' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'])"; done
