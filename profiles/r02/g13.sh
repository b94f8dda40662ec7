mkdir -p gpurun_out
python -m pytest tests/test_gpu_abi_round2.py -m gpu -x -q > gpurun_out/r02_gpu_tests_13.log 2>&1; tail -15 gpurun_out/r02_gpu_tests_13.log
for cfg in "20 5" "20 50" "200 5" "20 5"; do set -- $cfg; python bench.py --gpus 1 --steps $1 --warmup $2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('steps $1 warmup $2: value %.3e launch_us %.1f frac %.3f pg_us %.1f' % (d['value'], d['roofline']['launch_us'], d['roofline']['frac'], d['powergrid']['roofline']['launch_us']))"; done
