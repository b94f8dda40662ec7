mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02_gpu_tests_2.log 2>&1; tail -15 gpurun_out/r02_gpu_tests_2.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench_driver.json 2> gpurun_out/r02_bench_driver.err; tail -c 3000 gpurun_out/r02_bench_driver.json; tail -3 gpurun_out/r02_bench_driver.err
