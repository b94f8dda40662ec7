mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02_gpu_tests_12.log 2>&1; tail -40 gpurun_out/r02_gpu_tests_12.log
