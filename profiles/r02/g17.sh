mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rollout or fused or full_size" > gpurun_out/r02_gpu_tests_17.log 2>&1; tail -4 gpurun_out/r02_gpu_tests_17.log
bash profiles/ab.sh "cur trearly" "cr 65536 full aos" "cr 1048576 full aos" "pg 262144 full aos" 2>&1 | tee gpurun_out/r02_ab_tr.log
