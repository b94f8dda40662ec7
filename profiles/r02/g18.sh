mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02_gpu_tests_18.log 2>&1; tail -5 gpurun_out/r02_gpu_tests_18.log
bash profiles/ab.sh "cur pg512" "pg 262144 full aos --mode graph --plan-steps 50" "pg 1048576 full aos --mode graph --plan-steps 20" "pg 65536 full aos --mode graph --plan-steps 50" 2>&1 | tee gpurun_out/r02_ab_pg512.log
