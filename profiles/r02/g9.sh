mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02_gpu_tests_9.log 2>&1; tail -5 gpurun_out/r02_gpu_tests_9.log
bash profiles/ab.sh "coop3 coop5" "pg 262144 full aos" "pg 262144 min aos" "pg 262144 full aos --mode graph --plan-steps 50" "pg 1048576 full aos" 2>&1 | tee gpurun_out/r02_ab_pg5.log
