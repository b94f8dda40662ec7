mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02_gpu_tests_3.log 2>&1; tail -15 gpurun_out/r02_gpu_tests_3.log
bash profiles/ab.sh "base coop1" "pg 262144 full aos" "pg 262144 min aos" "pg 1048576 full aos" 2>&1 | tee gpurun_out/r02_ab_pg1.log
