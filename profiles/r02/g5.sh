mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02_gpu_tests_5.log 2>&1; tail -5 gpurun_out/r02_gpu_tests_5.log
bash profiles/ab.sh "coop1 coop2" "pg 262144 full aos" "cr 65536 full aos" "cr 262144 full aos" "ra 262144 full aos" 2>&1 | tee gpurun_out/r02_ab_pg2.log
