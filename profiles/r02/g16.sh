mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02_gpu_tests_16.log 2>&1; tail -6 gpurun_out/r02_gpu_tests_16.log
bash profiles/ab.sh "racoop crcoop" "cr 65536 full aos" "cr 65536 min aos" "cr 262144 full aos" "cr 65536 full aos --mode graph" "cr 4194304 full aos --mode graph --plan-steps 20" 2>&1 | tee gpurun_out/r02_ab_cr.log
