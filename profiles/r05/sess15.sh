# round-5 GPU session 15: bench-related GPU tests after the ring default, then the profile session's part A again (driver line + phases) on the final build
mkdir -p gpurun_out
export TMPDIR=/tmp NIG_NO_AUTOBUILD=1
timeout -k 10 900 python -m pytest -x -q -m gpu tests/test_gpu_rccl_one_rank.py tests/test_gpu_round3.py -k "bench or rccl" > gpurun_out/r05_s15_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r05_s15_tests.log
bash profiles/r05/prof_final.sh A 2>&1 | tail -12
