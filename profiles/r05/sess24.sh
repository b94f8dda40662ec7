# round-5 GPU session 24: the single-env class (configs[0]) with the host polling the stream (default) against the blocking wait
set -o pipefail
export NIG_NO_AUTOBUILD=1
for r in 1 2; do
  NIG_HOST_SPIN_US=0 timeout -k 10 120 python profiles/tools/single_env_rate.py 2>/dev/null
  timeout -k 10 120 python profiles/tools/single_env_rate.py 2>/dev/null
done > gpurun_out/r05_s24_single_env.txt 2>&1
cat gpurun_out/r05_s24_single_env.txt
timeout -k 10 600 python -m pytest -x -q -m gpu tests/test_gpu_parity.py -k "single or make or host" > gpurun_out/r05_s24_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r05_s24_tests.log
