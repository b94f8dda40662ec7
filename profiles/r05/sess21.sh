# round-5 GPU session 21: PowerGrid's restart passes pooled over the block (two LDS-only barriers per step) -- bit-identity first
# (every PowerGrid rollout test + the mixed launch), then the same-box A/B against the wave-private passes (libnig_nopool.so)
set -o pipefail
run() { t=$1; shift; timeout -k 10 $t "$@"; }
run 900 python -m pytest -x -q -m gpu tests/test_gpu_round3.py tests/test_gpu_mixed.py tests/test_gpu_action_layout.py tests/test_gpu_noise_rollout.py tests/test_gpu_reference_stats.py -k "pg or PowerGrid or powergrid or mixed or wide or lds" > gpurun_out/r05_s21_tests.log 2>&1; rc=$?; tail -4 gpurun_out/r05_s21_tests.log; [ $rc -eq 0 ] || exit $rc
bash profiles/ab.sh "nopool base" "pg 262144 full aos --ring-layout aos" "pg 262144 min aos --ring-layout aos" "pg 262144 none aos --ring-layout aos" "pg 65536 full aos" > gpurun_out/r05_s21_pool_ab.txt 2>&1
cat gpurun_out/r05_s21_pool_ab.txt
