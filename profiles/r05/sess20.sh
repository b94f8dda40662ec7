# round-5 GPU session 20: the row-major ring's copy path after hardening (only the slots a call reads, grid y capped, argument
# checks before any launch) + the ABI tests
set -o pipefail
timeout -k 10 900 python -m pytest -x -q -m gpu tests/test_gpu_action_layout.py tests/test_gpu_abi_round2.py tests/test_abi.py > gpurun_out/r05_s20_tests.log 2>&1; rc=$?; tail -4 gpurun_out/r05_s20_tests.log; exit $rc
