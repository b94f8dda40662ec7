# round-5 GPU session 16: row-major action rings -- the new tests, then PowerGrid's wide form with rows vs row-major slots (native reads)
mkdir -p gpurun_out
export TMPDIR=/tmp NIG_NO_AUTOBUILD=1
timeout -k 10 900 python -m pytest -x -q -m gpu tests/test_gpu_action_layout.py tests/test_gpu_round3.py tests/test_gpu_noise_rollout.py > gpurun_out/r05_s16_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r05_s16_tests.log
run() { echo -n "$*: "; timeout -k 10 150 python bench.py --env pg --batch 262144 "$@" --steps 30 --warmup 6 --settle 0.4 --no-cpu-baseline --no-step-api --no-parity --no-brackets 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('launch_us %.1f  frac %.3f' % (d['roofline']['launch_us'], d['roofline']['frac']))"; }
for rep in 1 2 3; do for o in full min; do for l in rows aos; do run --outputs $o --ring-layout $l --ring 34; done; done; done > gpurun_out/r05_s16_pg_ring_layout.txt 2>&1
cat gpurun_out/r05_s16_pg_ring_layout.txt
