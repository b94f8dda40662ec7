# round-5 GPU session 27: row-major action slots read natively by every LDS-resident PowerGrid form (wide 512 / 256, paired with
# the LDS stepper) -- layout tests, then rows vs row-major at 65 536 lanes (paired form) and 131 072 (wide 256-lane form)
set -o pipefail
export NIG_NO_AUTOBUILD=1
timeout -k 10 900 python -m pytest -x -q -m gpu tests/test_gpu_action_layout.py > gpurun_out/r05_s27_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r05_s27_tests.log; [ $rc -eq 0 ] || exit $rc
for r in 1 2; do for b in 65536 131072; do for lay in rows aos; do
  echo -n "pg $b full, ring $lay: "
  timeout -k 10 200 python bench.py --env pg --batch $b --outputs full --traj aos --ring-layout $lay --ring 64 --steps 12 --warmup 3 --settle 0.4 --no-cpu-baseline --no-step-api --no-parity --no-powergrid --no-mixed --no-robotassembly --no-brackets --no-single-env 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('%.3e env-steps/s  launch_us %.1f  frac %.3f  %s' % (d['value'], r['launch_us'], r['frac'], r['kernel']))"
done; done; done > gpurun_out/r05_s27_small_batch_ring.txt 2>&1
cat gpurun_out/r05_s27_small_batch_ring.txt
