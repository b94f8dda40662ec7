# round-5 GPU session 19: RobotAssembly's launch against the batch size -- 262 144 lanes are 4 096 waves on 3 072 wave slots
# (three waves per SIMD at 168 registers): 1.33 rounds.  How much of the 0.59-0.61 is the second round's tail?
set -o pipefail
export NIG_NO_AUTOBUILD=1
for r in 1 2; do for b in 196608 262144 393216 589824; do
  echo -n "ra $b full: "
  timeout -k 10 200 python bench.py --env ra --batch $b --outputs full --traj aos --steps 12 --warmup 3 --settle 0.4 --no-cpu-baseline --no-step-api --no-parity --no-powergrid --no-mixed --no-robotassembly --no-brackets --no-single-env 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%.3e env-steps/s  launch_us %.1f  frac %.3f' % (d['value'], d['roofline']['launch_us'], d['roofline']['frac']))"
done; done > gpurun_out/r05_s19_ra_batch.txt 2>&1
cat gpurun_out/r05_s19_ra_batch.txt
