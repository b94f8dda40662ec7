# round-5 GPU session 23: the headline launch against its length -- 125 / 250 / 500 / 1000 env.step per launch (ring 250):
# how much of the 165 us per 250 steps is the launch's fixed part (block start, table staging, first / last step)?
set -o pipefail
export NIG_NO_AUTOBUILD=1
for r in 1 2; do for P in 125 250 500 1000; do
  echo -n "cr 65536 full, $P steps per launch: "
  timeout -k 10 200 python bench.py --env cr --batch 65536 --outputs full --traj aos --plan-steps $P --ring 250 --steps 20 --warmup 5 --settle 0.6 --no-cpu-baseline --no-step-api --no-parity --no-powergrid --no-mixed --no-robotassembly --no-brackets --no-single-env 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('%.4e env-steps/s  launch_us %.1f  per 250 steps %.1f us  frac %.3f' % (d['value'], r['launch_us'], r['launch_us']*250/$P, r['frac']))"
done; done > gpurun_out/r05_s23_launch_length.txt 2>&1
cat gpurun_out/r05_s23_launch_length.txt
