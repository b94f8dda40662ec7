# round-5 GPU session 13: how much does PowerGrid's full-output launch vary from PROCESS to process on one box (same build, same inputs)?
mkdir -p gpurun_out
export TMPDIR=/tmp NIG_NO_AUTOBUILD=1
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout -k 10 100 python bench.py --env pg --batch 262144 --outputs full --steps 30 --warmup 6 --settle 0.4 --no-cpu-baseline --no-step-api --no-parity --no-brackets 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']; c=d['rank_times'].get('clock',{}); p=d['rank_times'].get('dpm',{})
print('run $i  launch_us %.1f  frac %.3f  clock %.0f MHz  sclk %s  power %s W' % (r['launch_us'], r['frac'], c.get('shader_clock_mhz',0), p.get('sclk_mhz'), p.get('power_w')))"
done > gpurun_out/r05_s13_pg_process_spread.txt 2>&1
cat gpurun_out/r05_s13_pg_process_spread.txt
