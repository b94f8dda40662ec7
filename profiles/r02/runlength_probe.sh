# per-launch time of the headline rollout as a function of how long it has been running back to back
# (--settle 0: no untimed load before the warm-up launches), then the driver's command with the default settle
export NIG_NO_AUTOBUILD=1
mkdir -p gpurun_out
{
for sb in 256 0; do for o in full none; do for k in 20 200 4000; do
  NIG_SPLIT_BLOCKS=$sb timeout -k 10 200 python bench.py --env cr --batch 65536 --outputs $o --steps $k --warmup 5 --settle 0 --no-cpu-baseline --no-parity --no-step-api --no-powergrid 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('split<=$sb outputs $o steps $k settle 0: launch_us %.1f frac %.3f' % (d['roofline']['launch_us'], d['roofline']['frac']))"
done; done; done
for sb in 256 0; do
NIG_SPLIT_BLOCKS=$sb python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('split<=$sb DRIVER COMMAND (default settle): value %.3e launch_us %.1f frac %.3f | pg launch_us %.1f frac %.3f' % (d['value'], d['roofline']['launch_us'], d['roofline']['frac'], d['powergrid']['roofline']['launch_us'], d['powergrid']['roofline']['frac']))"
done
} | tee gpurun_out/r02_runlength_probe2.txt
