# three-wave form run in several rounds of one block per CU (NIG_SPLIT_BLOCKS raised) against the one-wave form at larger batches
export NIG_NO_AUTOBUILD=1
mkdir -p gpurun_out
for b in 98304 131072 196608 262144 327680 524288 1048576; do for sb in 256 100000; do
  NIG_SPLIT_BLOCKS=$sb timeout -k 10 200 python bench.py --env cr --batch $b --outputs full --steps 30 --warmup 5 --settle 0.4 --no-cpu-baseline --no-parity --no-step-api --no-powergrid 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('batch $b split<=$sb: launch_us %.1f frac %.3f value %.3e' % (d['roofline']['launch_us'], d['roofline']['frac'], d['value']))"
done; done | tee gpurun_out/r02_rounds_probe.txt
