#!/bin/bash
# PowerGrid, reward + flags: the wide form (512-lane blocks, 2 per CU = 4 waves per SIMD) against the 256-lane LDS form
# (3 blocks per CU = 3 waves per SIMD) over batch sizes -- whole rounds of either form and the sizes between
export NIG_NO_AUTOBUILD=1
run() { echo -n "$1 lanes $2 outputs $3: "; NIG_WIDE_MIN_BLOCKS=$4 timeout -k 10 150 python bench.py --env pg --batch $2 --outputs $3 --steps 16 --warmup 4 --settle 0.3 --no-cpu-baseline --no-step-api --no-parity --no-powergrid --no-mixed --no-brackets --no-single-env 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%.3e  launch_us %.1f  kernel %s' % (d['value'], d['roofline']['launch_us'], d['roofline'].get('kernel')))"; }
for o in min full; do
for b in 196608 262144 393216 524288 786432 1048576; do
  run wide512 $b $o 1
  run lds256  $b $o 100000000
done; done
