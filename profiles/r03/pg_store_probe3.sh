#!/bin/bash
# PowerGrid's 0.45 ms of trajectory stores: issue or HBM drain?  The round's first probe (pg_store_probe.sh) overwrote ONE
# row set with the kernel's STREAMING stores; profiles/ubench/store_addr.hip later showed that streaming stores write
# through once the footprint exceeds L2, so that probe could not tell the two apart.  Here: ordinary (write-back) stores,
# (a) into distinct rows (all bytes must reach HBM), (b) into one reused row set of 35 MB (may stay in L2 / Infinity Cache),
# against the streaming-store build, same box.  Variant plainst = env_pg built with -DNIG_DIAG_PG_PLAINSTORE.
export NIG_NO_AUTOBUILD=1
B="python bench.py --env pg --steps 40 --warmup 5 --settle 0.4 --no-step-api --no-cpu-baseline --no-parity --no-brackets --no-mixed --outputs full"
show() { python -c "import json,sys; d=json.load(sys.stdin); print('launch_us %.1f  frac %.3f' % (d['roofline']['launch_us'], d['roofline']['frac']))"; }
for r in 1 2; do
  echo -n "streaming stores, distinct rows:  "; $B 2>/dev/null | show
  echo -n "streaming stores, one row set:    "; NIG_DIAG_OVERWRITE=1 $B 2>/dev/null | show
  echo -n "ordinary stores, distinct rows:   "; NIG_LIB_PATH=$PWD/neorl-industrial-gym_amd/libnig_plainst.so $B 2>/dev/null | show
  echo -n "ordinary stores, one row set:     "; NIG_DIAG_OVERWRITE=1 NIG_LIB_PATH=$PWD/neorl-industrial-gym_amd/libnig_plainst.so $B 2>/dev/null | show
  echo -n "no outputs:                       "; python bench.py --env pg --steps 40 --warmup 5 --settle 0.4 --no-step-api --no-cpu-baseline --no-parity --no-brackets --no-mixed --outputs none 2>/dev/null | show
done
