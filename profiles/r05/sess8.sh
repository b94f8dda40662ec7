# round-5 GPU session 8: ChemicalReactor beyond one round -- three-wave form in rounds (default rule) against the one-wave kernel (NIG_SPLIT_BLOCKS=0),
# by output mode; the mixed-floor table showed the rounds losing badly with reward + flags outputs (5.33e6 vs 3.24e6 chip cycles at 1 M lanes)
mkdir -p gpurun_out
export TMPDIR=/tmp NIG_NO_AUTOBUILD=1
run() { v=$1; shift; echo -n "split_blocks=$v $*: "; NIG_SPLIT_BLOCKS=$v timeout -k 10 150 python bench.py --env cr "$@" --steps 12 --warmup 3 --settle 0.3 --no-cpu-baseline --no-step-api --no-parity --no-brackets --no-single-env 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%.3e  launch_us %.1f  %s' % (d['value'], d['roofline']['launch_us'], d['roofline']['kernel']))"; }
for B in 131072 262144 1048576; do for o in none min full; do for v in 256 0; do run $v --batch $B --outputs $o; done; done; done > gpurun_out/r05_s8_cr_rounds.txt 2>&1
for o in full; do for v in 256 0; do run $v --batch 1048576 --outputs $o --traj soa; done; done >> gpurun_out/r05_s8_cr_rounds.txt 2>&1
cat gpurun_out/r05_s8_cr_rounds.txt
