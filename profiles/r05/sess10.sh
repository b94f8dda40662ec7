# round-5 GPU session 10: RobotAssembly 262 144 lanes -- the last residency round in the three-wave form (default) against everything one-wave
# (NIG_SPLIT_BLOCKS=0), by output mode; the new bit-identity test
mkdir -p gpurun_out
export TMPDIR=/tmp NIG_NO_AUTOBUILD=1
timeout -k 10 600 python -m pytest -x -q -m gpu tests/test_gpu_split.py -k "robot_assembly" > gpurun_out/r05_s10_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r05_s10_tests.log
run() { v=$1; shift; echo -n "split_blocks=$v $*: "; NIG_SPLIT_BLOCKS=$v timeout -k 10 150 python bench.py --env ra "$@" --steps 12 --warmup 3 --settle 0.3 --no-cpu-baseline --no-step-api --no-parity --no-brackets --no-single-env 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%.3e  launch_us %.1f  frac %.3f  %s' % (d['value'], d['roofline']['launch_us'], d['roofline']['frac'], d['roofline']['kernel']))"; }
for rep in 1 2; do for o in full min none; do for v in 256 0; do run $v --batch 262144 --outputs $o; done; done; done > gpurun_out/r05_s10_ra_tail.txt 2>&1
for v in 256 0; do run $v --batch 458752 --outputs full; done >> gpurun_out/r05_s10_ra_tail.txt 2>&1
cat gpurun_out/r05_s10_ra_tail.txt
