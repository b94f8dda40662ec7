# integrator + helper wave form of the ChemicalReactor rollout (nig_split.hpp) against the one-wave form, same box
# usage: bash profiles/r02/split_probe.sh [quick]    (quick: no tests first)
mkdir -p gpurun_out
export NIG_NO_AUTOBUILD=1
if [ "$1" != quick ]; then
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_abi_round2.py tests/test_spec_envs.py -m gpu -x -q > gpurun_out/split_tests.txt 2>&1 || { tail -30 gpurun_out/split_tests.txt; exit 1; }
tail -3 gpurun_out/split_tests.txt
fi
for rep in 1 2; do for sb in 0 256; do for cfg in "65536 full" "65536 none" "65536 min" "32768 full" "16384 full"; do read -r b o <<< "$cfg"
  NIG_SPLIT_BLOCKS=$sb timeout -k 10 120 python bench.py --env cr --batch $b --outputs $o --steps 60 --warmup 10 --no-cpu-baseline --no-parity --no-step-api --no-powergrid 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('split<=$sb cr $b $o: launch_us %.1f  value %.3e frac %.3f' % (d['roofline']['launch_us'], d['value'], d['roofline']['frac']))"
done; done; done | tee gpurun_out/r02_split_probe.txt
