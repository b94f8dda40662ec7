# round-5 GPU session 2: PowerGrid with TWO action register sets (the wait for an action lands behind stores that are two steps old)
# against the one-set build of session 1 (libnig_oneset.so) and the no-load diagnostic; the 4x4x1 MFMA head; per-CU clock stamps
mkdir -p gpurun_out
export TMPDIR=/tmp NIG_NO_AUTOBUILD=1
run() { echo "== $*" >&2; timeout -k 10 "$@"; rc=$?; echo "   rc=$rc" >&2; return $rc; }
run 900 python -m pytest -x -q -m gpu tests/test_gpu_parity.py -k "mlp" > gpurun_out/r05_s2_mlp_tests.log 2>&1; echo "mlp tests rc=$?"; tail -3 gpurun_out/r05_s2_mlp_tests.log
run 900 python -m pytest -x -q -m gpu tests/test_gpu_round3.py tests/test_gpu_noise_rollout.py tests/test_gpu_mixed.py -k "pg or PowerGrid or power or wide or pair or mixed" > gpurun_out/r05_s2_pg_tests.log 2>&1; echo "pg tests rc=$?"; tail -3 gpurun_out/r05_s2_pg_tests.log
bash profiles/ab.sh "base oneset noact" "pg 262144 full aos" "pg 262144 min aos" "pg 262144 none aos" "pg 131072 full aos" > gpurun_out/r05_s2_pg_ab.txt 2>&1
cat gpurun_out/r05_s2_pg_ab.txt
for v in base oneset; do
  if [ "$v" = base ]; then unset NIG_LIB_PATH; else export NIG_LIB_PATH=$PWD/neorl-industrial-gym_amd/libnig_$v.so; fi
  echo "== $v policy bench" ; run 300 python profiles/bench_policy.py > gpurun_out/r05_s2_policy_$v.txt 2>&1; tail -12 gpurun_out/r05_s2_policy_$v.txt
done
unset NIG_LIB_PATH
run 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_s2_bench.json 2> gpurun_out/r05_s2_bench.err
echo "bench rc=$?"; python - <<'PY'
import json
t=open('gpurun_out/r05_s2_bench.json').read()
d=json.loads(t[t.find('{"metric"'):].splitlines()[0])
print('headline', d['ms_per_step'], d['roofline']['frac'], d['rank_times'].get('clock'))
print('pg', d['powergrid']['ms_per_step'], d['powergrid']['roofline']['frac'], d['powergrid']['rank_times'].get('clock'))
print('ra', d['robotassembly']['ms_per_step'], d['robotassembly']['roofline']['frac'], d['robotassembly']['rank_times'].get('clock'))
print('mixed', d['mixed']['ms_per_step'], d['mixed']['roofline']['frac'])
PY
