# round-5 GPU session 1 (GPU box, repo root): the new tests, the driver line with its new records, PowerGrid A/B of
# (a) the one-instruction-shorter generator index (base vs probit_r04), (b) a conflict-free table gather (noconf: upper bound of the
# LDS bank-conflict lead), (c) no global load in the loop (noact: what the in-order vmcnt wait is worth)
mkdir -p gpurun_out
export TMPDIR=/tmp NIG_NO_AUTOBUILD=1
run() { echo "== $*" >&2; timeout -k 10 "$@"; rc=$?; echo "   rc=$rc" >&2; return $rc; }
run 900 python -m pytest -x -q -m gpu tests/test_gpu_reference_stats.py tests/test_gpu_rccl_one_rank.py tests/test_gpu_abi_round2.py -s > gpurun_out/r05_s1_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r05_s1_tests.log
run 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_s1_bench.json 2> gpurun_out/r05_s1_bench.err
echo "bench rc=$?"; tail -3 gpurun_out/r05_s1_bench.err
bash profiles/ab.sh "base probit_r04 noconf noact" "pg 262144 full aos" "pg 262144 min aos" "pg 262144 none aos" > gpurun_out/r05_s1_pg_ab.txt 2>&1
cat gpurun_out/r05_s1_pg_ab.txt
bash profiles/ab.sh "base probit_r04" "cr 65536 full aos" "mixed 1048576 full aos --mixed-outputs min" > gpurun_out/r05_s1_cr_ab.txt 2>&1
cat gpurun_out/r05_s1_cr_ab.txt
