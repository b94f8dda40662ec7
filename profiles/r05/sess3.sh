# round-5 GPU session 3: what do PowerGrid's action loads cost with full outputs -- the wait, or the read traffic in the write stream?
#   deadload: the loads go out but nothing waits for them (hash actions);  noact: no loads at all;  plainst: ordinary instead of
#   streaming trajectory stores;  --ring 16: the action ring (134 MB) stays in the Infinity Cache, the reads never reach HBM
mkdir -p gpurun_out
export TMPDIR=/tmp NIG_NO_AUTOBUILD=1
bash profiles/ab.sh "base noact deadload plainst" "pg 262144 full aos" "pg 262144 full aos --ring 16" > gpurun_out/r05_s3_pg_ab.txt 2>&1
cat gpurun_out/r05_s3_pg_ab.txt
run() { echo "== $*" >&2; timeout -k 10 "$@"; rc=$?; echo "   rc=$rc" >&2; return $rc; }
run 900 python -m pytest -x -q -m gpu tests/test_gpu_split.py -k "closed or policy" > gpurun_out/r05_s3_split_tests.log 2>&1; echo "closed-loop tests rc=$?"; tail -5 gpurun_out/r05_s3_split_tests.log
run 300 python profiles/bench_policy.py --env PowerGrid-v0 > gpurun_out/r05_s3_policy_pg.txt 2>&1; tail -2 gpurun_out/r05_s3_policy_pg.txt
run 300 python profiles/bench_policy.py --env RobotAssembly-v0 > gpurun_out/r05_s3_policy_ra.txt 2>&1; tail -2 gpurun_out/r05_s3_policy_ra.txt
