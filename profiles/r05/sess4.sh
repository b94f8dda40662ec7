# round-5 GPU session 4: PowerGrid's action reads as two 16-byte loads per lane from a row-major slot (rowmaj: same bytes, 2 KiB contiguous
# per wave) and as non-temporal loads (ntload); the headline's recorder as a pure store wave (recnocomp: upper bound of a 4th wave);
# PowerGrid's closed loop with the observation stream on the register-resident stepper
mkdir -p gpurun_out
export TMPDIR=/tmp NIG_NO_AUTOBUILD=1
bash profiles/ab.sh "base rowmaj ntload" "pg 262144 full aos" "pg 262144 min aos" > gpurun_out/r05_s4_pg_ab.txt 2>&1
cat gpurun_out/r05_s4_pg_ab.txt
bash profiles/ab.sh "base recnocomp" "cr 65536 full aos" "cr 65536 min aos" > gpurun_out/r05_s4_cr_ab.txt 2>&1
cat gpurun_out/r05_s4_cr_ab.txt
run() { echo "== $*" >&2; timeout -k 10 "$@"; rc=$?; echo "   rc=$rc" >&2; return $rc; }
run 600 python -m pytest -x -q -m gpu tests/test_gpu_split.py -k "powergrid_paired_closed" > gpurun_out/r05_s4_pgpol_tests.log 2>&1; echo "pg closed-loop tests rc=$?"; tail -2 gpurun_out/r05_s4_pgpol_tests.log
for i in 1 2; do run 300 python profiles/bench_policy.py --env PowerGrid-v0 > gpurun_out/r05_s4_policy_pg_$i.txt 2>&1; tail -1 gpurun_out/r05_s4_policy_pg_$i.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:round(v['us_per_step'],3) for k,v in d.items()})"; done
