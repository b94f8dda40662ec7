# round-5 GPU session 9: the closed loop beyond one round (three-wave rounds vs one-wave, with and without the transition stream), the
# suite's split / policy / abi tests after the rounds rule and the version bump
mkdir -p gpurun_out
export TMPDIR=/tmp NIG_NO_AUTOBUILD=1
run() { echo "== $*" >&2; timeout -k 10 "$@"; rc=$?; echo "   rc=$rc" >&2; return $rc; }
run 900 python -m pytest -x -q -m gpu tests/test_gpu_split.py tests/test_gpu_parity.py tests/test_gpu_round3.py > gpurun_out/r05_s9_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r05_s9_tests.log
for B in 131072 262144; do for v in 256 0; do echo -n "closed loop CR $B lanes split_blocks=$v: "; NIG_SPLIT_BLOCKS=$v timeout -k 10 200 python profiles/bench_policy.py --batch $B --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print({k: round(v['us_per_step'],3) for k,v in d.items() if k.startswith('affine')})"; done; done > gpurun_out/r05_s9_policy_rounds.txt 2>&1
cat gpurun_out/r05_s9_policy_rounds.txt
