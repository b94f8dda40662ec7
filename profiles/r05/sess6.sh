# round-5 GPU session 6: the whole GPU suite on the current build; PowerGrid after the instruction diet (maximum clips, fused economic
# sum, one action set) against the session-1 build (oneset = generator index change only); the mixed launch against its bodies' rates
mkdir -p gpurun_out
export TMPDIR=/tmp NIG_NO_AUTOBUILD=1
run() { echo "== $*" >&2; timeout -k 10 "$@"; rc=$?; echo "   rc=$rc" >&2; return $rc; }
run 1100 python -m pytest -x -q -m gpu tests > gpurun_out/r05_s6_gpu_tests.log 2>&1; rc=$?; echo "gpu suite rc=$rc"; tail -4 gpurun_out/r05_s6_gpu_tests.log
[ $rc -eq 0 ] || exit 1
bash profiles/ab.sh "base oneset" "pg 262144 full aos" "pg 262144 min aos" "pg 262144 none aos" "mixed 1048576 full aos --mixed-outputs min" > gpurun_out/r05_s6_pg_ab.txt 2>&1
cat gpurun_out/r05_s6_pg_ab.txt
mkdir -p gpurun_out/mixed_floor
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/mixed_floor -o mf -- python3 profiles/tools/mixed_floor.py run > gpurun_out/r05_s6_mixed_floor_run.txt 2> gpurun_out/r05_s6_mixed_floor.err || { tail -5 gpurun_out/r05_s6_mixed_floor.err; exit 1; }
f=$(find gpurun_out/mixed_floor -name "*counter_collection.csv" | head -1)
python3 profiles/tools/mixed_floor.py table $f > gpurun_out/r05_s6_mixed_floor.txt 2>&1; cat gpurun_out/r05_s6_mixed_floor.txt
