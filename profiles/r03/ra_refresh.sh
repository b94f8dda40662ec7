# after the RobotAssembly instruction diet (quadrant from the mantissa, bit selects, clip skipped when no lane needs it,
# float32 bound compares): refresh what contains RobotAssembly's body -- its own kernel stats + SQ and the mixed launch's.
mkdir -p gpurun_out profiles/r03
export TMPDIR=/tmp
python3 -c 'import __graft_entry__ as g; g.build(force=False)' > /dev/null
export NIG_NO_AUTOBUILD=1
R=profiles/r03
run() { echo "== $*" ; timeout -k 10 170 "$@"; echo "   rc=$?"; }
NIG_PROFILE_EXTRA="--no-step-api --no-brackets" run bash profiles/run_profile.sh r03_ra262144 --env ra --steps 10 --warmup 2 > gpurun_out/r03_prof_ra.log 2>&1 && cp gpurun_out/prof_r03_ra262144/r03_ra262144_kernel_stats.csv $R/ra262144_rollout_full_kernel_stats.csv && cp gpurun_out/prof_r03_ra262144/bench.json $R/ra262144_rollout_full_bench.json
run bash profiles/run_sq.sh r03_ra262144 --env ra --steps 10 --warmup 2 > $R/ra262144_rollout_full_sq.txt 2>&1
for o in min full; do
  run bash profiles/run_profile.sh r03_mixed1m_$o --env mixed --mixed-outputs $o --steps 8 --warmup 2 > gpurun_out/r03_prof_mixed_$o.log 2>&1 && cp gpurun_out/prof_r03_mixed1m_$o/r03_mixed1m_${o}_kernel_stats.csv $R/mixed1048576_${o}_kernel_stats.csv && cp gpurun_out/prof_r03_mixed1m_$o/bench.json $R/mixed1048576_${o}_bench.json
  run bash profiles/run_sq.sh r03_mixed1m_$o --env mixed --mixed-outputs $o --steps 6 --warmup 1 > $R/mixed1048576_${o}_sq.txt 2>&1
done
mkdir -p gpurun_out/profiles_r03 && cp -r $R/* gpurun_out/profiles_r03/
for f in ra262144_rollout_full mixed1048576_min mixed1048576_full; do echo "--- $f"; grep -i "rollout" $R/${f}_kernel_stats.csv | head -3; tail -12 $R/${f}_sq.txt; done
