# round-4 profile session, part D (GPU box, repo root): the FINAL build of the round (library 0.4.0: plants' model v2, paired bodies
# in the mixed kernel, PowerGrid's register-resident closed-loop stepper, wide-form threshold) -- what changed since parts A-C:
#  1. the driver's command, un-profiled (the line the README quotes) and under the tracer cut into phases;
#  2. the mixed launch: kernel stats for reward + flags / with observation rows, SQ issue counters for reward + flags (the
#     instruction count behind the 2.9 -> 2.7 ms), HBM traffic passes;
#  3. the closed loops (PowerGrid paired form with the register-resident stepper).
mkdir -p gpurun_out profiles/r04
export TMPDIR=/tmp NIG_PROFILE_ROUND=r04
python3 -c 'import __graft_entry__ as g; g.build(force=False)' > /dev/null
export NIG_NO_AUTOBUILD=1
R=profiles/r04
run() { echo "== $*" >&2; timeout -k 10 240 "$@"; echo "   rc=$?" >&2; }
run python3 bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r04_driver_final.err | grep '^{' > $R/driver_bench_final.json
run bash profiles/run_profile.sh r04_cr65536_driver_final --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_prof_driver_final.log 2>&1 \
  && cp gpurun_out/prof_r04_cr65536_driver_final/bench.json $R/cr65536_driver_final_bench.json \
  && python3 profiles/phase_stats.py $(find gpurun_out/prof_r04_cr65536_driver_final -name "*kernel_trace.csv" | head -1) $R/cr65536_driver_final_bench.json $R/cr65536_driver_final_phases.csv
for o in min full; do
  NIG_PROFILE_EXTRA="--no-step-api --no-brackets --no-single-env" run bash profiles/run_profile.sh r04_mixed1048576_${o}_final --env mixed --mixed-outputs $o --steps 10 --warmup 2 > gpurun_out/r04_prof_mixed_${o}_final.log 2>&1 \
    && cp gpurun_out/prof_r04_mixed1048576_${o}_final/r04_mixed1048576_${o}_final_kernel_stats.csv $R/mixed1048576_${o}_final_kernel_stats.csv \
    && cp gpurun_out/prof_r04_mixed1048576_${o}_final/bench.json $R/mixed1048576_${o}_final_bench.json
done
run bash profiles/run_sq.sh r04_mixed1m_min_final --env mixed --mixed-outputs min --steps 6 --warmup 1 > $R/mixed1048576_min_final_sq.txt 2> gpurun_out/r04_sq_mixed_final.err
for o in min full; do
  run bash profiles/run_pmc.sh r04_mixed1048576_${o}_final --env mixed --mixed-outputs $o --steps 6 --warmup 1 > gpurun_out/r04_pmc_mixed_${o}_final.log 2>&1 \
    && cp gpurun_out/pmc_r04_mixed1048576_${o}_final/traffic_r04_mixed1048576_${o}_final.json $R/r04_mixed1048576_${o}_final_pmc_traffic.json
done
for e in PowerGrid-v0 RobotAssembly-v0; do
  run python3 profiles/bench_policy.py --env $e --steps 200 --mlp-steps 20 2> gpurun_out/r04_policy_final_$e.log | grep '^{' > $R/policy_${e}_final.json
done
cp profiles/traffic.json gpurun_out/r04_traffic_merged_final.json
mkdir -p gpurun_out/profiles_r04 && cp -r $R/* gpurun_out/profiles_r04/
cat $R/cr65536_driver_final_phases.csv; head -3 $R/mixed1048576_min_final_kernel_stats.csv; head -3 $R/mixed1048576_full_final_kernel_stats.csv; cat $R/mixed1048576_min_final_sq.txt | head -12
python3 - <<'PY'
import json
d = json.load(open("profiles/r04/driver_bench_final.json"))
r = d["roofline"]
print("headline %.4g env-steps/s, %.1f us, frac %.3f (alg %.3f); pg %.3f ms frac %.3f; mixed %.3f ms frac %.3f; step %.2f us; single %.0f/s" % (
    d["value"], r["launch_us"], r["frac"], r["frac_algorithmic"], d["powergrid"]["ms_per_step"], d["powergrid"]["roofline"]["frac"],
    d["mixed"]["ms_per_step"], d["mixed"]["roofline"]["frac"], d["step_api"]["launch_us"], d["single_env"]["value"]))
PY
