# round-4 profile session, part A (GPU box, repo root): what the driver line's fractions are recomputed from.
#  1. kernel trace of the driver's exact command (brackets included) cut into phases by the launch counts the line
#     itself records (profiles/phase_stats.py): steady / warm-up / TIMED / cold / big-ring rows of the headline kernel;
#  2. kernel stats of the same command without the brackets (one population per kernel);
#  3. HBM traffic (FETCH_SIZE / WRITE_SIZE, own passes, --settle 0) of the kernels round 3 left without a figure:
#     mixed_rollout_kernel<1|2> (config 4), rollout_kernel<RobotAssembly,3>, rollout_pg_pair_kernel<3> (65 536 lanes).
mkdir -p gpurun_out profiles/r04
export TMPDIR=/tmp NIG_PROFILE_ROUND=r04
python3 -c 'import __graft_entry__ as g; g.build(force=False)' > /dev/null
export NIG_NO_AUTOBUILD=1
R=profiles/r04
run() { echo "== $*" >&2; timeout -k 10 200 "$@"; echo "   rc=$?" >&2; }
run bash profiles/run_profile.sh r04_cr65536_driver --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_prof_driver.log 2>&1 \
  && cp gpurun_out/prof_r04_cr65536_driver/bench.json $R/cr65536_driver_bench.json \
  && cp gpurun_out/prof_r04_cr65536_driver/r04_cr65536_driver_kernel_stats.csv $R/cr65536_driver_all_launches_kernel_stats.csv \
  && python3 profiles/phase_stats.py $(find gpurun_out/prof_r04_cr65536_driver -name "*kernel_trace.csv" | head -1) $R/cr65536_driver_bench.json $R/cr65536_driver_phases.csv
NIG_PROFILE_EXTRA="--no-brackets" run bash profiles/run_profile.sh r04_cr65536_driver_nb --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_prof_driver_nb.log 2>&1 \
  && cp gpurun_out/prof_r04_cr65536_driver_nb/r04_cr65536_driver_nb_kernel_stats.csv $R/cr65536_driver_nobrackets_kernel_stats.csv \
  && cp gpurun_out/prof_r04_cr65536_driver_nb/bench.json $R/cr65536_driver_nobrackets_bench.json
for o in min full; do
  run bash profiles/run_pmc.sh r04_mixed1048576_$o --env mixed --mixed-outputs $o --steps 6 --warmup 1 > gpurun_out/r04_pmc_mixed_$o.log 2>&1 \
    && cp gpurun_out/pmc_r04_mixed1048576_$o/traffic_r04_mixed1048576_$o.json $R/r04_mixed1048576_${o}_pmc_traffic.json
done
run bash profiles/run_pmc.sh r04_ra262144_rollout_full --env ra --steps 10 --warmup 2 > gpurun_out/r04_pmc_ra.log 2>&1 \
  && cp gpurun_out/pmc_r04_ra262144_rollout_full/traffic_r04_ra262144_rollout_full.json $R/r04_ra262144_rollout_full_pmc_traffic.json
run bash profiles/run_pmc.sh r04_pg65536_rollout_full --env pg --batch 65536 --steps 10 --warmup 2 > gpurun_out/r04_pmc_pgpair.log 2>&1 \
  && cp gpurun_out/pmc_r04_pg65536_rollout_full/traffic_r04_pg65536_rollout_full.json $R/r04_pg65536_rollout_full_pmc_traffic.json
cp profiles/traffic.json gpurun_out/r04_traffic_merged.json
mkdir -p gpurun_out/profiles_r04 && cp -r $R/* gpurun_out/profiles_r04/
for f in gpurun_out/r04_prof_driver.log gpurun_out/r04_pmc_mixed_min.log gpurun_out/r04_pmc_mixed_full.log gpurun_out/r04_pmc_ra.log gpurun_out/r04_pmc_pgpair.log; do echo "--- $f"; tail -n 4 $f; done
cat $R/cr65536_driver_phases.csv
