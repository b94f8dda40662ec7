# round-4 profile session, part E (GPU box, repo root): HBM traffic passes of the two BASELINE-config kernels whose figures in
# traffic.json still dated from round 3 (headline three-wave kernel, PowerGrid wide form), on the final build.
mkdir -p gpurun_out profiles/r04
export TMPDIR=/tmp NIG_PROFILE_ROUND=r04
python3 -c 'import __graft_entry__ as g; g.build(force=False)' > /dev/null
export NIG_NO_AUTOBUILD=1
R=profiles/r04
run() { echo "== $*" >&2; timeout -k 10 240 "$@"; echo "   rc=$?" >&2; }
run bash profiles/run_pmc.sh r04_cr65536_rollout_full --env cr --steps 20 --warmup 5 > gpurun_out/r04_pmc_cr.log 2>&1 \
  && cp gpurun_out/pmc_r04_cr65536_rollout_full/traffic_r04_cr65536_rollout_full.json $R/r04_cr65536_rollout_full_pmc_traffic.json
run bash profiles/run_pmc.sh r04_pg262144_rollout_full --env pg --steps 10 --warmup 2 > gpurun_out/r04_pmc_pg.log 2>&1 \
  && cp gpurun_out/pmc_r04_pg262144_rollout_full/traffic_r04_pg262144_rollout_full.json $R/r04_pg262144_rollout_full_pmc_traffic.json
cp profiles/traffic.json gpurun_out/r04_traffic_merged_e.json
mkdir -p gpurun_out/profiles_r04 && cp $R/r04_cr65536_rollout_full_pmc_traffic.json $R/r04_pg262144_rollout_full_pmc_traffic.json gpurun_out/profiles_r04/
tail -n 5 gpurun_out/r04_pmc_cr.log gpurun_out/r04_pmc_pg.log
