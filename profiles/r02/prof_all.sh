# round-2 profile session (GPU box, repo root): kernel stats of the driver's exact bench command, PowerGrid
# (BASELINE config 3) kernel stats + HBM traffic + SQ counters, the mixed launch, and ONE graph-mode PMC pass.
mkdir -p gpurun_out profiles/r02
export TMPDIR=/tmp
python3 -c 'import __graft_entry__ as g; g.build(force=False)' > /dev/null
export NIG_NO_AUTOBUILD=1
set -x
bash profiles/run_profile.sh r02_cr65536_driver --gpus 1 --steps 20 --warmup 5 > gpurun_out/prof_driver.log 2>&1 && cp gpurun_out/prof_r02_cr65536_driver/r02_cr65536_driver_kernel_stats.csv profiles/r02/cr65536_driver_kernel_stats.csv && cp gpurun_out/prof_r02_cr65536_driver/bench.json profiles/r02/cr65536_driver_bench.json
bash profiles/run_profile.sh r02_pg262144 --env pg --batch 262144 --steps 10 --warmup 2 > gpurun_out/prof_pg.log 2>&1 && cp gpurun_out/prof_r02_pg262144/r02_pg262144_kernel_stats.csv profiles/r02/pg262144_rollout_full_kernel_stats.csv && cp gpurun_out/prof_r02_pg262144/bench.json profiles/r02/pg262144_rollout_full_bench.json
bash profiles/run_profile.sh r02_mixed1m --env mixed --steps 8 --warmup 2 > gpurun_out/prof_mixed.log 2>&1 && cp gpurun_out/prof_r02_mixed1m/r02_mixed1m_kernel_stats.csv profiles/r02/mixed1048576_kernel_stats.csv && cp gpurun_out/prof_r02_mixed1m/bench.json profiles/r02/mixed1048576_bench.json
bash profiles/run_pmc.sh r02_cr65536_rollout_full --steps 20 --warmup 5 --no-powergrid > gpurun_out/pmc_cr.log 2>&1 && cp gpurun_out/pmc_r02_cr65536_rollout_full/traffic_r02_cr65536_rollout_full.json profiles/r02/r02_cr65536_rollout_full_pmc_traffic.json
bash profiles/run_pmc.sh r02_pg262144_rollout_full --env pg --batch 262144 --steps 10 --warmup 2 > gpurun_out/pmc_pg.log 2>&1 && cp gpurun_out/pmc_r02_pg262144_rollout_full/traffic_r02_pg262144_rollout_full.json profiles/r02/r02_pg262144_rollout_full_pmc_traffic.json
bash profiles/run_sq.sh r02_pg262144 --env pg --batch 262144 --steps 10 --warmup 2 --no-powergrid > profiles/r02/pg262144_rollout_full_sq.txt 2>&1
bash profiles/run_sq.sh r02_cr65536 --steps 20 --warmup 5 --no-powergrid > profiles/r02/cr65536_rollout_full_sq.txt 2>&1
# closed-loop policy rollouts (three-wave form at 65 536 lanes) under the tracer
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02_policy -o r02_policy -- python3 profiles/bench_policy.py --steps 100 > gpurun_out/prof_r02_policy.json 2> gpurun_out/prof_policy.log && cp $(find gpurun_out/prof_r02_policy -name "*kernel_stats.csv" | head -1) profiles/r02/policy_cr65536_kernel_stats.csv
# ONE graph-replay PMC pass (the abort of round 1 left no log; library built beforehand this time), stderr kept
timeout -k 10 200 bash profiles/run_pmc.sh r02_cr65536_graph_step --mode graph --steps 8 --warmup 2 > gpurun_out/pmc_graph.log 2>&1; echo "graph pmc rc=$?" | tee -a gpurun_out/pmc_graph.log
cp gpurun_out/pmc_r02_cr65536_graph_step/traffic_r02_cr65536_graph_step.json profiles/r02/r02_cr65536_graph_step_pmc_traffic.json 2>/dev/null
cp gpurun_out/pmc_r02_cr65536_graph_step/bench_FETCH_SIZE.err profiles/r02/graph_pmc_FETCH_SIZE.stderr.txt 2>/dev/null
cp profiles/traffic.json gpurun_out/traffic_merged.json
mkdir -p gpurun_out/profiles_r02 && cp -r profiles/r02/* gpurun_out/profiles_r02/
tail -5 gpurun_out/prof_driver.log gpurun_out/prof_pg.log gpurun_out/prof_mixed.log gpurun_out/pmc_pg.log gpurun_out/pmc_graph.log
