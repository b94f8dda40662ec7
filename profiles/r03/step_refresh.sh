# after the step-kernel changes (helper waves, early loads): the step-API probe, the graph-replay HBM traffic (two PMC passes
# of their own, --settle 0, 2 500 instrumented dispatches each), kernel stats of a graph-mode run.
mkdir -p gpurun_out profiles/r03
export TMPDIR=/tmp
python3 -c 'import __graft_entry__ as g; g.build(force=False)' > /dev/null
export NIG_NO_AUTOBUILD=1
R=profiles/r03
timeout -k 10 200 python3 profiles/r03/step_api_probe.py 2>/dev/null > $R/step_api_probe_after.txt; cat $R/step_api_probe_after.txt
timeout -k 10 170 bash profiles/run_pmc.sh r03_cr65536_graph_step --mode graph --steps 8 --warmup 2 > gpurun_out/r03_pmc_graph.log 2>&1 && cp gpurun_out/pmc_r03_cr65536_graph_step/traffic_r03_cr65536_graph_step.json $R/r03_cr65536_graph_step_pmc_traffic.json; tail -n 4 gpurun_out/r03_pmc_graph.log
NIG_PROFILE_EXTRA="--no-step-api --no-brackets --no-mixed --no-powergrid" timeout -k 10 170 bash profiles/run_profile.sh r03_cr65536_graph --mode graph --steps 8 --warmup 2 > gpurun_out/r03_prof_graph.log 2>&1 && cp gpurun_out/prof_r03_cr65536_graph/r03_cr65536_graph_kernel_stats.csv $R/cr65536_graph_step_kernel_stats.csv; grep step_kernel $R/cr65536_graph_step_kernel_stats.csv | cut -c1-200
cp profiles/traffic.json gpurun_out/r03_traffic_merged.json
mkdir -p gpurun_out/profiles_r03 && cp -r $R/* gpurun_out/profiles_r03/
