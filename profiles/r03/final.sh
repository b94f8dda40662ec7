# final round-3 measurement session (GPU box, repo root): the sweep behind DESIGN.md's table, the closed-loop policy
# benchmarks, the driver's exact command unprofiled.  (Tests and profiles: prof_all.sh.)
mkdir -p gpurun_out/profiles_r03
export NIG_NO_AUTOBUILD=1
bash profiles/sweep.sh r03 2>&1 | tail -60
cp gpurun_out/sweep_r03.jsonl gpurun_out/profiles_r03/sweep_r03.jsonl
for e in ChemicalReactor-v0 PowerGrid-v0 RobotAssembly-v0; do timeout -k 10 120 python profiles/bench_policy.py --steps 100 --env $e 2>/dev/null | tee gpurun_out/profiles_r03/policy_$e.json | cut -c1-600; done
timeout -k 10 200 python profiles/bench_single_env.py 2>/dev/null | tee gpurun_out/profiles_r03/single_env.json | cut -c1-400
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null > gpurun_out/profiles_r03/driver_bench_unprofiled.json; python profiles/r03/show_bench.py gpurun_out/profiles_r03/driver_bench_unprofiled.json
