# final round-2 session: tests, profiles, sweep, policy benchmarks, the driver's command
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gpu_tests_final.log 2>&1; tail -4 gpurun_out/r02_gpu_tests_final.log
bash profiles/r02/prof_all.sh > gpurun_out/prof_all.log 2>&1; tail -3 gpurun_out/prof_all.log
bash profiles/sweep.sh r02 2>&1 | tail -40
cp gpurun_out/sweep_r02.jsonl gpurun_out/profiles_r02/sweep_r02.jsonl
for e in ChemicalReactor-v0 PowerGrid-v0; do timeout -k 10 120 python profiles/bench_policy.py --steps 100 --env $e 2>/dev/null | tee gpurun_out/profiles_r02/policy_$e.json | cut -c1-400; done
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tee gpurun_out/profiles_r02/driver_bench_unprofiled.json | python -c "import json,sys; d=json.load(sys.stdin); print('DRIVER value %.3e launch_us %.1f frac %.3f traffic %s pg frac %.3f step_api %.2f us' % (d['value'], d['roofline']['launch_us'], d['roofline']['frac'], d['roofline']['traffic'], d['powergrid']['roofline']['frac'], d['step_api']['launch_us']))"
