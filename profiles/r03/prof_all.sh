# round-3 profile session (GPU box, repo root).  Kernel stats of the driver's exact command; PowerGrid (configs[2]):
# kernel stats, HBM traffic, SQ counters (issue + LDS sets); the mixed launch (configs[3]) with and without observation
# rows: kernel stats + SQ; RobotAssembly: kernel stats + SQ; the MFMA actor: 60 launches under the tracer + SQ/MFMA counters.
# Counter passes are their own runs (never combined with a trace), --settle 0 (profiles/README.md).
mkdir -p gpurun_out profiles/r03
export TMPDIR=/tmp
python3 -c 'import __graft_entry__ as g; g.build(force=False)' > /dev/null
export NIG_NO_AUTOBUILD=1
R=profiles/r03
run() { echo "== $*" ; timeout -k 10 170 "$@"; echo "   rc=$?"; }
run bash profiles/run_profile.sh r03_cr65536_driver --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_prof_driver.log 2>&1 && cp gpurun_out/prof_r03_cr65536_driver/r03_cr65536_driver_kernel_stats.csv $R/cr65536_driver_kernel_stats.csv && cp gpurun_out/prof_r03_cr65536_driver/bench.json $R/cr65536_driver_bench.json
NIG_PROFILE_EXTRA="--no-step-api --no-brackets" run bash profiles/run_profile.sh r03_pg262144 --env pg --steps 10 --warmup 2 > gpurun_out/r03_prof_pg.log 2>&1 && cp gpurun_out/prof_r03_pg262144/r03_pg262144_kernel_stats.csv $R/pg262144_rollout_full_kernel_stats.csv && cp gpurun_out/prof_r03_pg262144/bench.json $R/pg262144_rollout_full_bench.json
run bash profiles/run_pmc.sh r03_pg262144_rollout_full --env pg --steps 10 --warmup 2 > gpurun_out/r03_pmc_pg.log 2>&1 && cp gpurun_out/pmc_r03_pg262144_rollout_full/traffic_r03_pg262144_rollout_full.json $R/r03_pg262144_rollout_full_pmc_traffic.json
run bash profiles/run_sq.sh r03_pg262144 --env pg --steps 10 --warmup 2 > $R/pg262144_rollout_full_sq.txt 2>&1
NIG_SQ_COUNTERS="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_INSTS_SMEM" run bash profiles/run_sq.sh r03_pg262144_lds --env pg --steps 10 --warmup 2 > $R/pg262144_rollout_full_sq_lds.txt 2>&1
for o in min full; do
  run bash profiles/run_profile.sh r03_mixed1m_$o --env mixed --mixed-outputs $o --steps 8 --warmup 2 > gpurun_out/r03_prof_mixed_$o.log 2>&1 && cp gpurun_out/prof_r03_mixed1m_$o/r03_mixed1m_${o}_kernel_stats.csv $R/mixed1048576_${o}_kernel_stats.csv && cp gpurun_out/prof_r03_mixed1m_$o/bench.json $R/mixed1048576_${o}_bench.json
  run bash profiles/run_sq.sh r03_mixed1m_$o --env mixed --mixed-outputs $o --steps 6 --warmup 1 > $R/mixed1048576_${o}_sq.txt 2>&1
done
NIG_PROFILE_EXTRA="--no-step-api --no-brackets" run bash profiles/run_profile.sh r03_ra262144 --env ra --steps 10 --warmup 2 > gpurun_out/r03_prof_ra.log 2>&1 && cp gpurun_out/prof_r03_ra262144/r03_ra262144_kernel_stats.csv $R/ra262144_rollout_full_kernel_stats.csv && cp gpurun_out/prof_r03_ra262144/bench.json $R/ra262144_rollout_full_bench.json
run bash profiles/run_sq.sh r03_ra262144 --env ra --steps 10 --warmup 2 > $R/ra262144_rollout_full_sq.txt 2>&1
# the MFMA actor: 60 launches of 20 env.step under the tracer, then its issue / MFMA counters in a pass of their own
run rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03_mlp -o r03_mlp -- python3 profiles/bench_policy.py --only-mlp --mlp-launches 60 --mlp-steps 20 > $R/mlp_cr65536_bench.json 2> gpurun_out/r03_prof_mlp.log && cp $(find gpurun_out/prof_r03_mlp -name "*kernel_stats.csv" | head -1) $R/mlp_cr65536_kernel_stats.csv
run rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d gpurun_out/sq_r03_mlp -o r03_mlp -- python3 profiles/bench_policy.py --only-mlp --mlp-launches 12 --mlp-steps 20 > gpurun_out/r03_sq_mlp.json 2> gpurun_out/r03_sq_mlp.log
python3 - <<'PY' > $R/mlp_cr65536_sq.txt 2>&1
import csv, glob
from collections import defaultdict
f = glob.glob("gpurun_out/sq_r03_mlp/**/*counter_collection.csv", recursive=True)
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if "rollout_mlp" in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        v = sorted(v); print("   %-26s median %.4g  (n=%d)" % (c, v[len(v) // 2], len(v)))
PY
cp profiles/traffic.json gpurun_out/r03_traffic_merged.json
mkdir -p gpurun_out/profiles_r03 && cp -r $R/* gpurun_out/profiles_r03/
for f in gpurun_out/r03_prof_driver.log gpurun_out/r03_pmc_pg.log $R/pg262144_rollout_full_sq.txt $R/mixed1048576_min_sq.txt $R/ra262144_rollout_full_sq.txt $R/mlp_cr65536_sq.txt; do echo "--- $f"; tail -n 3 $f; done
