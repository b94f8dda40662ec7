# round-4 profile session, part B (GPU box, repo root):
#  1. the MFMA actor: MFMA busy cycles, issue stalls and the clock the chip held (GRBM_GUI_ACTIVE / 8 / wall, 16.8 ms
#     dispatches) in ONE counter pass -> duty cycle of the MFMA pipe ("prove or remove the DVFS claim");
#  2. PowerGrid 262 144 (configs[2]): kernel stats + SQ issue counters of this round's build;
#  3. the mixed launch (configs[3]), both output modes: kernel stats;
#  4. closed loop at 65 536 lanes, three envs (un-profiled): profiles/r04/policy_<env>.json;
#  5. the driver's exact command, un-profiled.
mkdir -p gpurun_out profiles/r04
export TMPDIR=/tmp NIG_PROFILE_ROUND=r04
python3 -c 'import __graft_entry__ as g; g.build(force=False)' > /dev/null
export NIG_NO_AUTOBUILD=1
R=profiles/r04
run() { echo "== $*" >&2; timeout -k 10 200 "$@"; echo "   rc=$?" >&2; }
run rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/sq_r04_mlp -o r04_mlp -- python3 profiles/bench_policy.py --only-mlp --mlp-launches 8 --mlp-steps 200 > gpurun_out/r04_sq_mlp.json 2> gpurun_out/r04_sq_mlp.log
run rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04_mlp -o r04_mlp -- python3 profiles/bench_policy.py --only-mlp --mlp-launches 8 --mlp-steps 200 > $R/mlp_cr65536_bench.json 2> gpurun_out/r04_prof_mlp.log && cp $(find gpurun_out/prof_r04_mlp -name "*kernel_stats.csv" | head -1) $R/mlp_cr65536_kernel_stats.csv
python3 - <<'PY' > $R/mlp_cr65536_sq.txt 2>&1
import csv, glob
from collections import defaultdict
f = glob.glob("gpurun_out/sq_r04_mlp/**/*counter_collection.csv", recursive=True)
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if "rollout_mlp" in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    med = {}
    for c, v in sorted(d.items()):
        v = sorted(v); med[c] = v[len(v) // 2]; print("   %-26s median %.5g  (n=%d)" % (c, med[c], len(v)))
    if "GRBM_GUI_ACTIVE" in med and "SQ_VALU_MFMA_BUSY_CYCLES" in med:
        cyc = med["GRBM_GUI_ACTIVE"] / 8.0                 # rocprofv3 sums the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
        print("   chip cycles per dispatch (GRBM_GUI_ACTIVE / 8)        %.5g" % cyc)
        print("   MFMA busy cycles per SIMD (SQ_VALU_MFMA_BUSY / 1024)   %.5g" % (med["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0))
        print("   MFMA duty cycle                                         %.3f" % (med["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc))
PY
NIG_PROFILE_EXTRA="--no-step-api --no-brackets --no-single-env --no-mixed" run bash profiles/run_profile.sh r04_pg262144 --env pg --steps 10 --warmup 2 > gpurun_out/r04_prof_pg.log 2>&1 && cp gpurun_out/prof_r04_pg262144/r04_pg262144_kernel_stats.csv $R/pg262144_rollout_full_kernel_stats.csv && cp gpurun_out/prof_r04_pg262144/bench.json $R/pg262144_rollout_full_bench.json
run bash profiles/run_sq.sh r04_pg262144 --env pg --steps 10 --warmup 2 > $R/pg262144_rollout_full_sq.txt 2>&1
for o in min full; do
  run bash profiles/run_profile.sh r04_mixed1m_$o --env mixed --mixed-outputs $o --steps 8 --warmup 2 > gpurun_out/r04_prof_mixed_$o.log 2>&1 && cp gpurun_out/prof_r04_mixed1m_$o/r04_mixed1m_${o}_kernel_stats.csv $R/mixed1048576_${o}_kernel_stats.csv && cp gpurun_out/prof_r04_mixed1m_$o/bench.json $R/mixed1048576_${o}_bench.json
done
for e in ChemicalReactor-v0 PowerGrid-v0 RobotAssembly-v0; do
  run python3 profiles/bench_policy.py --env $e --steps 200 --mlp-steps 20 > $R/policy_$e.json 2> gpurun_out/r04_policy_$e.log
done
run python3 bench.py --gpus 1 --steps 20 --warmup 5 > $R/driver_bench_unprofiled.json 2> gpurun_out/r04_driver_unprofiled.err
mkdir -p gpurun_out/profiles_r04 && cp -r $R/* gpurun_out/profiles_r04/
cat $R/mlp_cr65536_sq.txt; tail -3 $R/pg262144_rollout_full_sq.txt; head -5 $R/pg262144_rollout_full_kernel_stats.csv
