# round-5 profile session on the round's final build (GPU box, repo root): usage  bash profiles/r05/prof_final.sh A|B|C|D
#  A  the driver's command un-profiled and under the tracer (phases of the headline kernel), PowerGrid / RobotAssembly / mixed kernel stats
#  B  HBM traffic passes (FETCH_SIZE / WRITE_SIZE, own passes) for the headline, PowerGrid, RobotAssembly and the mixed launch
#  C  SQ issue + LDS counters for PowerGrid, SQ for the mixed launch, the MFMA actor (stats + MFMA duty cycle), closed-loop benches
#  D  the sweep behind DESIGN.md's measurement table
part=${1:-A}
mkdir -p gpurun_out/profiles_r05
export TMPDIR=/tmp NIG_PROFILE_ROUND=r05
python3 -c 'import __graft_entry__ as g; g.build(force=False)' > /dev/null
export NIG_NO_AUTOBUILD=1
R=gpurun_out/profiles_r05
run() { echo "== $*" >&2; timeout -k 10 300 "$@"; echo "   rc=$?" >&2; }
if [ $part = A ]; then
  run python3 bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r05_driver_final.err | grep '^{' > $R/driver_bench_final.json
  run bash profiles/run_profile.sh r05_cr65536_driver --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_prof_driver.log 2>&1 \
    && cp gpurun_out/prof_r05_cr65536_driver/bench.json $R/cr65536_driver_bench.json \
    && cp gpurun_out/prof_r05_cr65536_driver/r05_cr65536_driver_kernel_stats.csv $R/cr65536_driver_all_launches_kernel_stats.csv \
    && python3 profiles/phase_stats.py $(find gpurun_out/prof_r05_cr65536_driver -name "*kernel_trace.csv" | head -1) $R/cr65536_driver_bench.json $R/cr65536_driver_phases.csv
  NIG_PROFILE_EXTRA="--no-step-api --no-brackets --no-single-env" run bash profiles/run_profile.sh r05_pg262144_rollout_full --env pg --steps 300 --warmup 5 > gpurun_out/r05_prof_pg.log 2>&1 \
    && cp gpurun_out/prof_r05_pg262144_rollout_full/r05_pg262144_rollout_full_kernel_stats.csv $R/pg262144_rollout_full_kernel_stats.csv && cp gpurun_out/prof_r05_pg262144_rollout_full/bench.json $R/pg262144_rollout_full_bench.json
  NIG_PROFILE_EXTRA="--no-step-api --no-brackets --no-single-env" run bash profiles/run_profile.sh r05_pg262144_rollout_min --env pg --outputs min --steps 300 --warmup 5 > gpurun_out/r05_prof_pg_min.log 2>&1 \
    && cp gpurun_out/prof_r05_pg262144_rollout_min/r05_pg262144_rollout_min_kernel_stats.csv $R/pg262144_rollout_min_kernel_stats.csv
  NIG_PROFILE_EXTRA="--no-step-api --no-brackets --no-single-env" run bash profiles/run_profile.sh r05_ra262144_rollout_full --env ra --steps 200 --warmup 5 > gpurun_out/r05_prof_ra.log 2>&1 \
    && cp gpurun_out/prof_r05_ra262144_rollout_full/r05_ra262144_rollout_full_kernel_stats.csv $R/ra262144_rollout_full_kernel_stats.csv && cp gpurun_out/prof_r05_ra262144_rollout_full/bench.json $R/ra262144_rollout_full_bench.json
  for o in min full; do
    NIG_PROFILE_EXTRA="--no-step-api --no-brackets --no-single-env" run bash profiles/run_profile.sh r05_mixed1048576_$o --env mixed --mixed-outputs $o --steps 200 --warmup 4 > gpurun_out/r05_prof_mixed_$o.log 2>&1 \
      && cp gpurun_out/prof_r05_mixed1048576_$o/r05_mixed1048576_${o}_kernel_stats.csv $R/mixed1048576_${o}_kernel_stats.csv && cp gpurun_out/prof_r05_mixed1048576_$o/bench.json $R/mixed1048576_${o}_bench.json
  done
  cat $R/cr65536_driver_phases.csv; for f in pg262144_rollout_full pg262144_rollout_min ra262144_rollout_full mixed1048576_min mixed1048576_full; do head -3 $R/${f}_kernel_stats.csv; done
fi
if [ $part = B ]; then
  run bash profiles/run_pmc.sh r05_cr65536_rollout_full --env cr --steps 20 --warmup 5 > gpurun_out/r05_pmc_cr.log 2>&1 \
    && cp gpurun_out/pmc_r05_cr65536_rollout_full/traffic_r05_cr65536_rollout_full.json $R/r05_cr65536_rollout_full_pmc_traffic.json
  run bash profiles/run_pmc.sh r05_pg262144_rollout_full --env pg --steps 10 --warmup 2 > gpurun_out/r05_pmc_pg.log 2>&1 \
    && cp gpurun_out/pmc_r05_pg262144_rollout_full/traffic_r05_pg262144_rollout_full.json $R/r05_pg262144_rollout_full_pmc_traffic.json
  run bash profiles/run_pmc.sh r05_ra262144_rollout_full --env ra --steps 10 --warmup 2 > gpurun_out/r05_pmc_ra.log 2>&1 \
    && cp gpurun_out/pmc_r05_ra262144_rollout_full/traffic_r05_ra262144_rollout_full.json $R/r05_ra262144_rollout_full_pmc_traffic.json
  for o in min full; do
    run bash profiles/run_pmc.sh r05_mixed1048576_$o --env mixed --mixed-outputs $o --steps 6 --warmup 1 > gpurun_out/r05_pmc_mixed_$o.log 2>&1 \
      && cp gpurun_out/pmc_r05_mixed1048576_$o/traffic_r05_mixed1048576_$o.json $R/r05_mixed1048576_${o}_pmc_traffic.json
  done
  cp profiles/traffic.json $R/traffic_merged.json
  tail -n 4 gpurun_out/r05_pmc_cr.log gpurun_out/r05_pmc_pg.log gpurun_out/r05_pmc_ra.log
fi
if [ $part = C ]; then
  for o in min full; do
    run bash profiles/run_sq.sh r05_pg262144_${o}_issue --env pg --outputs $o --steps 8 --warmup 2 > $R/pg262144_rollout_${o}_sq.txt 2> gpurun_out/r05_sq_pg_${o}.err
    NIG_SQ_COUNTERS="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" run bash profiles/run_sq.sh r05_pg262144_${o}_lds --env pg --outputs $o --steps 8 --warmup 2 > $R/pg262144_rollout_${o}_lds_sq.txt 2> gpurun_out/r05_sq_pg_${o}_lds.err
  done
  run bash profiles/run_sq.sh r05_mixed1m_min --env mixed --mixed-outputs min --steps 6 --warmup 1 > $R/mixed1048576_min_sq.txt 2> gpurun_out/r05_sq_mixed.err
  run rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/sq_r05_mlp -o r05_mlp -- python3 profiles/bench_policy.py --only-mlp --mlp-launches 8 --mlp-steps 200 > gpurun_out/r05_sq_mlp.json 2> gpurun_out/r05_sq_mlp.log
  python3 - <<'PY' > $R/mlp_cr65536_head4_sq.txt 2>&1
import csv, glob
from collections import defaultdict
f = glob.glob("gpurun_out/sq_r05_mlp/**/*counter_collection.csv", recursive=True)
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if "rollout_mlp" in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    med = {}
    for c, v in sorted(d.items()):
        v = sorted(v); med[c] = v[len(v) // 2]; print("   %-26s median %.5g  (n=%d)" % (c, med[c], len(v)))
    cyc = med["GRBM_GUI_ACTIVE"] / 8.0
    print("   chip cycles per dispatch (GRBM_GUI_ACTIVE / 8)        %.5g" % cyc)
    print("   MFMA busy cycles per SIMD (SQ_VALU_MFMA_BUSY / 1024)   %.5g" % (med["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0))
    print("   MFMA duty cycle                                         %.3f" % (med["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc))
PY
  run rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05_mlp -o r05_mlp -- python3 profiles/bench_policy.py --only-mlp --mlp-launches 8 --mlp-steps 200 > $R/mlp_cr65536_bench.json 2> gpurun_out/r05_prof_mlp.log && cp $(find gpurun_out/prof_r05_mlp -name "*kernel_stats.csv" | head -1) $R/mlp_cr65536_kernel_stats.csv
  for e in ChemicalReactor-v0 PowerGrid-v0 RobotAssembly-v0; do
    run python3 profiles/bench_policy.py --env $e --steps 200 --mlp-steps 20 2> gpurun_out/r05_policy_$e.log | grep '^{' > $R/policy_${e}_final.json
  done
  cat $R/mlp_cr65536_head4_sq.txt; head -12 $R/pg262144_rollout_full_sq.txt; head -12 $R/pg262144_rollout_full_lds_sq.txt
fi
if [ $part = E ]; then
  # PowerGrid with the row-major action ring (the driver line's PowerGrid record since the end of round 5): kernel stats, traffic, SQ
  run python3 bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r05_driver_final3.err | grep '^{' > $R/driver_bench_final3.json
  NIG_PROFILE_EXTRA="--no-step-api --no-brackets --no-single-env" run bash profiles/run_profile.sh r05_pg262144_aos_full --env pg --ring 34 --ring-layout aos --steps 300 --warmup 5 > gpurun_out/r05_prof_pg_aos.log 2>&1 \
    && cp gpurun_out/prof_r05_pg262144_aos_full/r05_pg262144_aos_full_kernel_stats.csv $R/pg262144_rowmajor_ring_full_kernel_stats.csv && cp gpurun_out/prof_r05_pg262144_aos_full/bench.json $R/pg262144_rowmajor_ring_full_bench.json
  NIG_PROFILE_EXTRA="--no-step-api --no-brackets --no-single-env" run bash profiles/run_profile.sh r05_pg262144_rows_full --env pg --ring 34 --steps 300 --warmup 5 > gpurun_out/r05_prof_pg_rows.log 2>&1 \
    && cp gpurun_out/prof_r05_pg262144_rows_full/r05_pg262144_rows_full_kernel_stats.csv $R/pg262144_rows_ring_full_kernel_stats_box3.csv
  run bash profiles/run_pmc.sh r05_pg262144_rollout_full --env pg --ring 34 --ring-layout aos --steps 10 --warmup 2 > gpurun_out/r05_pmc_pg_aos.log 2>&1 \
    && cp gpurun_out/pmc_r05_pg262144_rollout_full/traffic_r05_pg262144_rollout_full.json $R/r05_pg262144_rowmajor_ring_pmc_traffic.json
  cp profiles/traffic.json $R/traffic_merged.json
  run bash profiles/run_sq.sh r05_pg262144_aos_issue --env pg --ring 34 --ring-layout aos --steps 8 --warmup 2 > $R/pg262144_rowmajor_ring_full_sq.txt 2> gpurun_out/r05_sq_pg_aos.err
  head -3 $R/pg262144_rowmajor_ring_full_kernel_stats.csv; head -2 $R/pg262144_rows_ring_full_kernel_stats_box3.csv | tail -1; tail -4 gpurun_out/r05_pmc_pg_aos.log; head -10 $R/pg262144_rowmajor_ring_full_sq.txt
  python3 - <<'PY'
import json
d = json.load(open("gpurun_out/profiles_r05/driver_bench_final3.json"))
r = d["roofline"]; pg = d["powergrid"]
print("headline %.4g, %.1f us, frac %.3f | pg %.3f ms frac %.3f (rows ring %.3f ms frac %.3f) | ra %.3f ms %.3f | mixed %.3f ms %.3f" % (
    d["value"], r["launch_us"], r["frac"], pg["ms_per_step"], pg["roofline"]["frac"], pg["rows_ring"]["ms_per_step"], pg["rows_ring"]["frac"],
    d["robotassembly"]["ms_per_step"], d["robotassembly"]["roofline"]["frac"], d["mixed"]["ms_per_step"], d["mixed"]["roofline"]["frac"]))
PY
fi
if [ $part = D ]; then
  timeout -k 10 1100 bash profiles/sweep.sh r05 > gpurun_out/r05_sweep.log 2>&1
  cp gpurun_out/sweep_r05.jsonl $R/sweep_r05.jsonl; tail -45 gpurun_out/r05_sweep.log > $R/sweep_r05.txt; cat $R/sweep_r05.txt
fi
