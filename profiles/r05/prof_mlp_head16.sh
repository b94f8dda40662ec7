# round-5 profile session F: the MFMA actor with the 16 x 16 x 1 head (PowerGrid, 65 536 lanes): MFMA counters (own pass, no tracing),
# kernel stats (own pass), and the policy bench records of the three reference envs on the final build
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export NIG_NO_AUTOBUILD=1
R=profiles/r05
run() { timeout -k 10 400 "$@"; }
run rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/sq_r05_mlp_pg -o r05_mlp_pg -- python3 profiles/bench_policy.py --env PowerGrid-v0 --only-mlp --mlp-launches 8 --mlp-steps 200 > gpurun_out/r05_sq_mlp_pg.json 2> gpurun_out/r05_sq_mlp_pg.log || { tail -5 gpurun_out/r05_sq_mlp_pg.log; exit 1; }
python3 - <<'PY' > gpurun_out/mlp_pg65536_head16_sq.txt 2>&1
import csv, glob
from collections import defaultdict
f = glob.glob("gpurun_out/sq_r05_mlp_pg/**/*counter_collection.csv", recursive=True)
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
cat gpurun_out/mlp_pg65536_head16_sq.txt
run rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05_mlp_pg -o r05_mlp_pg -- python3 profiles/bench_policy.py --env PowerGrid-v0 --only-mlp --mlp-launches 8 --mlp-steps 200 > gpurun_out/mlp_pg65536_bench.json 2> gpurun_out/r05_prof_mlp_pg.log && cp $(find gpurun_out/prof_r05_mlp_pg -name "*kernel_stats.csv" | head -1) gpurun_out/mlp_pg65536_kernel_stats.csv && head -3 gpurun_out/mlp_pg65536_kernel_stats.csv
for e in ChemicalReactor-v0 PowerGrid-v0 RobotAssembly-v0; do
  run python3 profiles/bench_policy.py --env $e --steps 200 --mlp-launches 8 --mlp-steps 200 2> gpurun_out/r05_policy_$e.log | grep '^{' > gpurun_out/policy_${e}_head16.json && python3 -c "
import json; d=json.load(open('gpurun_out/policy_${e}_head16.json')); print('$e', {k: round(v['us_per_step'],2) for k,v in d.items()}, round(d['mlp_mfma_fused']['actor_TFLOPs'],1))"
done
