# round-4 profile session, part C (GPU box, repo root): after the late-round changes -- the MFMA actor at two blocks per CU
# (duty cycle again), PowerGrid with the six-block reset (kernel stats), the driver's command un-profiled.
mkdir -p gpurun_out profiles/r04
export TMPDIR=/tmp NIG_PROFILE_ROUND=r04
python3 -c 'import __graft_entry__ as g; g.build(force=False)' > /dev/null
export NIG_NO_AUTOBUILD=1
R=profiles/r04
run() { echo "== $*" >&2; timeout -k 10 200 "$@"; echo "   rc=$?" >&2; }
run rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/sq_r04_mlp2 -o r04_mlp2 -- python3 profiles/bench_policy.py --only-mlp --mlp-launches 8 --mlp-steps 200 > gpurun_out/r04_sq_mlp2.json 2> gpurun_out/r04_sq_mlp2.log
python3 - <<'PY' > $R/mlp_cr65536_two_blocks_sq.txt 2>&1
import csv, glob
from collections import defaultdict
f = glob.glob("gpurun_out/sq_r04_mlp2/**/*counter_collection.csv", recursive=True)
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
NIG_PROFILE_EXTRA="--no-step-api --no-brackets --no-single-env --no-mixed" run bash profiles/run_profile.sh r04_pg262144_v3 --env pg --steps 10 --warmup 2 > gpurun_out/r04_prof_pg_v3.log 2>&1 && cp gpurun_out/prof_r04_pg262144_v3/r04_pg262144_v3_kernel_stats.csv $R/pg262144_rollout_full_v3_kernel_stats.csv
run python3 bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r04_driver_unprofiled2.err | grep '^{' > $R/driver_bench_unprofiled.json
for e in ChemicalReactor-v0 PowerGrid-v0 RobotAssembly-v0; do
  run python3 profiles/bench_policy.py --env $e --steps 200 --mlp-steps 20 2> gpurun_out/r04_policy_$e.log | grep '^{' > $R/policy_$e.json
done
mkdir -p gpurun_out/profiles_r04 && cp -r $R/* gpurun_out/profiles_r04/
cat $R/mlp_cr65536_two_blocks_sq.txt; head -3 $R/pg262144_rollout_full_v3_kernel_stats.csv
python3 - <<'PY'
import json
d = json.load(open("profiles/r04/driver_bench_unprofiled.json"))
r = d["roofline"]
print("headline %.4g env-steps/s, %.1f us, frac %.3f (alg %.3f); pg %.3f ms frac %.3f; mixed %.3f ms frac %.3f; step %.2f us; single %.0f/s" % (
    d["value"], r["launch_us"], r["frac"], r["frac_algorithmic"], d["powergrid"]["ms_per_step"], d["powergrid"]["roofline"]["frac"],
    d["mixed"]["ms_per_step"], d["mixed"]["roofline"]["frac"], d["step_api"]["launch_us"], d["single_env"]["value"]))
PY
