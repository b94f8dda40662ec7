# round-5 GPU session 17: the actor's head on v_mfma_f32_16x16x1 (four blocks, K = 1) for envs with five to sixteen actions --
# bit-identity vs the oracle (every env incl. SupplyChain, new), then the closed-loop rates for PowerGrid / RobotAssembly
set -o pipefail
run() { t=$1; shift; timeout -k 10 $t "$@"; }
mkdir -p gpurun_out
run 900 python -m pytest -x -q -m gpu tests/test_gpu_parity.py -k "mfma or mlp" tests/test_spec_envs.py > gpurun_out/r05_s17_tests.log 2>&1; rc=$?; tail -5 gpurun_out/r05_s17_tests.log; [ $rc -eq 0 ] || exit $rc
for e in PowerGrid-v0 RobotAssembly-v0 ChemicalReactor-v0; do
  run 300 python3 profiles/bench_policy.py --env $e --only-mlp --mlp-launches 8 --mlp-steps 200 2> gpurun_out/r05_s17_policy_$e.log | grep '^{' > gpurun_out/r05_s17_policy_$e.json && python3 -c "
import json,sys; d=json.load(open('gpurun_out/r05_s17_policy_$e.json')); m=d['mlp_mfma_fused']; print('$e', round(m['us_per_step'],2), 'us per step', round(m['actor_TFLOPs'],1), 'TFLOP/s')"
done
