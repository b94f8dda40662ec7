# round-5 GPU session 18: same-box A/B of the actor's head -- libnig_head32.so (the 32 x 32 x 2 head of the previous commit, built
# from a stash) against the 16 x 16 x 1 head, 8 launches x 200 steps at 65 536 lanes; then the spec-env tests (SupplyChain's actor)
set -o pipefail
export NIG_NO_AUTOBUILD=1
run() { t=$1; shift; timeout -k 10 $t "$@"; }
for r in 1 2; do for v in head32 base; do
  if [ "$v" = base ]; then unset NIG_LIB_PATH; else export NIG_LIB_PATH=$PWD/neorl-industrial-gym_amd/libnig_$v.so; fi
  for e in PowerGrid-v0 RobotAssembly-v0 AdvancedChemicalReactor-v0 HVACControl-v0; do
    run 200 python3 profiles/bench_policy.py --env $e --only-mlp --mlp-launches 8 --mlp-steps 200 2>/dev/null | grep '^{' | python3 -c "
import json,sys; m=json.loads(sys.stdin.read())['mlp_mfma_fused']; print('$v $e: %.2f us per step  %.1f TFLOP/s' % (m['us_per_step'], m['actor_TFLOPs']))"
  done
done; done > gpurun_out/r05_s18_head_ab.txt 2>&1
unset NIG_LIB_PATH
cat gpurun_out/r05_s18_head_ab.txt
run 900 python -m pytest -x -q -m gpu tests/test_spec_envs.py > gpurun_out/r05_s18_spec_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r05_s18_spec_tests.log; exit $rc
