# round-5 GPU session 31: the MFMA actor's LDS-DMA fills spread over the chunk (one piece per 13 MFMAs) against the burst at the
# top of the chunk (libnig_burstfill.so = the same sources with -DNIG_MLP_SPREAD_FILL=0); bit-identity first
set -o pipefail
export NIG_NO_AUTOBUILD=1
timeout -k 10 600 python -m pytest -x -q -m gpu tests/test_gpu_parity.py -k "mfma or mlp" > gpurun_out/r05_s31_tests.log 2>&1; rc=$?; tail -2 gpurun_out/r05_s31_tests.log; [ $rc -eq 0 ] || exit $rc
for r in 1 2; do for v in burstfill base; do
  if [ "$v" = base ]; then unset NIG_LIB_PATH; else export NIG_LIB_PATH=$PWD/neorl-industrial-gym_amd/libnig_$v.so; fi
  for e in ChemicalReactor-v0 PowerGrid-v0 RobotAssembly-v0 HVACControl-v0; do
    timeout -k 10 200 python3 profiles/bench_policy.py --env $e --only-mlp --mlp-launches 8 --mlp-steps 200 2>/dev/null | grep '^{' | python3 -c "
import json,sys; m=json.loads(sys.stdin.read())['mlp_mfma_fused']; print('$v $e: %.2f us per step  %.1f TFLOP/s' % (m['us_per_step'], m['actor_TFLOPs']))"
  done
done; done > gpurun_out/r05_s31_mlp_fill_ab.txt 2>&1
cat gpurun_out/r05_s31_mlp_fill_ab.txt
