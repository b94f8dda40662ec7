# round-5 GPU session 30: what do the MFMA actor's nine chunk barriers per step cost?  Upper bound: a diagnostic build without
# them (WRONG results: waves read chunks that are still being filled) against the production library, same box
export NIG_NO_AUTOBUILD=1
for r in 1 2; do for v in base mlpnobar; do
  if [ "$v" = base ]; then unset NIG_LIB_PATH; else export NIG_LIB_PATH=$PWD/neorl-industrial-gym_amd/libnig_$v.so; fi
  for e in ChemicalReactor-v0 PowerGrid-v0; do
    timeout -k 10 200 python3 profiles/bench_policy.py --env $e --only-mlp --mlp-launches 8 --mlp-steps 200 2>/dev/null | grep '^{' | python3 -c "
import json,sys; m=json.loads(sys.stdin.read())['mlp_mfma_fused']; print('$v $e: %.2f us per step  %.1f TFLOP/s' % (m['us_per_step'], m['actor_TFLOPs']))"
  done
done; done > gpurun_out/r05_s30_mlp_barrier_bound.txt 2>&1
cat gpurun_out/r05_s30_mlp_barrier_bound.txt
