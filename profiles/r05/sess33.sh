# round-5 GPU session 33: the MFMA actor with parts removed (diagnostic builds, WRONG results): no env step / no LDS-DMA fills
# after the first step / neither -- which part of a step do the partner wave's MFMAs fail to cover?
export NIG_NO_AUTOBUILD=1
for r in 1 2; do for v in base noenv nofill noenvfill; do
  if [ "$v" = base ]; then unset NIG_LIB_PATH; else export NIG_LIB_PATH=$PWD/neorl-industrial-gym_amd/libnig_$v.so; fi
  for e in ChemicalReactor-v0 PowerGrid-v0; do
    timeout -k 10 200 python3 profiles/bench_policy.py --env $e --only-mlp --mlp-launches 8 --mlp-steps 200 2>/dev/null | grep '^{' | python3 -c "
import json,sys; m=json.loads(sys.stdin.read())['mlp_mfma_fused']; print('$v $e: %.2f us per step  %.1f TFLOP/s' % (m['us_per_step'], m['actor_TFLOPs']))"
  done
done; done > gpurun_out/r05_s33_mlp_parts.txt 2>&1
cat gpurun_out/r05_s33_mlp_parts.txt
