# is the full-output rollout bound by issuing its stores or by draining them to HBM?  (row stride 0 = same rows every step)
export NIG_NO_AUTOBUILD=1
mkdir -p gpurun_out
for rep in 1 2; do for ow in "" 1; do for sb in 0 256; do
  NIG_DIAG_OVERWRITE=$ow NIG_SPLIT_BLOCKS=$sb timeout -k 10 120 python bench.py --env cr --batch 65536 --outputs full --steps 60 --warmup 10 --no-cpu-baseline --no-parity --no-step-api --no-powergrid 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('overwrite=${ow:-0} split<=$sb: launch_us %.1f' % d['roofline']['launch_us'])"
done; done; done | tee gpurun_out/r02_overwrite_probe.txt
