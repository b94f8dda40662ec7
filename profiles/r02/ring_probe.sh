# does the action ring's size (cache-resident or not) move the full-output rollout?  (reads competing with the output drain)
export NIG_NO_AUTOBUILD=1
mkdir -p gpurun_out
for rep in 1 2; do for r in 2 16 64 250; do
  timeout -k 10 120 python bench.py --env cr --batch 65536 --outputs full --ring $r --steps 60 --warmup 10 --no-cpu-baseline --no-parity --no-step-api --no-powergrid 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('ring $r: launch_us %.1f frac %.3f' % (d['roofline']['launch_us'], d['roofline']['frac']))"
done; done | tee gpurun_out/r02_ring_probe.txt
