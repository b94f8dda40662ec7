# how much of the SIMD does one wave use?  ChemicalReactor fused rollout at 1, 2, 3, 4 waves per SIMD
mkdir -p gpurun_out
export NIG_NO_AUTOBUILD=1
for b in 65536 131072 196608 262144; do for o in full none; do python bench.py --env cr --batch $b --outputs $o --steps 60 --warmup 10 --no-cpu-baseline --no-parity --no-step-api --no-powergrid 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('cr $b $o: launch_us %.1f  value %.3e frac %.3f' % (d['roofline']['launch_us'], d['value'], d['roofline']['frac']))"; done; done | tee gpurun_out/r02_scale_probe.txt
