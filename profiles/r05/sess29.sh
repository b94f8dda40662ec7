# round-5 GPU session 29: bench.py with NO flags (the contract's default: N = 1, K / W that finish within minutes)
( time timeout -k 10 600 python bench.py > gpurun_out/r05_s29_bench_default.json 2> gpurun_out/r05_s29_bench_default.err ) 2>&1 | tail -3
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05_s29_bench_default.json') if l.startswith('{')][-1])
print(d['n_gpus'], d['steps'], d['warmup'], d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
