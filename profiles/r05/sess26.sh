# round-5 GPU session 26: driver's command after the single-env settle
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_s26_bench.json 2> gpurun_out/r05_s26_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05_s26_bench.json') if l.startswith('{')][-1])
print(d['value'], d['roofline']['launch_us'], d['single_env'])
PY
