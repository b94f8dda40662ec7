# round-5 GPU session 22: the driver's command with the new stream_probe bracket
set -o pipefail
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_s22_bench.json 2> gpurun_out/r05_s22_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05_s22_bench.json') if l.startswith('{')][-1])
r=d['roofline']
print(r['launch_us'], r['frac'], r.get('stream_probe'))
PY
