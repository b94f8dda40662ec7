# the driver's round-end sequence on the final build: pytest -m gpu, smoke(), bench.py
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r05_final_gpu_tests.log 2>&1; rc=$?; echo "gpu suite rc=$rc"; tail -3 gpurun_out/r05_final_gpu_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_final_bench.json 2> gpurun_out/r05_final_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05_final_bench.json') if l.startswith('{')][-1])
r=d['roofline']
print("headline %.4g, %.1f us, frac %.3f alg %.3f | pg %.3f ms %.3f | ra %.3f ms %.3f | mixed %.3f ms %.3f | step %.2f us | single %.0f/s | stats ok %s" % (
  d['value'], r['launch_us'], r['frac'], r['frac_algorithmic'], d['powergrid']['ms_per_step'], d['powergrid']['roofline']['frac'],
  d['robotassembly']['ms_per_step'], d['robotassembly']['roofline']['frac'], d['mixed']['ms_per_step'], d['mixed']['roofline']['frac'],
  d['step_api']['launch_us'], d['single_env']['value'], d['parity']['fast_mode_statistics']['within_4_sigma']))
PY
