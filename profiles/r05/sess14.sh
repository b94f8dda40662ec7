# round-5 GPU session 14: the headline's action ring length -- speed (does a 196 MB ring still live in the Infinity Cache?) and the timed
# workload's own statistics against the reference's (a lane's actions repeat every --ring steps); RobotAssembly record with four cycled rings
mkdir -p gpurun_out
export TMPDIR=/tmp NIG_NO_AUTOBUILD=1
for rep in 1 2; do for R in 64 128 250; do
  timeout -k 10 200 python bench.py --gpus 1 --steps 40 --warmup 8 --ring $R --no-cpu-baseline --no-step-api --no-powergrid --no-mixed --no-robotassembly --no-brackets --no-single-env 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']; f=d['parity']['fast_mode_statistics']; t=f['timed_workload']
print('ring $R  launch_us %.1f  frac %.3f | timed workload: viol/ep %.3f (%+.1f sigma)  length %.2f (%+.1f sigma) | reference %.3f' % (r['launch_us'], r['frac'], t['violations_per_episode'], t['violations_per_episode_deviation_sigma'], t['episode_length_mean'], t['episode_length_mean_deviation_sigma'], f['reference']['violations_per_episode']))"
done; done > gpurun_out/r05_s14_ring_length.txt 2>&1
cat gpurun_out/r05_s14_ring_length.txt
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); ra=d['robotassembly']; t=ra['fast_mode_statistics']['timed_workload']
print('RobotAssembly record: %.3f ms frac %.3f | timed workload viol/ep %.3f (%+.1f sigma), reference %.3f' % (ra['ms_per_step'], ra['roofline']['frac'], t['violations_per_episode'], t['violations_per_episode_deviation_sigma'], ra['fast_mode_statistics']['reference']['violations_per_episode']))" >> gpurun_out/r05_s14_ring_length.txt 2>&1
tail -1 gpurun_out/r05_s14_ring_length.txt
