mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
bash profiles/run_sq.sh r02_pg_coop1 --env pg --batch 262144 --outputs full --steps 12 --warmup 2 --no-powergrid 2>&1 | tail -15
