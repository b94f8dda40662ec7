# round-4 profile session, part F (GPU box, repo root): PowerGrid's LDS counters on the final build (the "second resource at its
# limit" of DESIGN section 5), reward + flags and full outputs, and the sweep of the measurement table on the final build.
mkdir -p gpurun_out profiles/r04
export TMPDIR=/tmp NIG_PROFILE_ROUND=r04
python3 -c 'import __graft_entry__ as g; g.build(force=False)' > /dev/null
export NIG_NO_AUTOBUILD=1
R=profiles/r04
run() { echo "== $*" >&2; timeout -k 10 240 "$@"; echo "   rc=$?" >&2; }
for o in min full; do
  NIG_SQ_COUNTERS="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" run bash profiles/run_sq.sh r04_pg262144_${o}_lds --env pg --outputs $o --steps 8 --warmup 2 > $R/pg262144_rollout_${o}_lds_sq.txt 2> gpurun_out/r04_sq_pg_${o}_lds.err
  run bash profiles/run_sq.sh r04_pg262144_${o}_issue --env pg --outputs $o --steps 8 --warmup 2 > $R/pg262144_rollout_${o}_final_sq.txt 2> gpurun_out/r04_sq_pg_${o}_issue.err
done
run bash profiles/sweep.sh r04_final > gpurun_out/r04_sweep_final.log 2>&1
cp gpurun_out/sweep_r04_final.jsonl $R/sweep_r04_final.jsonl; tail -40 gpurun_out/r04_sweep_final.log > $R/sweep_r04_final.txt
mkdir -p gpurun_out/profiles_r04 && cp $R/pg262144_rollout_*_lds_sq.txt $R/pg262144_rollout_*_final_sq.txt $R/sweep_r04_final.* gpurun_out/profiles_r04/
cat $R/pg262144_rollout_min_lds_sq.txt $R/pg262144_rollout_full_lds_sq.txt; cat $R/sweep_r04_final.txt
