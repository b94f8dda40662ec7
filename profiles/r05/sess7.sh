# round-5 GPU session 7: where do PowerGrid's outputs land (placement probe); the mixed launch against its bodies AND their stand-alone kernels
mkdir -p gpurun_out
export TMPDIR=/tmp NIG_NO_AUTOBUILD=1
timeout -k 10 300 python profiles/tools/pg_placement.py > gpurun_out/r05_s7_pg_placement.txt 2>&1; tail -25 gpurun_out/r05_s7_pg_placement.txt
mkdir -p gpurun_out/mixed_floor2
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/mixed_floor2 -o mf -- python3 profiles/tools/mixed_floor.py run > gpurun_out/r05_s7_mixed_floor_run.txt 2> gpurun_out/r05_s7_mixed_floor.err || { tail -5 gpurun_out/r05_s7_mixed_floor.err; exit 1; }
f=$(find gpurun_out/mixed_floor2 -name "*counter_collection.csv" | head -1)
python3 profiles/tools/mixed_floor.py table $f > gpurun_out/r05_s7_mixed_floor.txt 2>&1; cat gpurun_out/r05_s7_mixed_floor.txt | grep -v '^{'
