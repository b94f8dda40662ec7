mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/r02_counters_list.txt 2>&1
grep -o "SQ_[A-Z0-9_]*" gpurun_out/r02_counters_list.txt | sort -u | tr '\n' ' ' | head -c 6000
