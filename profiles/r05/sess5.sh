# round-5 GPU session 5: store data from AGPRs (ubench), the headline's recorder as a pure store wave
mkdir -p gpurun_out
export TMPDIR=/tmp NIG_NO_AUTOBUILD=1
./profiles/ubench/store_src > gpurun_out/r05_s5_store_src.txt 2>&1; cat gpurun_out/r05_s5_store_src.txt
./profiles/ubench/store_src >> gpurun_out/r05_s5_store_src.txt 2>&1; tail -5 gpurun_out/r05_s5_store_src.txt
NIG_DIAG_NO_TALLY_CHECK=1 bash profiles/ab.sh "base recnocomp" "cr 65536 full aos" "cr 65536 min aos" > gpurun_out/r05_s5_cr_ab.txt 2>&1
cat gpurun_out/r05_s5_cr_ab.txt
