# round-5 GPU session 11: cache policy of the 16-byte trajectory stores -- nt (production) vs sc1 (write-through, dropped from L2) vs sc0 sc1 vs nt sc1
mkdir -p gpurun_out
export TMPDIR=/tmp NIG_NO_AUTOBUILD=1
bash profiles/ab.sh "base st1 st2 st3" "pg 262144 full aos" "cr 65536 full aos" "cr 1048576 full aos" > gpurun_out/r05_s11_store_policy.txt 2>&1
cat gpurun_out/r05_s11_store_policy.txt
