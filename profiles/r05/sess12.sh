# round-5 GPU session 12: ring-protocol soak under the bounded-wait variant, then the same families on the production library
mkdir -p gpurun_out
export TMPDIR=/tmp NIG_NO_AUTOBUILD=1
NIG_LIB_PATH=$PWD/neorl-industrial-gym_amd/libnig_ringlimit.so timeout -k 10 700 python profiles/tools/ring_soak.py 40 > gpurun_out/r05_s12_ring_soak_bounded.txt 2>&1; echo "bounded rc=$?"; cat gpurun_out/r05_s12_ring_soak_bounded.txt | grep -v '^{'
timeout -k 10 400 python profiles/tools/ring_soak.py 20 > gpurun_out/r05_s12_ring_soak_production.txt 2>&1; echo "production rc=$?"; cat gpurun_out/r05_s12_ring_soak_production.txt | grep -v '^{'
