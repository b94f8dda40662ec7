# usage: bash profiles/r02/split_ab.sh "<variants>"   -- tests with the LAST variant as libnig.so, then A/B of all
export NIG_NO_AUTOBUILD=1
mkdir -p gpurun_out
last=$(echo $1 | awk '{print $NF}')
cp neorl-industrial-gym_amd/libnig.so /tmp/libnig_orig.so
trap 'cp /tmp/libnig_orig.so neorl-industrial-gym_amd/libnig.so' EXIT   # (round-2 script, kept as the record of that A/B: it swaps libnig.so in place; new A/Bs load variants through NIG_LIB_PATH, profiles/ab.sh)
cp neorl-industrial-gym_amd/libnig_$last.so neorl-industrial-gym_amd/libnig.so
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_abi_round2.py tests/test_spec_envs.py -m gpu -x -q > gpurun_out/split_tests.txt 2>&1 || { tail -30 gpurun_out/split_tests.txt; exit 1; }
tail -2 gpurun_out/split_tests.txt
cp /tmp/libnig_orig.so neorl-industrial-gym_amd/libnig.so
bash profiles/ab.sh "$1" "cr 65536 none aos" "cr 65536 full aos" "cr 65536 min aos" "cr 16384 full aos" "cr 32768 full aos"
