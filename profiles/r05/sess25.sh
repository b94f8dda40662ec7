# round-5 GPU session 25: where the single-env class's first-measurement 60 us per step comes from
export NIG_NO_AUTOBUILD=1
timeout -k 10 200 python profiles/tools/single_env_warmup.py > gpurun_out/r05_s25_single_env_warmup.txt 2>&1
cat gpurun_out/r05_s25_single_env_warmup.txt
