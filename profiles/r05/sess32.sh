# round-5 GPU session 32: MFMA pacing microbenchmark (profiles/ubench/mfma_pace.hip)
timeout -k 10 120 profiles/ubench/mfma_pace > gpurun_out/r05_s32_mfma_pace.txt 2>&1; echo rc=$?
cat gpurun_out/r05_s32_mfma_pace.txt
