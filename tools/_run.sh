ROUNDS=2 tools/ab.sh "NVSM_NT=0" "NVSM_NT=1" "NVSM_NT=3" "NVSM_NT=5" "NVSM_NT=9" "NVSM_NT=13" "NVSM_NT=15" > gpurun_out/s8_ab.txt 2>&1
cat gpurun_out/s8_ab.txt
