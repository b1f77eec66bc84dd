#!/bin/bash
# round 6, call E: bounds + reservation + fill as one launch of a small batch's CSR build — parity, then interleaved A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lazy.py tests/test_gpu_sort.py -x -q -m gpu 2>&1 | tail -8
  timeout 900 python -m pytest tests/test_gpu_switches.py -x -q -m gpu -k "experiment_switches" 2>&1 | tail -8 ) > gpurun_out/r06_e_tests.txt 2>&1
SHAPES="--config=lse_small --batch=6400 --batch=3200" STEPS=300 tools/ab_shapes.sh "NVSM_CSR_FILL_IN_BOUNDS=0" "NVSM_CSR_FILL_IN_BOUNDS=1" > gpurun_out/r06_e_ab.txt 2>&1
cat gpurun_out/r06_e_tests.txt gpurun_out/r06_e_ab.txt
