#!/bin/bash
# round 6, call A: the gather inside the per-rank forward product — parity, then interleaved A/B against the separate launch
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gather or forward_backward or fused_step" 2>&1 | tail -15
  python -m pytest tests/test_gpu_switches.py -x -q -m gpu -k "experiment_switches" 2>&1 | tail -15 ) > gpurun_out/r06_a_tests.txt 2>&1
SHAPES="--batch=6400 --config=lse_small --batch=3200 --batch=8192" STEPS=200 tools/ab_shapes.sh "NVSM_GATHER_FUSE=0" "NVSM_GATHER_FUSE=3" > gpurun_out/r06_a_ab.txt 2>&1
cat gpurun_out/r06_a_tests.txt gpurun_out/r06_a_ab.txt
