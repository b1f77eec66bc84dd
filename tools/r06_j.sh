#!/bin/bash
cd "$(dirname "$0")/.."
SHAPES="--config=lse_small" STEPS=400 tools/ab_shapes.sh "NVSM_JOIN_T_EARLY=0" "NVSM_JOIN_T_EARLY=1" "NVSM_JOIN_T_EARLY=0" "NVSM_JOIN_T_EARLY=1" > gpurun_out/r06_j_ab.txt 2>&1
cat gpurun_out/r06_j_ab.txt
