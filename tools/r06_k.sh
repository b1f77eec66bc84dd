#!/bin/bash
# the dT product of a 25 600-window rank that cannot take the main-stream gemm_dt (data parallel / lazy tables: NVSM_DT_ON_MAIN=0 stands in): tiled fp32 vs wave-sized
cd "$(dirname "$0")/.."
SHAPES="--batch=25600 --batch=20480" STEPS=200 tools/ab_shapes.sh "NVSM_DT_ON_MAIN=0 NVSM_DTW_MAX_B=16383" "NVSM_DT_ON_MAIN=0 NVSM_DTW_MAX_B=40959" "NVSM_DT_ON_MAIN=0 NVSM_DTW_MAX_B=40959 NVSM_DTW_SLABS=24" "NVSM_DT_ON_MAIN=0 NVSM_DTW_MAX_B=40959 NVSM_DTW_SLABS=48" > gpurun_out/r06_k_ab.txt 2>&1
cat gpurun_out/r06_k_ab.txt
