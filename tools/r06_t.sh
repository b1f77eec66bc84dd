#!/bin/bash
cd "$(dirname "$0")/.."
SHAPES="--config=lse_small --batch=6400" STEPS=300 tools/ab_shapes.sh "NVSM_SKIP_WAITS=0" "NVSM_SKIP_WAITS=1" "NVSM_SKIP_WAITS=0 NVSM_STOP_EVENTS=0" > gpurun_out/r06_t_ab.txt 2>&1
cat gpurun_out/r06_t_ab.txt
