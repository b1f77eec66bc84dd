#!/bin/bash
cd "$(dirname "$0")/.."
SHAPES="--config=lse_small" STEPS=300 tools/ab_shapes.sh "NVSM_LOSS_EPW=0" "NVSM_LOSS_EPW=1" "NVSM_LOSS_EPW=2" "NVSM_LOSS_EPW=3" "NVSM_LOSS_EPW=4" > gpurun_out/r06_u_ab.txt 2>&1
SHAPES="--batch=6400" STEPS=300 tools/ab_shapes.sh "NVSM_LOSS_EPW=0" "NVSM_LOSS_EPW=2" "NVSM_LOSS_EPW=3" "NVSM_LOSS_EPW=5" >> gpurun_out/r06_u_ab.txt 2>&1
cat gpurun_out/r06_u_ab.txt
