#!/bin/bash
# round 6, call B: is the dT product on the per-rank steps' critical cycle, and does the gather inside the product pay once it is not?
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
SHAPES="--batch=6400 --config=lse_small --batch=12800" STEPS=200 tools/ab_shapes.sh "NVSM_GATHER_FUSE=0" "NVSM_GATHER_FUSE=3" "NVSM_GATHER_FUSE=0 NVSM_SKIP_DT=1" "NVSM_GATHER_FUSE=3 NVSM_SKIP_DT=1" > gpurun_out/r06_b_ab.txt 2>&1
cat gpurun_out/r06_b_ab.txt
