#!/bin/bash
# round 6, call C: steady-state timelines of the batch-6400 step with the gather as a launch of its own / inside the forward product
cd "$(dirname "$0")/.."
export CUNVSM_AMD_LIB=$PWD/cunvsm_amd/libcunvsm_amd_dbg.so
NVSM_GATHER_FUSE=0 tools/timeline.sh r06_b6400_unfused --batch=6400 --gate-every 4 > /dev/null 2>&1
NVSM_GATHER_FUSE=3 tools/timeline.sh r06_b6400_fused --batch=6400 --gate-every 4 > /dev/null 2>&1
ls gpurun_out
