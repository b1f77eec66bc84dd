#!/bin/bash
cd "$(dirname "$0")/.."
for sh in "--batch=6400" "--batch=12800" "--config=lse_small" "--batch=51200" "--batch=25600"; do
BENCH_FLAGS="$sh" tools/ab_lib.sh libcunvsm_amd.so libcunvsm_amd_wp.so 2>&1 | sed "s/^/[$sh] /"
done > gpurun_out/r06_p_ab.txt
cat gpurun_out/r06_p_ab.txt
