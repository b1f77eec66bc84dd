#!/bin/bash
# as tools/timeline.sh but free-running (no gate kernel): one steady-state step from one prologue to the next
set -u
TAG=${1:-tlf}; shift || true
cd "$(dirname "$0")/.."; ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/tl_$TAG; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace -d $OUT/tl -o bench -- python $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-legs --no-profile "$@" > $OUT/bench.json 2> $OUT/bench.err
cd $ROOT
T=$(find $OUT/tl -name "*.db" | head -1)
python tools/rocprof_summary.py timeline $T step_prologue_kernel > gpurun_out/timeline_$TAG.txt
rm -rf $OUT/tl
cat gpurun_out/timeline_$TAG.txt
