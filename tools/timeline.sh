#!/bin/bash
# GPU-side schedule of one steady-state step (rocprofv3 kernel trace of bench.py --gate-us): tools/timeline.sh <tag> [bench flags]
set -u
TAG=${1:-tl}; shift || true
cd "$(dirname "$0")/.."; ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/tl_$TAG; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace -d $OUT/tl -o bench -- python $ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra-legs --gate-us 4000 --no-profile "$@" > $OUT/bench.json 2> $OUT/bench.err
cd $ROOT
T=$(find $OUT/tl -name "*.db" | head -1)
python tools/rocprof_summary.py timeline $T delay_kernel > gpurun_out/timeline_$TAG.txt
rm -rf $OUT/tl
cat gpurun_out/timeline_$TAG.txt
