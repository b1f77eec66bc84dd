#!/bin/bash
# per-kernel durations of a bench run under rocprofv3 --kernel-trace --stats: tools/kstats.sh <tag> [bench flags]
set -u
TAG=${1:-ks}; shift || true
cd "$(dirname "$0")/.."; ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/ks_$TAG; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/st -o bench -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs --no-profile "$@" > $OUT/bench.json 2> $OUT/bench.err
cd $ROOT
S=$(find $OUT/st -name "*.db" | head -1)
python tools/rocprof_summary.py stats $S > gpurun_out/kstats_$TAG.txt
rm -rf $OUT/st
head -40 gpurun_out/kstats_$TAG.txt
