#!/bin/bash
# FETCH_SIZE per kernel of a bench run (one --pmc pass): tools/fetch_pmc.sh <tag> [bench flags]; env vars pass through
set -u
TAG=${1:-f}; shift || true
cd "$(dirname "$0")/.."; ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/fetch_$TAG; mkdir -p $OUT
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o bench -- python $ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extra-legs "$@" > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o bench -- python $ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extra-legs "$@" > $OUT/bench_w.json 2> $OUT/bench_w.err
cd $ROOT
F=$(find $OUT/fetch -name "*.db" | head -1); W=$(find $OUT/write -name "*.db" | head -1)
python tools/rocprof_summary.py pmc $F $W $OUT/pmc.json x > gpurun_out/fetch_$TAG.txt
rm -rf $OUT/fetch $OUT/write
head -12 gpurun_out/fetch_$TAG.txt | cut -c1-160
