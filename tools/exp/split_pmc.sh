#!/bin/bash
# counters of the split-bf16 GEMM alone: tools/exp/split_pmc.sh <tag> "<counters>" (one rocprofv3 --pmc pass, kernel trace only)
TAG=$1; CTRS=$2
cd "$(dirname "$0")/../.."; ROOT=$PWD; export TMPDIR=/tmp; OUT=$ROOT/gpurun_out/spmc_$TAG; mkdir -p $OUT; cd /tmp
rocprofv3 --pmc $CTRS --kernel-trace -d $OUT/run -o g -- python $ROOT/tools/exp/gemm_time.py 51200 > $OUT/out.txt 2> $OUT/err.txt
cd $ROOT; DB=$(find $OUT/run -name "*.db" | head -1); python tools/pmc_dump.py $DB gemm_split > gpurun_out/spmc_$TAG.txt; rm -rf $OUT/run; cat gpurun_out/spmc_$TAG.txt
