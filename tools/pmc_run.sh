#!/bin/bash
# one rocprofv3 --pmc pass (--kernel-trace only) of a short bench run: tools/pmc_run.sh <tag> "<counters>" [bench flags]
TAG=$1; CTRS=$2; shift 2
cd "$(dirname "$0")/.."; ROOT=$PWD; export TMPDIR=/tmp; OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT; cd /tmp
rocprofv3 --pmc $CTRS --kernel-trace -d $OUT/run -o bench -- python $ROOT/bench.py --steps 6 --warmup 3 --repeats 1 --no-cpu-baseline --no-extra-legs --no-profile "$@" > $OUT/bench.json 2> $OUT/bench.err
cd $ROOT; DB=$(find $OUT/run -name "*.db" | head -1); python tools/pmc_dump.py $DB > gpurun_out/pmc_$TAG.txt; rm -rf $OUT/run; cat gpurun_out/pmc_$TAG.txt | head -30
