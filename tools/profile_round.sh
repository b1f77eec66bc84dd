#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + two PMC passes of the default bench, condensed into
# profiles/<tag>_*.txt by tools/rocprof_summary.py.   usage: tools/profile_round.sh <tag>
set -u
TAG=${1:-r01_x}
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
BENCH="python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- $BENCH > $OUT/bench_stats.json 2> $OUT/bench_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o bench -- $BENCH --steps 6 > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o bench -- $BENCH --steps 6 > $OUT/bench_write.json 2> $OUT/bench_write.err
cd $ROOT
find $OUT -name "*.db" | head
S=$(find $OUT/stats -name "*.db" | head -1); F=$(find $OUT/fetch -name "*.db" | head -1); W=$(find $OUT/write -name "*.db" | head -1)
python tools/rocprof_summary.py stats $S > $OUT/${TAG}_kernel_stats.txt
python tools/rocprof_summary.py pmc $F $W $OUT/${TAG}_hbm_pmc.json > $OUT/${TAG}_hbm_pmc.txt
cp $OUT/bench_stats.json $OUT/${TAG}_bench_under_rocprof.json
rm -f $S $F $W      # the raw databases are large; the summaries are what is kept
head -30 $OUT/${TAG}_kernel_stats.txt; head -16 $OUT/${TAG}_hbm_pmc.txt
