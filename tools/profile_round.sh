#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + three PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA counters — each
# in its own run, with --kernel-trace only) of the default bench, condensed into gpurun_out/prof_<tag>/<tag>_*.txt|json by
# tools/rocprof_summary.py; copy those into profiles/.   usage: tools/profile_round.sh <tag> [extra bench flags]
set -u
TAG=${1:-r02_x}
shift || true
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
BENCH="python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs $*"
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- $BENCH > $OUT/bench_stats.json 2> $OUT/bench_stats.err
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o bench -- $BENCH --steps 6 > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o bench -- $BENCH --steps 6 > $OUT/bench_write.json 2> $OUT/bench_write.err
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace -d $OUT/mfma -o bench -- $BENCH --steps 6 > $OUT/bench_mfma.json 2> $OUT/bench_mfma.err
# the GPU-side schedule of one steady-state step: a spin kernel in front of every step lets the host queue it first
timeout 400 rocprofv3 --kernel-trace -d $OUT/tl -o bench -- $BENCH --steps 8 --gate-us 4000 --no-profile > $OUT/bench_tl.json 2> $OUT/bench_tl.err
cd $ROOT
S=$(find $OUT/stats -name "*.db" | head -1); F=$(find $OUT/fetch -name "*.db" | head -1); W=$(find $OUT/write -name "*.db" | head -1)
M=$(find $OUT/mfma -name "*.db" | head -1); T=$(find $OUT/tl -name "*.db" | head -1)
SIG=$(python -c "import json,sys; print(json.loads(open('$OUT/bench_stats.json').read().strip().splitlines()[-1])['config']['workload_signature'])")
python tools/rocprof_summary.py stats $S > $OUT/${TAG}_kernel_stats.txt
python tools/rocprof_summary.py pmc $F $W $OUT/${TAG}_hbm_pmc.json "$SIG" > $OUT/${TAG}_hbm_pmc.txt
python tools/rocprof_summary.py mfma $M $OUT/${TAG}_mfma_pmc.json > $OUT/${TAG}_mfma_pmc.txt
python tools/rocprof_summary.py timeline $T delay_kernel > $OUT/${TAG}_step_timeline.txt
cp $OUT/bench_stats.json $OUT/${TAG}_bench_under_rocprof.json
rm -f $S $F $W $M $T      # the raw databases are large; the summaries are what is kept
head -30 $OUT/${TAG}_kernel_stats.txt; head -16 $OUT/${TAG}_hbm_pmc.txt; head -12 $OUT/${TAG}_mfma_pmc.txt
