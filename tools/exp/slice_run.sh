#!/bin/bash
# tools/exp/slice_exp.hip on the GPU box: times of the three variants, then one counter pass (fabric requests, L2 hits) over each
cd "$(dirname "$0")/../.."; ROOT=$PWD; export TMPDIR=/tmp; OUT=$ROOT/gpurun_out/slice; rm -rf $OUT; mkdir -p $OUT
EXE=$ROOT/tools/exp/slice_exp.out
$EXE -1 "$@" | tee $OUT/times.txt
for V in 0 1 2; do
  cd /tmp
  timeout 60 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/run$V -o g -- $EXE $V "$@" > $OUT/out$V.txt 2> $OUT/err$V.txt
  cd $ROOT; DB=$(find $OUT/run$V -name "*.db" | head -1)
  if [ -n "$DB" ]; then python tools/pmc_dump.py $DB _kernel | tee -a $OUT/pmc.txt; else echo "pass $V failed: $(tail -2 $OUT/err$V.txt)" | tee -a $OUT/pmc.txt; fi
  rm -rf $OUT/run$V
done
