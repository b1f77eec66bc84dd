#!/bin/bash
# durations of the dT kernels alone (rocprofv3 kernel trace of tools/exp/dt_time.py: launches in the order of its loops)
cd "$(dirname "$0")/../.."; ROOT=$PWD; export TMPDIR=/tmp; OUT=$ROOT/gpurun_out/dt_alone; rm -rf $OUT; mkdir -p $OUT; cd /tmp
rocprofv3 --kernel-trace -d $OUT/run -o t -- python $ROOT/tools/exp/dt_time.py > $OUT/out.txt 2>&1
cd $ROOT; python - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob('gpurun_out/dt_alone/run/*.db')[0])
for name, dur in db.execute("select name, duration from kernels order by start"):
    if 'gemm' in name or 'splitk' in name: print("%-60s %8.1f us" % (name[:60], dur / 1e3))
PY
