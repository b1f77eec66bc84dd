#!/bin/bash
# counters of the dT product (gemm_dt.hip) ALONE: tools/exp/dt_pmc.sh <rows> <slabs> [out file]
# (several rocprofv3 --pmc passes, kernel trace only, each under its own timeout; appends one line per pass)
ROWS=${1:-51200}; SLABS=${2:-128}
cd "$(dirname "$0")/../.."; ROOT=$PWD; export TMPDIR=/tmp; OUT=$ROOT/gpurun_out/dt_pmc_$SLABS; rm -rf $OUT; mkdir -p $OUT
TXT=${3:-$ROOT/gpurun_out/dt_pmc.txt}
cat > $OUT/run.py <<PY
import ctypes as C, sys
sys.path.insert(0, "$ROOT")
import cunvsm_amd as ca
a, b = C.c_float(), C.c_float()
ca._lib.check(ca.lib().nvsm_debug_dt_time(300, 256, $ROWS, $SLABS, 10, 0, C.byref(a), C.byref(b)))
print(a.value * 1e3, b.value * 1e3)
PY
echo "# gemm_dt_kernel alone: M = 300, N = 256, rows = $ROWS, slabs = $SLABS (nvsm_debug_dt_time, 10 timed + 3 warm-up launches per pass)" >> $TXT
i=0
for CTRS in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
            "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  i=$((i+1)); cd /tmp
  timeout 150 rocprofv3 --pmc $CTRS --kernel-trace -d $OUT/run$i -o g -- python $OUT/run.py > $OUT/out$i.txt 2> $OUT/err$i.txt
  cd $ROOT; DB=$(find $OUT/run$i -name "*.db" | head -1)
  if [ -n "$DB" ]; then python tools/pmc_dump.py $DB gemm_dt >> $TXT; else echo "pass $i failed: $(tail -2 $OUT/err$i.txt)" >> $TXT; fi
  rm -rf $OUT/run$i
done
