#!/bin/bash
# an A/B library with extra compile flags: tools/build_variant.sh <suffix> "<flags>" [files...]   -> cunvsm_amd/libcunvsm_amd_<suffix>.so
# (files: only these sources are recompiled with the flags, the rest are the objects of the regular build)
set -e
cd "$(dirname "$0")/../cunvsm_amd/csrc"
SUF=$1; FLAGS=$2; shift 2
FILES=${@:-gather_gemm.hip gemm_panel.hip gemm_tstat.hip gemm_rows.hip gemm_rsplit.hip gemm_split.hip gemm_dt.hip gemm_dtw.hip loss_bn.hip update.hip sort.hip model.cpp tuning.cpp c_api.cpp}
mkdir -p build_$SUF
OBJS=""
for f in gather_gemm.hip gemm_panel.hip gemm_tstat.hip gemm_rows.hip gemm_rsplit.hip gemm_split.hip gemm_dt.hip gemm_dtw.hip loss_bn.hip update.hip sort.hip model.cpp tuning.cpp c_api.cpp; do
  o=${f%.*}.o
  if echo " $FILES " | grep -q " $f "; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $FLAGS -x hip -c $f -o build_$SUF/$o &
    OBJS="$OBJS build_$SUF/$o"
  else
    OBJS="$OBJS build/$o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libcunvsm_amd_$SUF.so $OBJS -ldl -Wl,-rpath,/opt/rocm/lib
rm -rf build_$SUF
echo built ../libcunvsm_amd_$SUF.so
