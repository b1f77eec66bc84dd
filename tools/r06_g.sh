#!/bin/bash
# round 6, call G: the wave-sized dT kernel alone, its slab count in the step, and the steady-state timeline at batch 6400
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python - > gpurun_out/r06_g_alone.txt 2>&1 <<'PY'
import ctypes as C, sys
sys.path.insert(0, ".")
import cunvsm_amd as ca
L = ca.lib()
for (M, N, K) in ((300, 256, 6400), (128, 256, 4096), (300, 256, 12800), (300, 256, 3200)):
    for which, slabs in ((2, 16), (0, 100), (1, 4), (1, 8), (1, 16), (1, 32), (1, 64)):
        a, b = C.c_float(), C.c_float()
        ca._lib.check(L.nvsm_debug_dt_time(M, N, K, slabs, 50, which, C.byref(a), C.byref(b)))
        print("alone", (M, N, K), {2: "tiled fp32", 0: "gemm_dt", 1: "gemm_dtw"}[which], slabs, "slabs: %.1f us + reduce %.1f us" % (a.value * 1e3, b.value * 1e3))
PY
SHAPES="--batch=6400 --config=lse_small --batch=12800 --batch=3200" STEPS=300 tools/ab_shapes.sh "NVSM_DTW_SLABS=4" "NVSM_DTW_SLABS=6" "NVSM_DTW_SLABS=8" "NVSM_DTW_SLABS=12" "NVSM_SKIP_DT=1" > gpurun_out/r06_g_ab.txt 2>&1
export CUNVSM_AMD_LIB=$PWD/cunvsm_amd/libcunvsm_amd_dbg.so
NVSM_DTW_SLABS=8 tools/timeline.sh r06_b6400_dtw --batch=6400 --gate-every 4 > /dev/null 2>&1
cat gpurun_out/r06_g_alone.txt gpurun_out/r06_g_ab.txt
