#!/bin/bash
# round 6, call F: the wave-sized dT kernel — unit parity, model parity at per-rank batches, interleaved A/B against the tiled kernel
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm_dtw or forward_backward or fused_step or update_parity" 2>&1 | tail -8
  timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "per_rank" 2>&1 | tail -8
  timeout 600 python tools/exp/lse_tol.py 2>&1 | tail -5 ) > gpurun_out/r06_f_tests.txt 2>&1
python - > gpurun_out/r06_f_alone.txt 2>&1 <<'PY'
import ctypes as C, sys
sys.path.insert(0, ".")
import cunvsm_amd as ca
L = ca.lib()
for (M, N, K) in ((300, 256, 6400), (128, 256, 4096), (300, 256, 12800)):
    for which, slabs in ((2, 16), (0, 100)):
        a, b = C.c_float(), C.c_float()
        ca._lib.check(L.nvsm_debug_dt_time(M, N, K, slabs, 50, which, C.byref(a), C.byref(b)))
        print("alone", (M, N, K), "tiled fp32" if which == 2 else "gemm_dt", slabs, "slabs: %.1f us + reduce %.1f us" % (a.value * 1e3, b.value * 1e3))
PY
SHAPES="--batch=6400 --config=lse_small --batch=12800 --batch=3200" STEPS=300 tools/ab_shapes.sh "NVSM_DTW_MAX_B=0" "NVSM_DTW_MAX_B=16383" "NVSM_DTW_MAX_B=16383 NVSM_DTW_SLABS=32" "NVSM_DTW_MAX_B=16383 NVSM_DTW_SLABS=8" > gpurun_out/r06_f_ab.txt 2>&1
cat gpurun_out/r06_f_tests.txt gpurun_out/r06_f_alone.txt gpurun_out/r06_f_ab.txt
