"""Error of the projection products against an fp64 product, for the exact-fp32 MFMA kernels (NVSM_GEMM_SPLIT=0) and the
split-bf16 kernel with nine / six partial products (gemm_split.hip), and their times alone on an idle GPU:
   python tools/exp/gemm_accuracy.py [M]
Errors are given relative to Σ_k |a_k b_k| (the scale roundoff lives on): max over the outputs and root mean square."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cunvsm_amd as ca

M = int(sys.argv[1]) if len(sys.argv) > 1 else 51200
rs = np.random.RandomState(5)
for name, bl, N, K, extras in (("forward", 0, 256, 300, 1), ("backward", 1, 300, 256, 2)):
    for dist in ("normal", "wide"):
        A = rs.standard_normal((M, K)).astype(np.float32)
        B = (rs.standard_normal((K, N)) * 0.1).astype(np.float32)
        if dist == "wide":      # magnitudes over many binades, as gradients have
            A *= np.exp2(rs.randint(-12, 4, A.shape)).astype(np.float32)
            B *= np.exp2(rs.randint(-6, 3, B.shape)).astype(np.float32)
        ref = A.astype(np.float64) @ B.astype(np.float64)
        scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
        Bdev = np.ascontiguousarray(B.T) if bl else B
        for mode in ("0", "9", "6"):
            os.environ["NVSM_GEMM_SPLIT"] = mode
            out = np.empty((M, N), np.float32)
            ca._lib.check(ca.lib().nvsm_debug_gemm(bl, M, N, K, A.ctypes.data, Bdev.ctypes.data, out.ctypes.data))
            err = (out.astype(np.float64) - ref) / scale
            ms = C.c_float()
            ca._lib.check(ca.lib().nvsm_debug_gemm_time(bl, M, N, K, extras, 50, C.byref(ms)))
            ms0 = C.c_float()
            ca._lib.check(ca.lib().nvsm_debug_gemm_time(bl, M, N, K, 0, 50, C.byref(ms0)))
            print("%-8s %-6s split=%s  max %.3e  rms %.3e  mean %+.3e | %.1f us with statistics, %.1f us plain"
                  % (name, dist, mode, np.abs(err).max(), np.sqrt((err ** 2).mean()), err.mean(), ms.value * 1e3, ms0.value * 1e3), flush=True)
