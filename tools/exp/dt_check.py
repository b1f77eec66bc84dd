"""The dT product on bf16 planes (gemm_dt.hip): correctness against fp64 on ragged shapes, then its time alone by batch size
and slab count next to round 3's kernel (which 1) and the tiled fp32 kernel (which 2)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cunvsm_amd as ca
L = ca.lib()
rs = np.random.RandomState(0)
for (M, N, K, split) in [(300, 256, 6400 + 17, 50), (64, 200, 1000, 3), (320, 132, 77, 1), (16, 256, 64, 4), (300, 256, 51200, 247), (300, 256, 16, 1), (128, 128, 4096, 256)]:
    A = (rs.standard_normal((K, M)) * np.exp2(rs.randint(-10, 3, (K, M)))).astype(np.float32)
    Bm = (rs.standard_normal((K, N)) * np.exp2(rs.randint(-10, 3, (K, N))) + np.arange(N)[None, :] * 1e-3).astype(np.float32)
    out = np.empty((M, N), np.float32)
    ca._lib.check(L.nvsm_debug_gemm((split << 2) | 2, M, N, K, A.ctypes.data, Bm.ctypes.data, out.ctypes.data))
    ref = A.T.astype(np.float64) @ Bm.astype(np.float64)
    scale = np.abs(A.T).astype(np.float64) @ np.abs(Bm).astype(np.float64)
    err = np.abs((out - ref) / scale)
    print("M %d N %d K %d split %d: max err %.3g (at %s) rms %.3g" % (M, N, K, split, err.max(), np.unravel_index(err.argmax(), err.shape), np.sqrt((err ** 2).mean())), flush=True)
if "--time" in sys.argv:
    a, b = C.c_float(), C.c_float()
    for K in (6400, 12800, 51200):
        for which, slabs in ((0, 32), (0, 64), (0, 100), (0, 128), (2, 16)):
            if which == 0 and slabs > K // 32: continue
            ca._lib.check(L.nvsm_debug_dt_time(300, 256, K, slabs, 30, which, C.byref(a), C.byref(b)))
            print("rows %6d which %d slabs %4d: product %7.1f us  reduce %6.1f us" % (K, which, slabs, a.value * 1e3, b.value * 1e3), flush=True)
