"""Where a turn of the split-bf16 dT kernel goes (make -C cunvsm_amd/csrc dbg; CUNVSM_AMD_LIB=.../libcunvsm_amd_dbg.so): shader-clock
stamps of waves 0 and 4 (one SIMD) of workgroup 0."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cunvsm_amd as ca
K, split = 51200, int(sys.argv[1]) if len(sys.argv) > 1 else 64
rs = np.random.RandomState(0)
A = rs.standard_normal((K, 300)).astype(np.float32); B = rs.standard_normal((K, 256)).astype(np.float32)
out = np.empty((300, 256), np.float32)
L = ca.lib()
ca._lib.check(L.nvsm_debug_gemm((split << 2) | 2, 300, 256, K, A.ctypes.data, B.ctypes.data, out.ctypes.data))
t = np.zeros(256, np.uint64)
L.nvsm_debug_dt_ticks.argtypes = [C.c_void_p, C.c_int]
assert L.nvsm_debug_dt_ticks(t.ctypes.data, t.size) == 0
tk = t.astype(np.int64).reshape(2, 16, 8)
print("cycles since the wave's first stamp: top | k step 1 | MFMAs issued | barrier 1 passed | pieces written | registers moved | barrier 2 passed")
for wv in range(2):
    for kt in range(1, 6):
        r = tk[wv, kt] - tk[wv, 1, 0]
        print("wave %d tile %d: " % (wv * 4, kt) + " ".join("%7d" % x for x in r[:7]))
