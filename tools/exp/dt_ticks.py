"""Where a turn of the dT kernel on planes goes (make -C cunvsm_amd/csrc dbg; CUNVSM_AMD_LIB=.../libcunvsm_amd_dbg.so): shader-clock
stamps of waves 0 and 4 (one SIMD), 1 and 7 of workgroups 0 and 100."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cunvsm_amd as ca
K, split = 51200, int(sys.argv[1]) if len(sys.argv) > 1 else 128
L = ca.lib()
a, b = C.c_float(), C.c_float()
ca._lib.check(L.nvsm_debug_dt_time(300, 256, K, split, 1, 0, C.byref(a), C.byref(b)))
t = np.zeros(2 * 4 * 16 * 8, np.uint64)
L.nvsm_debug_dt_ticks.argtypes = [C.c_void_p, C.c_int]
assert L.nvsm_debug_dt_ticks(t.ctypes.data, t.size) == 0
tk = t.astype(np.int64).reshape(2, 4, 16, 8)
print("cycles since the workgroup wave 0 tile 2 top; half 0: top | fragments requested | MFMAs + staging issued | barrier passed; half 1: top | MFMAs + staging issued | fragments requested | barrier passed")
for bw in range(2):
    for wv, name in enumerate(("w0", "w4", "w1", "w7")):
        for kt in range(2, 8):
            r = tk[bw, wv, kt] - tk[bw, 0, 2, 0]
            print("wg %d %s tile %2d: " % (bw, name, kt) + " ".join("%7d" % x for x in r[:4]))
