import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cunvsm_amd as ca
L = ca.lib(); a, b = C.c_float(), C.c_float()
for K, slabs in ((51200, 128), (51200, 100), (12800, 128), (6400, 64), (6400, 128)):
    ca._lib.check(L.nvsm_debug_dt_time(300, 256, K, slabs, 30, 0, C.byref(a), C.byref(b)))
    print("rows %6d slabs %4d: product %7.1f us  reduce %6.1f us" % (K, slabs, a.value * 1e3, b.value * 1e3), flush=True)
