"""Times the row sort alone (nvsm_debug_sort) at the shapes of the bench: python tools/exp/sort_time.py"""
import ctypes as C
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cunvsm_amd as ca

def zipf(rs, rows, n):
    p = 1.0 / np.arange(1, rows + 1); p /= p.sum()
    return rs.choice(rows, size=n, p=p)

rs = np.random.RandomState(1)
for name, keys, bits in (("docs 870k/17b uniform", rs.randint(0, 100000, 870400), 17), ("words 512k/16b zipf", zipf(rs, 50000, 512000), 16),
                         ("lse words 41k/18b zipf", zipf(rs, 200000, 40960), 18), ("lse docs 70k/17b", rs.randint(0, 100000, 69632), 17),
                         ("docs 870k/21b", rs.randint(0, 2000000, 870400), 21)):
    k = np.ascontiguousarray(keys, np.int32); ko = np.empty_like(k); vo = np.empty_like(k)
    ms = C.c_float()
    ca._lib.check(ca.lib().nvsm_debug_sort(k.size, bits, k.ctypes.data, ko.ctypes.data, vo.ctypes.data, 20, C.byref(ms)))
    print("%-28s %.1f us" % (name, ms.value * 1e3))
