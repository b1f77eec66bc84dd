"""Where a workgroup of the split-bf16 GEMM spends its time (make -C cunvsm_amd/csrc dbg; CUNVSM_AMD_LIB=.../libcunvsm_amd_dbg.so):
wall-clock stamps of wave 0 of every workgroup of ONE launch, relative to the earliest start."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cunvsm_amd as ca

M = int(sys.argv[1]) if len(sys.argv) > 1 else 51200
L = ca.lib()
for name, bl, N, K, extras in (("fwd", 0, 256, 300, 1), ("bwd", 1, 300, 256, 2)):
    ms = C.c_float()
    ca._lib.check(L.nvsm_debug_gemm_time(bl, M, N, K, extras | 4, 1, C.byref(ms)))
    t = np.zeros(256 * 8, np.uint64)
    L.nvsm_debug_split_times.argtypes = [C.c_void_p, C.c_int]
    assert L.nvsm_debug_split_times(t.ctypes.data, t.size) == 0
    t = t.reshape(256, 8)[:, :6].astype(np.int64)
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0
    names = ("start", "loop entered", "first pass: loop done", "last pass: loop done", "stores issued", "end")
    print(name, "event: min / median / max over the workgroups (us since the first start)")
    for j, nm in enumerate(names):
        print("   %-24s %6.1f %6.1f %6.1f" % (nm, us[:, j].min(), np.median(us[:, j]), us[:, j].max()))
    print("   kernel (events): %.1f us" % (ms.value * 1e3), flush=True)
    t = np.zeros(1024 * 8, np.uint64)
    assert L.nvsm_debug_split_times(t.ctypes.data, t.size) == 0
    tk = t[4096:4096 + 256].astype(np.int64).reshape(2, 16, 8)
    print("   K loop of workgroup 0, shader cycles since the wave's first stamp: top | block 0 done | half the blocks | MFMAs issued | B taken over | barrier passed")
    for wv in range(2):
        for kt in range(0, 6):
            r = tk[wv, kt] - tk[wv, 0, 0]
            print("   wave %d tile %d: %7d %7d %7d %7d %7d %7d" % (wv * 4, kt, r[0], r[4], r[5], r[1], r[2], r[3]))
