"""Times launch_gemm on device operands for the two projection products at several batch sizes (alone, idle GPU):
   python tools/exp/gemm_time.py [M ...]    (NVSM_GEMM_SPLIT / NVSM_GEMM_ROWS_MAX / NVSM_ROWS_TPW / NVSM_GEMM_TSTAT select the kernel)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cunvsm_amd as ca

sizes = [int(a) for a in sys.argv[1:]] or [6400, 12800, 25600, 51200]
for M in sizes:
    out = []
    for name, bl, N, K, extras in (("fwd", 0, 256, 300, 1), ("bwd", 1, 300, 256, 2)):
        ms = C.c_float()
        ca._lib.check(ca.lib().nvsm_debug_gemm_time(bl, M, N, K, extras | 4, 100, C.byref(ms)))      # 4: the planes of B stay valid
        tf = 2.0 * M * N * K / (ms.value * 1e-3) / 1e12
        out.append("%s %.1f us (%.0f TF/s)" % (name, ms.value * 1e3, tf))
    print(os.environ.get("TAG", ""), M, " | ".join(out), flush=True)
