"""dT product alone: the split-bf16 kernel (gemm_dt.hip) against the tiled exact-fp32 kernel, by batch size and slab count
(wall time of nvsm_debug_gemm minus a memcpy-only baseline is too noisy: use rocprofv3 --kernel-trace --stats on this script)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cunvsm_amd as ca
rs = np.random.RandomState(0)
for K in (6400, 12800, 51200):
    A = rs.standard_normal((K, 300)).astype(np.float32); B = rs.standard_normal((K, 256)).astype(np.float32)
    out = np.empty((300, 256), np.float32)
    for mode, split in (("1", 16), ("1", 64), ("1", 128), ("0", 16), ("0", 50)):
        os.environ["NVSM_DT_SPLIT"] = mode
        ca._lib.check(ca.lib().nvsm_debug_gemm((split << 2) | 2, 300, 256, K, A.ctypes.data, B.ctypes.data, out.ctypes.data))
