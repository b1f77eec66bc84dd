"""alone times of the batch-sized projection products at per-rank batch sizes: gemm_rsplit against gemm_rows (NVSM_GEMM_RSPLIT=0 in the experiments build)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cunvsm_amd as ca
L = ca.lib()
def t(bl, M, N, K, extras, rep=200):
    ms = ctypes.c_float()
    ca._lib.check(L.nvsm_debug_gemm_time(bl, M, N, K, extras | 4, rep, ctypes.byref(ms)))
    return ms.value * 1e3
for M in [int(x) for x in (sys.argv[1:] or ["4096", "6400", "8192"])]:
    print("M=%d  fwd(300->256 +colstats) %.1f us  bwd(256->300 +rowsq) %.1f us  lse fwd(128->256) %.1f  lse bwd(256->128 +rowsq) %.1f" %
          (M, t(0, M, 256, 300, 1), t(1, M, 300, 256, 2), t(0, M, 256, 128, 0), t(1, M, 128, 256, 2)))
