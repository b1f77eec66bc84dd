"""Where the time of the row-panel GEMM (gemm_rows.hip) goes: a build with -DNVSM_ROWS_DBG switches parts of the kernel off
(NVSM_ROWS_DBG bits: 1 no multiply, 2 no tile loads in the loop, 4 no LDS stores, 8 no epilogue); alone on an idle GPU.
   make -C cunvsm_amd/csrc dbg && CUNVSM_AMD_LIB=$PWD/cunvsm_amd/libcunvsm_amd_dbg.so python tools/exp/rows_exp.py"""
import ctypes as C
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

SHAPES = [("fwd  6400", 0, 6400, 256, 300, 1), ("bwd  6400", 1, 6400, 300, 256, 2), ("fwd 12800", 0, 12800, 256, 300, 1),
          ("fwd  4096 lse", 0, 4096, 256, 128, 0), ("bwd  4096 lse", 1, 4096, 128, 256, 2)]


def run_once():
    import cunvsm_amd as ca
    out = []
    for name, bl, M, N, K, extras in SHAPES:
        ms = C.c_float()
        ca._lib.check(ca.lib().nvsm_debug_gemm_time(bl, M, N, K, extras, 200, C.byref(ms)))
        out.append("%s %.1f us" % (name, ms.value * 1e3))
    print(os.environ.get("TAG", ""), " | ".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "once":
        run_once()
    else:
        for tag, env in (("rows kernel       ", {}), ("no multiply       ", {"NVSM_ROWS_DBG": "1"}), ("no loop loads     ", {"NVSM_ROWS_DBG": "2"}),
                         ("no LDS stores     ", {"NVSM_ROWS_DBG": "4"}), ("no epilogue       ", {"NVSM_ROWS_DBG": "8"}),
                         ("only multiply     ", {"NVSM_ROWS_DBG": "14"}), ("nothing (launch)  ", {"NVSM_ROWS_DBG": "15"}),
                         ("one tile per wave ", {"NVSM_ROWS_TPW": "1"}), ("  no epilogue     ", {"NVSM_ROWS_TPW": "1", "NVSM_ROWS_DBG": "8"}),
                         ("  only multiply   ", {"NVSM_ROWS_TPW": "1", "NVSM_ROWS_DBG": "14"}),
                         ("large-batch kernels", {"NVSM_GEMM_ROWS_MAX": "0"})):
            subprocess.call([sys.executable, os.path.abspath(__file__), "once"], env=dict(os.environ, TAG=tag, **env))
