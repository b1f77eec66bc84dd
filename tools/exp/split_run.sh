#!/bin/bash
# times the projection products alone with the exact-fp32 kernels (split=0) and the split-bf16 kernel (9 / 6 products)
for s in 9 6; do for nt in 0 1; do TAG="split=$s nt=$nt" NVSM_GEMM_SPLIT=$s NVSM_SPLIT_NT=$nt python tools/exp/gemm_time.py 51200 25600 2>&1 | grep -v amdgpu.ids; done; done
