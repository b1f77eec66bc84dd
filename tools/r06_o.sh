#!/bin/bash
cd "$(dirname "$0")/.."
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm_split or gemm" 2>&1 | tail -3 > gpurun_out/r06_o_tests.txt
python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3 >> gpurun_out/r06_o_tests.txt
for r in 1 2 3; do
for lib in libcunvsm_amd_late.so libcunvsm_amd.so; do
  TAG=$lib CUNVSM_AMD_LIB=$PWD/cunvsm_amd/$lib python tools/exp/gemm_time.py 51200 25600 12800 2>&1 | grep -v amdgpu.ids
done; done > gpurun_out/r06_o_gemm.txt
CUNVSM_AMD_LIB=$PWD/cunvsm_amd/libcunvsm_amd_dbg.so python tools/exp/split_times.py 51200 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_o_split_times.txt
tools/ab_lib.sh libcunvsm_amd_late.so libcunvsm_amd.so > gpurun_out/r06_o_ab.txt 2>&1
cat gpurun_out/r06_o_tests.txt gpurun_out/r06_o_gemm.txt gpurun_out/r06_o_ab.txt; grep -A 16 "^fwd" gpurun_out/r06_o_split_times.txt
