#!/bin/bash
cd "$(dirname "$0")/.."
for r in 1 2; do
for lib in libcunvsm_amd.so libcunvsm_amd_prio.so; do
  TAG=$lib CUNVSM_AMD_LIB=$PWD/cunvsm_amd/$lib python tools/exp/gemm_time.py 51200 25600 2>&1 | grep -v amdgpu.ids
done; done > gpurun_out/r06_m_gemm.txt
tools/ab_lib.sh libcunvsm_amd.so libcunvsm_amd_prio.so > gpurun_out/r06_m_ab.txt 2>&1
cat gpurun_out/r06_m_gemm.txt gpurun_out/r06_m_ab.txt
