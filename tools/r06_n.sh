#!/bin/bash
cd "$(dirname "$0")/.."
CUNVSM_AMD_LIB=$PWD/cunvsm_amd/libcunvsm_amd_priot.so python tools/exp/split_times.py 51200 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_n_split_times_prio.txt
cat gpurun_out/r06_n_split_times_prio.txt
