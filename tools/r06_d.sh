#!/bin/bash
# round 6, call D: the folded data-parallel collective, the soak test, the new bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_dp_gloo.py -x -q -m gpu 2>&1 | tail -15
  timeout 600 python -m pytest tests/test_gpu_soak.py -x -q -m gpu --durations=5 2>&1 | tail -15
  timeout 1200 python -m pytest tests/test_bench_gpu.py tests/test_bench_launch.py -x -q -m gpu --durations=5 2>&1 | tail -30 ) > gpurun_out/r06_d_tests.txt 2>&1
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/r06_d_bench.json 2> gpurun_out/r06_d_bench.err
tail -5 gpurun_out/r06_d_bench.err
cat gpurun_out/r06_d_tests.txt
