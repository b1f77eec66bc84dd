#!/bin/bash
cd "$(dirname "$0")/.."
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r06_r_bench.json 2> gpurun_out/r06_r_bench.err
tail -4 gpurun_out/r06_r_bench.err
python -m pytest tests/test_bench_gpu.py tests/test_bench_launch.py -q -m gpu 2>&1 | tail -3
