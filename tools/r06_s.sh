#!/bin/bash
cd "$(dirname "$0")/.."
python -m pytest tests/test_bench_gpu.py -q -m gpu -k two_ranks 2>&1 | tail -3
