#!/bin/bash
cd "$(dirname "$0")/.."
python -m pytest tests/test_gpu_soak.py -x -q -m gpu --durations=5 2>&1 | tail -10
