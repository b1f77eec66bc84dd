#!/bin/bash
for v in "NVSM_X=0" "NVSM_X=0" "NVSM_X=0" "NVSM_SORT_LAYOUT=2" "NVSM_SORT_LAYOUT=2" "NVSM_SORT_LAYOUT=4" "NVSM_SORT_LAYOUT=4"; do
  echo "== $v"; env $v timeout 300 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "back_to_back" 2>&1 | grep -E "passed|failed|Mismatched|Max abs"
done
