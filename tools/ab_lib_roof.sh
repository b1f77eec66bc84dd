#!/bin/bash
# interleaved A/B of library builds with the roofline kernel's time: tools/ab_lib_roof.sh libA.so libB.so ...
cd "$(dirname "$0")/.."
for round in 1 2 3; do for lib in "$@"; do
  r=$(CUNVSM_AMD_LIB=$PWD/cunvsm_amd/$lib python bench.py --steps 60 --repeats 3 --no-cpu-baseline --no-extra-legs $BENCH_FLAGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['kernel_breakdown']['gemm_fwd']['avg_ms'])")
  echo "round $round [$lib] $r"
done; done
