#!/bin/bash
# round 6, call H: the whole GPU suite + smoke + the default bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -15 ) > gpurun_out/r06_h_tests.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_h_smoke.txt 2>&1
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/r06_h_bench.json 2> gpurun_out/r06_h_bench.err
tail -4 gpurun_out/r06_h_bench.err; cat gpurun_out/r06_h_tests.txt gpurun_out/r06_h_smoke.txt
