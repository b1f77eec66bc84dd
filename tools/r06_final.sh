#!/bin/bash
# round 6, final check: steady-state timeline of the headline, the whole GPU suite, smoke, the default bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
tools/timeline.sh r06_nvsm --gate-every 4 > /dev/null 2>&1
( time timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -8 ) > gpurun_out/r06_final_tests.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_final_smoke.txt 2>&1
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r06_final_bench.json 2> gpurun_out/r06_final_bench.err
tail -4 gpurun_out/r06_final_bench.err; cat gpurun_out/r06_final_tests.txt gpurun_out/r06_final_smoke.txt
