#!/bin/bash
cd "$(dirname "$0")/.."
tools/timeline.sh r06_lse --config=lse_small --gate-every 4 > /dev/null 2>&1
tools/timeline.sh r06_b6400 --batch=6400 --gate-every 4 > /dev/null 2>&1
ls gpurun_out | grep timeline_r06
