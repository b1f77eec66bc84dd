#!/bin/bash
# round-6 profile set (GPU box): kernel stats, HBM / MFMA counters and gated timelines of the headline, the per-rank share, the LSE
# recipe and the large tables; summaries land in gpurun_out/prof_<tag>/ — copy the r06_* files into profiles/
cd "$(dirname "$0")/.."
tools/profile_round.sh r06_nvsm > gpurun_out/prof_r06_nvsm.log 2>&1
tools/profile_round.sh r06_b6400 --batch 6400 > gpurun_out/prof_r06_b6400.log 2>&1
tools/profile_round.sh r06_lse --config lse_small > gpurun_out/prof_r06_lse.log 2>&1
tools/profile_round.sh r06_large --config large_tables > gpurun_out/prof_r06_large.log 2>&1
ls gpurun_out/prof_r06_*/r06_* | head -40
