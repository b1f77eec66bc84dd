#!/bin/bash
# round-5 profile set (GPU box): kernel stats, HBM / MFMA counters and gated timelines of the headline, the per-rank share, the LSE
# recipe and the large tables; summaries land in gpurun_out/prof_<tag>/ — copy the r05_* files into profiles/
cd "$(dirname "$0")/.."
tools/profile_round.sh r05_nvsm > gpurun_out/prof_r05_nvsm.log 2>&1
tools/profile_round.sh r05_b6400 --batch 6400 > gpurun_out/prof_r05_b6400.log 2>&1
tools/profile_round.sh r05_lse --config lse_small > gpurun_out/prof_r05_lse.log 2>&1
tools/profile_round.sh r05_large --config large_tables > gpurun_out/prof_r05_large.log 2>&1
ls gpurun_out/prof_r05_*/r05_* | head -40
