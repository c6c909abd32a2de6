#!/bin/bash
# tools/profile_round.sh for the secondary configs (kernel stats + PMC passes); summaries via tools/summarize_profile.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
tools/profile_round.sh r01f_c2_fp16 --format fp16 > gpurun_out/r01f_c2.log 2>&1; tail -1 gpurun_out/r01f_c2.log | cut -c1-200
tools/profile_round.sh r01f_c3_int8_10m_d1024 --rows 10000000 --dim 1024 > gpurun_out/r01f_c3.log 2>&1; tail -1 gpurun_out/r01f_c3.log | cut -c1-200
tools/profile_round.sh r01f_int4_1m_d1024 --format int4 --dim 1024 > gpurun_out/r01f_i4.log 2>&1; tail -1 gpurun_out/r01f_i4.log | cut -c1-200
tools/profile_round.sh r01f_zipf --stream zipf > gpurun_out/r01f_zipf.log 2>&1; tail -1 gpurun_out/r01f_zipf.log | cut -c1-200
