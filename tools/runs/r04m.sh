#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04m}
mkdir -p $O
cd $R
line() { echo "$1: $(grep -o '"ms_per_step": [0-9.]*' $2) $(grep -o '"tokens_per_s": [0-9.]*' $2) $(grep -o '"per_step": {[^}]*}' $2)"; }
run() { # name cache st warm blocks
  f=$O/$1.json
  SCONE_STAGE_COPY_BLOCKS=$5 timeout -k 10 400 python tools/c4_zipf_probe.py --mode cached --cache-rows $2 --stage-tokens $3 --steps 40 --warmup $4 --prefetch-next > $f 2> ${f%.json}.err
  line "$1" $f
}
run c32m_st262144_pf 32000000 262144 600 128
run c16m_st262144_pf 16000000 262144 320 128
run c8m_st262144_pf_b96 8000000 262144 160 96
run c8m_st262144_pf_b192 8000000 262144 160 192
run c8m_st524288_pf 8000000 524288 160 128
run c32m_st524288_pf 32000000 524288 600 128
