#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04s}
mkdir -p $O
cd $R
line() { echo "$1: $(grep -o '"ms_per_step": [0-9.]*' $2) $(grep -o '"tokens_per_s": [0-9.]*' $2) $(grep -o '"per_step": {[^}]*}' $2)"; }
for b in 128 256 512 128 256; do
  f=$O/c16m_pf_b$b.json
  SCONE_STAGE_COPY_BLOCKS=$b timeout -k 10 400 python tools/c4_zipf_probe.py --mode cached --cache-rows 16000000 --stage-tokens 262144 --steps 40 --warmup 400 --prefetch-next > $f 2> ${f%.json}.err; line "16M pf blocks=$b" $f
done
