#!/bin/bash
# Round 4: the cold-row cache in steady state -- chunk size x copy grid x cache size, against zero-copy (same hot head, and a
# static head enlarged by the cache's HBM budget)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04f}
mkdir -p $O
cd $R
line() { echo "$1: $(grep -o '"ms_per_step": [0-9.]*' $2) $(grep -o '"tokens_per_s": [0-9.]*' $2) $(grep -o '"per_step": {[^}]*}' $2)"; }
for st in 65536 131072 262144; do for b in 96 128; do
  f=$O/cached_8m_st${st}_b$b.json
  SCONE_STAGE_COPY_BLOCKS=$b timeout -k 10 300 python tools/c4_zipf_probe.py --mode cached --cache-rows 8000000 --stage-tokens $st --steps 40 --warmup 160 > $f 2> ${f%.json}.err
  line "8M st=$st blocks=$b" $f
done; done
f=$O/cached_32m.json
SCONE_STAGE_COPY_BLOCKS=128 timeout -k 10 400 python tools/c4_zipf_probe.py --mode cached --cache-rows 32000000 --stage-tokens 131072 --steps 40 --warmup 600 > $f 2> ${f%.json}.err
line "32M st=131072 blocks=128" $f
f=$O/zero_hot1m.json
timeout -k 10 300 python tools/c4_zipf_probe.py --mode zero --steps 40 --warmup 10 > $f 2> ${f%.json}.err; line "zero-copy hot=1M" $f
f=$O/zero_hot9m.json
timeout -k 10 300 python tools/c4_zipf_probe.py --mode zero --hot 9000000 --steps 40 --warmup 10 > $f 2> ${f%.json}.err; line "zero-copy hot=9M" $f
