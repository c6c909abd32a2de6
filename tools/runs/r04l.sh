#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04l}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shape.py tests/test_gpu_fuzz.py -x -q -m gpu -k "stag or pinned or cold_row" > $O/pytest_stage.log 2>&1
echo "pytest rc=$? $(tail -2 $O/pytest_stage.log | head -c 300)"
line() { echo "$1: $(grep -o '"ms_per_step": [0-9.]*' $2) $(grep -o '"tokens_per_s": [0-9.]*' $2) $(grep -o '"per_step": {[^}]*}' $2)"; }
for st in 131072 262144; do for pf in "" "--prefetch-next"; do
  f=$O/cached_8m_st${st}${pf}.json
  timeout -k 10 300 python tools/c4_zipf_probe.py --mode cached --cache-rows 8000000 --stage-tokens $st --steps 40 --warmup 160 $pf > $f 2> ${f%.json}.err
  line "8M st=$st $pf" $f
done; done
for pf in "" "--prefetch-next"; do
  f=$O/cached_32m${pf}.json
  timeout -k 10 400 python tools/c4_zipf_probe.py --mode cached --cache-rows 32000000 --stage-tokens 131072 --steps 40 --warmup 600 $pf > $f 2> ${f%.json}.err
  line "32M st=131072 $pf" $f
done
f=$O/zero_hot1m.json
timeout -k 10 300 python tools/c4_zipf_probe.py --mode zero --steps 40 --warmup 10 > $f 2> ${f%.json}.err; line "zero-copy hot=1M" $f
