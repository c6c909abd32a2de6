#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04n}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shape.py tests/test_gpu_fuzz.py -x -q -m gpu -k "stag or pinned or cold_row" > $O/pytest_stage.log 2>&1
echo "pytest rc=$? $(tail -2 $O/pytest_stage.log | head -c 300)"
line() { echo "$1: $(grep -o '"ms_per_step": [0-9.]*' $2) $(grep -o '"tokens_per_s": [0-9.]*' $2) $(grep -o '"per_step": {[^}]*}' $2)"; }
run() { # name cache st warm second_chance
  f=$O/$1.json
  SCONE_STAGE_SECOND_CHANCE=$5 timeout -k 10 400 python tools/c4_zipf_probe.py --mode cached --cache-rows $2 --stage-tokens $3 --steps 40 --warmup $4 --prefetch-next > $f 2> ${f%.json}.err
  line "$1" $f
}
run c16m_st262144_fifo 16000000 262144 400 0
run c16m_st262144_clock 16000000 262144 400 1
run c8m_st131072_fifo 8000000 131072 250 0
run c8m_st131072_clock 8000000 131072 250 1
run c32m_st262144_fifo 32000000 262144 700 0
run c32m_st262144_clock 32000000 262144 700 1
