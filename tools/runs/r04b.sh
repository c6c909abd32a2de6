#!/bin/bash
# Round 4, second GPU call: the cold-row cache (tests, copy-grid sweep, steady state, kernel trace) and the CU-reserve curve.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04b}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shape.py tests/test_gpu_fuzz.py -x -q -m gpu \
  -k "stag or pinned or cold_row or cu_reserve or integration_stub" > $O/pytest_stage.log 2>&1
echo "pytest rc=$? $(tail -2 $O/pytest_stage.log | head -c 300)"
for blocks in 64 128 256 512 1024; do
  SCONE_STAGE_COPY_BLOCKS=$blocks timeout -k 10 300 python tools/c4_zipf_probe.py --mode cached --cache-rows 0 --steps 20 > $O/zipf_cached_min_b$blocks.json 2> $O/zipf_cached_min_b$blocks.err
  echo "blocks $blocks: $(grep -o '"ms_per_step": [0-9.]*' $O/zipf_cached_min_b$blocks.json) $(grep -o '"per_step": {[^}]*}' $O/zipf_cached_min_b$blocks.json)"
done
timeout -k 10 400 python tools/c4_zipf_probe.py --mode cached --cache-rows 8000000 --steps 40 --warmup 160 > $O/zipf_cached_8m.json 2> $O/zipf_cached_8m.err
echo "8M steady: $(grep -o '"ms_per_step": [0-9.]*' $O/zipf_cached_8m.json) $(grep -o '"per_step": {[^}]*}' $O/zipf_cached_8m.json)"
timeout -k 10 400 python tools/c4_zipf_probe.py --mode zero --steps 40 --warmup 10 > $O/zipf_zero.json 2> $O/zipf_zero.err
echo "zero: $(grep -o '"ms_per_step": [0-9.]*' $O/zipf_zero.json)"
( cd /tmp && export TMPDIR=/tmp
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cached -- python3 $R/tools/c4_zipf_probe.py --mode cached --cache-rows 0 --steps 10 > $O/trace_cached.log 2>&1 )
cp $(ls $O/trace_cached/*/*kernel_stats.csv | head -1) $O/cached_kernel_stats.csv
cp $(ls $O/trace_cached/*/*kernel_trace.csv | head -1) $O/cached_kernel_trace.csv 2>/dev/null
rm -rf $O/trace_cached
grep -E "k_stage|k_match_ell|k_embed_wave" $O/cached_kernel_stats.csv | cut -c1-60,200-400
timeout -k 10 600 python tools/cu_reserve_curve.py > $O/cu_reserve_curve.json 2> $O/cu_reserve_curve.err
tail -c 1500 $O/cu_reserve_curve.json
