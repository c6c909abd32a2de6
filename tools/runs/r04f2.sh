#!/bin/bash
# the cold-row cache filled by the copy engines (SCONE_STAGE_FILL=sdma): parity tests, then the C4 Zipf step against the copy kernel
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04f2}
mkdir -p $O
cd $R
SCONE_STAGE_FILL=sdma timeout -k 10 500 python -m pytest tests/test_gpu_bench_shape.py -x -q -m gpu -k "cold_row_cache or rows_written_after or pinned" > $O/pytest_sdma_fill.txt 2>&1
echo "pytest rc=$? $(tail -2 $O/pytest_sdma_fill.txt | head -c 300)"
[ "${2:-}" = "testsonly" ] && exit 0
for cfg in "kernel 8" "sdma 8" "sdma 4" "sdma 12" "kernel 8" "sdma 8"; do
  set -- $cfg
  SCONE_STAGE_FILL=$1 SCONE_STAGE_FILL_THREADS=$2 timeout -k 10 300 python tools/c4_zipf_probe.py --mode cached --cache-rows 16000000 \
     --stage-tokens 262144 --prefetch-next --steps 40 --warmup 400 > $O/fill_$1_t$2.json 2> $O/fill_$1_t$2.err || { echo "$1 $2 failed"; tail -5 $O/fill_$1_t$2.err; exit 1; }
  python -c "
import json
j=json.loads(open('$O/fill_$1_t$2.json').read().strip().splitlines()[-1])
print('fill $1 threads $2: %.3f ms' % j['ms_per_step'], 'copied', round(j['per_step']['rows_copied']), 'status', j['status'], j['checksum_last'])"
done
