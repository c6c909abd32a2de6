#!/bin/bash
# Round 5: the cache-defeating variants side by side with their counter passes -- 10M-row INT8 d = 768 table, structured
# vocabulary of 50,257 tokens (hbm_variant) and of 262,144 tokens (mall_variant: the token-indexed rows of a launch no longer
# fit the Infinity Cache).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05i}
cd $R
quick() {  # tag, bench args...
  local T=$1; shift
  local O=$R/gpurun_out/$T
  mkdir -p $O
  timeout -k 10 600 python bench.py --steps 50 --warmup 5 --quick "$@" > $O/bench.json 2> $O/bench.err
  ( cd /tmp && export TMPDIR=/tmp
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 20 --warmup 3 --quick "$@" > $O/trace.log 2>&1
    for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
      n=$(echo $c | tr " " "_" | cut -c1-32)
      timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- python3 $R/bench.py --steps 5 --warmup 2 --quick "$@" > $O/pmc_$n.log 2>&1
    done )
  cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
  rm -rf $O/trace/*/*kernel_trace.csv $O/pmc_*/*/*kernel_trace.csv 2>/dev/null
  python3 tools/show_bench.py $O/bench.json | head -1
}
quick ${TAG}_mall_variant --rows 10000000 --keygen structured --vocab 262144
quick ${TAG}_hbm_variant --rows 10000000 --keygen structured
