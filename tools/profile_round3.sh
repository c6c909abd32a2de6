#!/bin/bash
# Round-3 evidence in one gpurun call: the headline (full bench line + kernel trace + PMC passes), the cache-defeating
# variant and C4-in-HBM (quick bench line + kernel trace + FETCH_SIZE / WRITE_SIZE passes), then every config of
# tools/run_configs.sh.   tools/profile_round3.sh <tag>      -> gpurun_out/<tag>{,_hbm_variant,_c4_int4_100m}/, gpurun_out/<tag>/configs.jsonl
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
tools/profile_round.sh $TAG > /dev/null 2>&1; echo "headline: $(tail -c 300 gpurun_out/$TAG/bench.json | head -c 200)"
quick() {  # tag, bench args...
  local T=$1; shift
  local O=$R/gpurun_out/$T
  mkdir -p $O
  timeout 600 python bench.py --steps 50 --warmup 5 --quick "$@" > $O/bench.json 2> $O/bench.err
  ( cd /tmp && export TMPDIR=/tmp
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 20 --warmup 3 --quick "$@" > $O/trace.log 2>&1
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 5 --warmup 2 --quick "$@" > $O/pmc_$c.log 2>&1
    done )
  cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
  echo "$T: $(head -c 200 $O/bench.json)"
}
quick ${TAG}_hbm_variant --rows 10000000 --keygen structured
quick ${TAG}_c4_int4_100m --rows 100000000 --format int4 --dim 1024 --keygen structured
tools/run_configs.sh $TAG 2>&1 | grep -v "^==" | tail -12
