#!/bin/bash
# A round's counter evidence in one gpurun call (rounds 5 and 6), on the round's loop: a DIFFERENT batch every step
# (bench.py's default), so the FETCH_SIZE / WRITE_SIZE averages are over launches that each see fresh rows -- nothing a
# previous step left in the 256-MB Infinity Cache.  The headline gets the full bench line + kernel trace + every PMC pass;
# every other HBM workload a quick line + kernel trace + FETCH_SIZE / WRITE_SIZE (separate runs, the program directly behind
# `rocprofv3 ... --`).  tools/summarize_profile.py <tag...> (build box) turns each gpurun_out/<tag>/ into profiles/<tag>/ and
# an entry of profiles/hbm_traffic.json keyed by workload signature + kernel-source hash.
#   tools/profile_counters.sh <tag> [headline|rest|all]
set -u
TAG=${1:-r05}
WHAT=${2:-all}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
if [ "$WHAT" != "rest" ]; then
  tools/profile_round.sh $TAG --no-configs --no-sharded-record > /dev/null 2>&1; echo "headline: $(head -c 300 gpurun_out/$TAG/bench.json)"
  # the same headline with ONE batch repeated (rounds 1-4), its counters beside the rotated ones
  O=$R/gpurun_out/${TAG}_same_batch; mkdir -p $O
  timeout -k 10 300 python bench.py --steps 50 --warmup 5 --quick --same-batch > $O/bench.json 2> $O/bench.err
  ( cd /tmp && export TMPDIR=/tmp
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 20 --warmup 3 --quick --same-batch > $O/trace.log 2>&1
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 5 --warmup 2 --quick --same-batch > $O/pmc_$c.log 2>&1
    done )
  cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
  python3 $R/tools/timed_kernel_avg.py $O/trace 20 $O/kernel_timed.json
  rm -rf $O/trace/*/*kernel_trace.csv $O/pmc_*/*/*kernel_trace.csv 2>/dev/null
  echo "same_batch: $(head -c 200 $O/bench.json)"
fi
[ "$WHAT" = "headline" ] && exit 0
quick() {  # tag, bench args...
  local T=$1; shift
  local O=$R/gpurun_out/$T
  mkdir -p $O
  timeout -k 10 600 python bench.py --steps 50 --warmup 5 --quick "$@" > $O/bench.json 2> $O/bench.err
  ( cd /tmp && export TMPDIR=/tmp
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 20 --warmup 3 --quick "$@" > $O/trace.log 2>&1
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 5 --warmup 2 --quick "$@" > $O/pmc_$c.log 2>&1
    done )
  cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
  python3 $R/tools/timed_kernel_avg.py $O/trace 20 $O/kernel_timed.json
  rm -rf $O/trace/*/*kernel_trace.csv $O/pmc_*/*/*kernel_trace.csv 2>/dev/null   # (keep the merge small: the counters and the stats are what is read)
  echo "$T: $(head -c 200 $O/bench.json)"
}
quick ${TAG}_hbm_variant --rows 10000000 --keygen structured
quick ${TAG}_mall_variant --rows 10000000 --keygen structured --vocab 262144
quick ${TAG}_c4_int4_100m --rows 100000000 --format int4 --dim 1024 --keygen structured
quick ${TAG}_c2_fp16 --format fp16
quick ${TAG}_c3_int8_10m_d1024 --rows 10000000 --dim 1024 --keygen zipf_gpu
quick ${TAG}_int4_1m_d1024 --format int4 --dim 1024
quick ${TAG}_int8_1m_d1280 --dim 1280
quick ${TAG}_zipf --stream zipf
