#!/bin/bash
# Alternating un-profiled bench runs of several library builds on one box:
#   tools/ab_multi.sh <tag> <rounds> lib1.so lib2.so ... [-- bench args]
set -u
TAG=$1; ROUNDS=$2; shift 2
LIBS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ $# -gt 0 ] && shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for round in $(seq 1 $ROUNDS); do
  for lib in "${LIBS[@]}"; do
    n=$(basename $lib .so)
    SCONE_HIP_LIB=$R/$lib timeout 300 python bench.py --steps 50 --warmup 5 --quick "$@" > $O/${n}_$round.json 2>> $O/err.log || { echo "$n failed"; tail -3 $O/err.log; exit 1; }
    python3 - $O/${n}_$round.json $n $round <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
print("%-10s round %s  step %.4f ms  gather %.4f ms  %.3f G tok/s" % (sys.argv[2], sys.argv[3], r["ms_per_step"], r["roofline"]["avg_kernel_ms"], r["value"] / 1e9), flush=True)
PY
  done
done
