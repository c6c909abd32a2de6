#!/bin/bash
# A/B of the working tree against the round-1 tree staged under gpurun_ab/r1tree (git worktree of the round-1 head,
# built): alternating un-profiled bench runs of the same workload on one box.
#   tools/ab_r1.sh <tag> <rounds> [bench args]
set -u
TAG=$1; ROUNDS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
show() { python3 - "$1" "$2" <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
print("%-8s step %.4f ms  gather %.4f ms  %.3f G tok/s" % (sys.argv[2], r["ms_per_step"], r["roofline"]["avg_kernel_ms"], r["value"] / 1e9), flush=True)
PY
}
for round in $(seq 1 $ROUNDS); do
  timeout 300 python gpurun_ab/r1tree/bench.py --steps 50 --warmup 5 --no-cpu-baseline "$@" > $O/r1_$round.json 2>> $O/err.log || { tail -3 $O/err.log; exit 1; }
  show $O/r1_$round.json "r1"
  timeout 300 python bench.py --steps 50 --warmup 5 --quick "$@" > $O/now_$round.json 2>> $O/err.log || { tail -3 $O/err.log; exit 1; }
  show $O/now_$round.json "now"
done
