#!/bin/bash
# k_match_ell duration against the number of workgroups (tail effect at 2048 resident workgroups): rocprofv3 kernel trace of
# short bench runs at batch sizes around 2048 x 512 tokens.   tools/match_tail.sh <tag> <batch>...
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for b in "$@"; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b$b -- python3 $R/bench.py --steps 20 --warmup 3 --quick --batch $b > $O/b$b.log 2>&1
  f=$(ls $O/b$b/*/*kernel_stats.csv | head -1)
  echo "batch $b: $(grep k_match_ell $f | awk -F'",' '{print "match avg ns", $4, "min", $6}')  $(grep k_embed_wave $f | awk -F'",' '{print "gather avg ns", $4}')" | tee -a $O/summary.txt
done
