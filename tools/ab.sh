#!/bin/bash
# A/B of two builds of libscone_hip.so on ONE box (boxes of the pool differ by several per cent):
#   tools/ab.sh <tag> <libA.so> <libB.so> [bench args]
# alternates un-profiled bench runs (ms_per_step, avg gather kernel ms) and takes one
# rocprofv3 --kernel-trace --stats pass per build; everything lands in gpurun_out/<tag>/.
set -u
TAG=$1; A=$2; B=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for round in 1 2 3; do
  for v in A B; do
    lib=$A; [ $v = B ] && lib=$B
    SCONE_HIP_LIB=$R/$lib timeout 300 python bench.py --steps 50 --warmup 5 --quick "$@" > $O/bench_$v$round.json 2>> $O/bench.err || exit 1
    python - "$O/bench_$v$round.json" $v$round <<'EOF'
import json, sys
r = json.load(open(sys.argv[1]))
print(sys.argv[2], "ms_per_step %.4f" % r["ms_per_step"], "gather_ms %.4f" % r["roofline"]["avg_kernel_ms"], "Gtok/s %.3f" % (r["value"] / 1e9), flush=True)
EOF
  done
done
cd /tmp && export TMPDIR=/tmp
for v in A B; do
  lib=$A; [ $v = B ] && lib=$B
  export SCONE_HIP_LIB=$R/$lib
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$v -- python3 $R/bench.py --steps 20 --warmup 3 --quick "$@" > $O/trace_$v.log 2>&1 || exit 1
  f=$(ls $O/trace_$v/*/*kernel_stats.csv | head -1)
  cp $f $O/kernel_stats_$v.csv
  echo "== $v ($lib)"; python3 - $f <<'EOF'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    n = row["Name"]
    if "k_match_ell" in n or "k_embed_wave" in n or "k_index_insert" in n:
        print("  %-18s calls %4s avg %10.1f us min %10.1f us" % (n.split("(")[0].split("::")[-1][:18], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3))
EOF
done
