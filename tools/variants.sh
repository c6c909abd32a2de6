#!/bin/bash
# Per-kernel times of several builds of libscone_hip.so on one box (ablation studies):
#   tools/variants.sh <tag> <pattern> lib1.so lib2.so ... [-- bench args]
# one rocprofv3 --kernel-trace --stats pass per build; prints the kernels whose name contains <pattern>.
set -u
TAG=$1; PAT=$2; shift 2
LIBS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ $# -gt 0 ] && shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in "${LIBS[@]}"; do
  n=$(basename $lib .so)
  export SCONE_HIP_LIB=$R/$lib
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$n -- python3 $R/bench.py --steps 20 --warmup 3 --quick "$@" > $O/trace_$n.log 2>&1 || { echo "$n failed"; tail -5 $O/trace_$n.log; exit 1; }
  f=$(ls $O/trace_$n/*/*kernel_stats.csv | head -1)
  cp $f $O/kernel_stats_$n.csv
  python3 - $f $n "$PAT" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    if sys.argv[3] in row["Name"]:
        print("%-20s %-24s calls %4s avg %9.1f us min %9.1f us" % (sys.argv[2], row["Name"].split("(")[0][-24:], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3), flush=True)
PY
  rm -rf $O/trace_$n
done
