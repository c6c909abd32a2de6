#!/bin/bash
# Round 6: (1) the NO_MATH timing probe (every load and store stays, the arithmetic becomes one XOR per word) alternating with
# the shipped library; (2) does the `latency` block (6 side streams + hipGraph captures earlier in the process) slow the
# pinned-host staging pipeline measured later in the same process?  bench.py with and without --no-latency, twice.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06d}
mkdir -p $O
cd $R
tools/ab_multi.sh ${1:-r06d}/probes 3 scone_amd/csrc/libscone_hip.so gpurun_ab/libno_math.so | tee $O/probes.txt
for round in 1 2; do
  for v in with_latency no_latency; do
    extra=""; [ $v = no_latency ] && extra="--no-latency"
    timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --no-configs --no-hbm-variant --no-cpu-baseline $extra > $O/${v}_$round.json 2> $O/${v}_$round.err
    python3 - $O/${v}_$round.json $v $round <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
z = r["sharded"]["n1_pinned_host_zipf"]
print("%-13s %s  cached+prefetch %.3f ms (%.3f G)  in place %.3f ms  static head %.3f ms  scrambled cached %.3f ms" % (
    sys.argv[2], sys.argv[3], z["ms_per_step"], z["value"] / 1e9, z["zero_copy_same_stream"]["ms_per_step"],
    z["zero_copy_static_head_same_hbm"]["ms_per_step"], z["scrambled_order"]["ms_per_step"]), flush=True)
PY
  done
done 2>&1 | tee $O/pinned.txt
