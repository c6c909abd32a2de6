#!/bin/bash
# Round 6: the staging pipeline's side streams against the process's OTHER streams.  bench.py's `latency` block (6 torch side
# streams + hipGraph captures) earlier in the process costs the pinned-host cached step 27 % (profiles/r06d): which stream
# priorities of the pipeline's PREP / COPY streams (SCONE_STAGE_PRIO="<prep>,<copy>"; -1 high, 0 normal, 1 low) make it robust?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06g}
mkdir -p $O
cd $R
for prio in "0,0" "0,-1" "1,-1" "-1,-1" "1,0"; do
  for v in with_latency no_latency; do
    extra=""; [ $v = no_latency ] && extra="--no-latency"
    SCONE_STAGE_PRIO=$prio timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --no-configs --no-hbm-variant --no-cpu-baseline $extra > $O/${v}_$prio.json 2> $O/${v}_$prio.err
    python3 - $O/${v}_$prio.json $v $prio <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
z = r["sharded"]["n1_pinned_host_zipf"]
print("prio %-6s %-13s cached+prefetch %.3f ms (%.3f G)  in place %.3f ms  static head %.3f ms  scrambled cached %.3f ms  n1 uniform in place %.3f ms" % (
    sys.argv[3], sys.argv[2], z["ms_per_step"], z["value"] / 1e9, z["zero_copy_same_stream"]["ms_per_step"],
    z["zero_copy_static_head_same_hbm"]["ms_per_step"], z["scrambled_order"]["ms_per_step"], r["sharded"]["n1_pinned_host"]["ms_per_step"]), flush=True)
PY
  done
done 2>&1 | tee $O/pinned.txt
