#!/bin/bash
# C4 on streams whose popularity is NOT the table's order (scrambled ranks; a hot set that moves): reading in place, the
# static head enlarged by the cache's HBM, and the cache with prefetch.  One process per mechanism.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04v}
mkdir -p $O
cd $R
run() {   # name, args...
  local name=$1; shift
  timeout -k 10 400 python tools/c4_zipf_probe.py "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -3 $O/$name.err; return 1; }
  python - "$O/$name.json" "$name" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "ms/step %.3f" % j["ms_per_step"], "G tok/s %.3f" % (j["tokens_per_s"] / 1e9), j.get("per_step"), j["per_batch_stats"][:1], "status", j["status"])
PY
}
for S in "--scramble" "--shift-per-step 50000"; do
  tag=$(echo $S | tr -d ' -')
  run zero_$tag --mode zero --steps 20 --warmup 400 $S &&
  run zero_head17M_$tag --mode zero --hot 17000000 --steps 20 --warmup 400 $S &&
  run cached16M_prefetch_$tag --mode cached --cache-rows 16000000 --stage-tokens 262144 --prefetch-next --steps 20 --warmup 400 $S || exit 1
done
