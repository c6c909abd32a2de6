#!/bin/bash
# bench.py --gpus 4 on ONE GPU over gloo at reduced table size: the N > 1 record with four ranks (world sanity, six exchanges)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04t4}
mkdir -p $O
cd $R
t0=$(date +%s)
SCONE_DIST_BACKEND=gloo SCONE_ONE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 timeout -k 10 700 python bench.py --gpus 4 \
  --sharded-rows-per-rank 20000000 --pinned-rows 20000000 --cpu-seconds 2 > $O/bench_4ranks.json 2> $O/bench_4ranks.err
echo "rehearsal rc=$? wall=$(( $(date +%s) - t0 )) s"
python - $O/bench_4ranks.json <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith('{"metric"'):
        d=json.loads(line); s=d["sharded"]
        print(d["n_gpus"], d.get("incomplete"), d.get("hung_stage"), s.get("world_sanity"), s.get("exchanges_agree"), s.get("status_bits"))
        for k,v in s["exchanges"].items(): print(k, {x:v.get(x) for x in ("ms_per_step","status_bits","transport_fallback_reason","error","skipped")})
PY
tail -3 $O/bench_4ranks.err | cut -c1-300
