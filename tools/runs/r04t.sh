#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04t}
mkdir -p $O
cd $R
t0=$(date +%s)
SCONE_DIST_BACKEND=gloo SCONE_ONE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 500 python bench.py --gpus 2 > $O/bench_2ranks.json 2> $O/bench_2ranks.err
echo "rehearsal rc=$? wall=$(( $(date +%s) - t0 )) s"
python - $O/bench_2ranks.json <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith('{"metric"'):
        d=json.loads(line); s=d["sharded"]
        print(d.get("incomplete"), d.get("hung_stage"), s.get("world_sanity"), s.get("exchanges_agree"), s.get("status_bits"))
        for k,v in s["exchanges"].items(): print(k, {x:v.get(x) for x in ("ms_per_step","status_bits","records_transport","transport_fallback_reason","scales_with_world","error","skipped")}, (v.get("with_cu_reserve") or {}).get("ms_per_step"), v.get("sync_free_plan"))
PY
tail -3 $O/bench_2ranks.err | cut -c1-300
tools/run_configs.sh $1 2>&1 | grep -v "^==" | tail -12
