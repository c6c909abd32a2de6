#!/bin/bash
# Round 5: (i) what the library's timing events cost the step; (ii) rehearsal of `bench.py --gpus 2` at the sharded record's
# full size with two ranks on ONE GPU over gloo (every stage of the N > 1 line runs, incl. the early world check; the times
# mean nothing).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05g}
mkdir -p $O
cd $R
timeout -k 10 300 python tools/prof_event_cost.py > $O/prof_event_cost.json 2> $O/prof_event_cost.err; cat $O/prof_event_cost.json
T0=$(date +%s)
SCONE_ONE_DEVICE=1 SCONE_DIST_BACKEND=gloo timeout -k 10 800 python bench.py --gpus 2 --steps 10 --warmup 2 --time-budget 600 > $O/bench_2ranks_one_gpu_gloo_rehearsal.json 2> $O/bench_2ranks.err
echo "2-rank rehearsal rc=$? wall $(( $(date +%s) - T0 )) s"
grep -E "world_sanity|bench.py" $O/bench_2ranks.err | head -8
python3 - $O/bench_2ranks_one_gpu_gloo_rehearsal.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("n_gpus", r["n_gpus"], "value %.3f G" % (r["value"] / 1e9), "world_sanity", r.get("world_sanity"), "incomplete", r.get("incomplete"), "hung", r.get("hung_stage"))
s = r.get("sharded", {})
print("sharded keys", sorted(s)[:20])
for k, e in (s.get("exchanges") or {}).items():
    print(" ", k, {x: e.get(x) for x in ("ms_per_step", "status_bits", "error", "skipped", "records_transport") if e.get(x) is not None})
print("agree", s.get("exchanges_agree"), "configs" in r)
PY
