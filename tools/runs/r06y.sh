#!/bin/bash
# Round 6, final tree: (i) the fuzz suite at 12,000 cases, (ii) the single-process soaks of the staging code (this round: the
# pipeline picks its side streams by probe and is dropped on a failed call), (iii) the N > 1 bench record rehearsed with 4 and 6
# ranks on ONE GPU over gloo (the sharded cache's side stream is now picked by probe; the line carries `sharded_summary`).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06y}
mkdir -p $O
cd $R
SCONE_FUZZ_CASES=12000 timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -x -q > $O/fuzz_12000_cases.txt 2>&1; echo "fuzz rc=$? $(tail -1 $O/fuzz_12000_cases.txt)"
timeout -k 10 200 python tools/soak.py 90 > $O/soak_single_process_cache_holds_everything.txt 2>&1; echo "soak1 rc=$? $(tail -2 $O/soak_single_process_cache_holds_everything.txt | tr '\n' ' ' | head -c 400)"
SCONE_SOAK_UNIFORM=1 SCONE_SOAK_CACHE_ROWS=200000 timeout -k 10 200 python tools/soak.py 90 > $O/soak_single_process_eviction_in_every_batch.txt 2>&1; echo "soak2 rc=$? $(tail -2 $O/soak_single_process_eviction_in_every_batch.txt | tr '\n' ' ' | head -c 400)"
for W in 4 6; do
  ROWS=$(( 80000000 / W ))
  T0=$(date +%s)
  SCONE_ONE_DEVICE=1 SCONE_DIST_BACKEND=gloo timeout -k 10 700 python bench.py --gpus $W --steps 5 --warmup 2 --time-budget 600 \
    --sharded-rows-per-rank $ROWS --pinned-rows $ROWS --cpu-seconds 2 > $O/bench_${W}ranks_one_gpu_gloo_rehearsal.json 2> $O/bench_${W}ranks.err
  echo "$W-rank rehearsal rc=$? wall $(( $(date +%s) - T0 )) s; $(grep world_sanity $O/bench_${W}ranks.err | head -1)"
  python3 - $O/bench_${W}ranks_one_gpu_gloo_rehearsal.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
s = r.get("sharded", {})
print(" n_gpus", r["n_gpus"], "chars", len(json.dumps(r)), "incomplete", r.get("incomplete"), "hung", r.get("hung_stage"), "agree", s.get("exchanges_agree"))
print(" sharded_summary", json.dumps(r.get("sharded_summary")))
for k, e in (s.get("exchanges") or {}).items():
    print("  ", k, {x: e.get(x) for x in ("ms_per_step", "status_bits", "error", "skipped", "transport_fallback_reason") if e.get(x) is not None})
PY
done
