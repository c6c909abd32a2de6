#!/bin/bash
# Round 6: the output buffer's placement.  (1) tools/placement_sensitivity.py (tables x out buffers), tools/out_placement_probe.py
# (separate allocations / windows of one block / shifted windows); (2) bench.py --quick with --out-candidates 1 and 4, alternating
# processes, four rounds.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06m}
mkdir -p $O
cd $R
python3 tools/placement_sensitivity.py > $O/placement.txt 2>&1; tail -9 $O/placement.txt
python3 tools/out_placement_probe.py > $O/out_placement.txt 2>&1; tail -9 $O/out_placement.txt
for round in 1 2 3 4; do
  for c in 1 4; do
    timeout -k 10 200 python3 bench.py --quick --steps 50 --warmup 5 --out-candidates $c > $O/cand${c}_$round.json 2> $O/cand${c}_$round.err
    python3 - $O/cand${c}_$round.json $c $round <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); rf = r["roofline"]
print("candidates %s round %s  step %.4f ms  kernel %.4f (min %.4f med %.4f)  %.3f G tok/s  %s" % (sys.argv[2], sys.argv[3], r["ms_per_step"], rf["avg_kernel_ms"],
      rf["kernel_ms"]["min"], rf["kernel_ms"]["median"], r["value"] / 1e9, r["config"].get("output_buffer")), flush=True)
PY
  done
done 2>&1 | tee $O/candidates.txt
