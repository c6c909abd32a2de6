#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04r}
mkdir -p $O
cd $R
timeout -k 10 600 python bench.py --steps 50 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"
python - $O/bench.json <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith('{"metric"'):
        d=json.loads(line); rf=d["roofline"]
        print({k:d[k] for k in ("value","ms_per_step")}, rf["frac"], rf["traffic_stale"], rf["avg_kernel_ms"], d.get("incomplete"), d.get("driver_run_s"))
        z=d["sharded"]["n1_pinned_host_zipf"]
        print({k:z.get(k) for k in ("value","ms_per_step","rows_over_pcie_per_step","prefetch_beats_zero_copy","error")}, z.get("zero_copy_same_stream"))
PY
tools/run_configs.sh $1 2>&1 | grep -v "^==" | tail -12
