#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04i}
mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
echo "pytest rc=$? $(tail -3 $O/pytest_gpu.log | head -c 600)"
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; tail -c 600 $O/bench.err
python - $O/bench.json <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith('{"metric"'):
        d=json.loads(line)
        print({k:d[k] for k in ("value","ms_per_step","n_gpus")}, d["roofline"].get("frac"), d.get("incomplete"))
        z=d["sharded"].get("n1_pinned_host_zipf"); print(json.dumps(z)[:1500])
        print(json.dumps(d["sharded"].get("n1_pinned_host"))[:300])
PY
