#!/bin/bash
# helper kernels of the columns flow at C5's true scale, per kernel (rocprofv3 kernel trace of the one-stream step), after the
# parity tests that cover them
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04k2}
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_bench_shape.py tests/test_gpu_parity.py -x -q -m gpu -k "shard or exchange or cols or sync_free" > $O/pytest.txt 2>&1
echo "pytest rc=$? $(tail -1 $O/pytest.txt)"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/c5_rank0_step.py --variants 2 --one-only --rounds 2 --steps 10 > $O/c5.json 2> $O/c5.err
echo "c5 rc=$?"
f=$(ls $O/trace/*/*kernel_stats.csv | head -1)
cp $f $O/kernel_stats.csv
python3 - $f <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    n = row["Name"]
    if any(k in n for k in ("k_cols", "k_gather_claim", "k_match_ell", "k_embed_wave", "k_claim", "k_plan", "pack")):
        print("  %-44s calls %5s avg %9.1f us" % (n.split("(")[0][-44:], row["Calls"], float(row["AverageNs"]) / 1e3))
PY
