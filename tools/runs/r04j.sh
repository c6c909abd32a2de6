#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04j}
mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
echo "pytest rc=$? $(tail -3 $O/pytest_gpu.log | head -c 600)"
