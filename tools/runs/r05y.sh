#!/bin/bash
# Round 5: the compact printed line -- the multi-process bench tests (launched the driver's way and self-launched), the default
# run's line length, and the details file beside it.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05y}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_multiprocess.py -x -q -m gpu > $O/pytest_multiprocess.txt 2>&1
echo "pytest rc=$? $(tail -2 $O/pytest_multiprocess.txt | tr '\n' ' ' | head -c 300)"
timeout -k 10 700 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$? line $(tail -1 $O/bench.json | wc -c) chars"
cp gpurun_out/bench_details_n1.json $O/bench_details.json
python3 tools/show_bench.py $O/bench.json
