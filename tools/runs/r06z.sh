#!/bin/bash
# Round 6, final tree: the GPU suite, smoke(), the default bench line exactly as the driver runs it, one line per single-GPU
# config, all on one box.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06z}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1
echo "pytest rc=$? $(tail -2 $O/gpu_tests.txt | tr '\n' ' ' | head -c 300)"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$? $(tail -1 $O/smoke.txt)"
T0=$(date +%s)
timeout -k 10 700 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench (driver's command) rc=$? wall $(( $(date +%s) - T0 )) s"
cp gpurun_out/bench_details_n1.json $O/bench_details.json
echo "printed line: $(tail -1 $O/bench.json | wc -c) characters"
python3 tools/show_bench.py $O/bench.json
T0=$(date +%s)
timeout -k 10 700 python3 bench.py > $O/bench_default_flags.json 2> $O/bench_default_flags.err; echo "bench (no flags) rc=$? wall $(( $(date +%s) - T0 )) s"
python3 tools/show_bench.py $O/bench_default_flags.json | head -4
bash tools/run_configs.sh $TAG 2>&1 | grep -v "^==" | tail -12
