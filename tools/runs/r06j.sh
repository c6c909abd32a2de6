#!/bin/bash
# Round 6: r06i again (cached pinned-host step against the number of side streams the process used before) with the pipeline
# choosing its PREP / COPY streams by measured overlap (scone_stage_bind); then the staged-path GPU tests.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06j}
mkdir -p $O
cd $R
export SCONE_STAGE_TRACE=1
for k in 0 1 2 3 4 5 6 7 8 9 10; do
  timeout -k 10 300 python3 tools/stream_state_probe.py streams --n-streams $k 2> $O/err_$k.log | tail -1 | tee -a $O/stream_count.jsonl
  grep scone_stage_bind $O/err_$k.log | tee -a $O/bind.txt
done
timeout -k 10 300 python3 tools/stream_state_probe.py latency 2> $O/err_latency.log | tail -1 | tee -a $O/stream_count.jsonl
grep scone_stage_bind $O/err_latency.log | tee -a $O/bind.txt
unset SCONE_STAGE_TRACE
python -m pytest tests/test_gpu_prefetch.py tests/test_gpu_bench_shape.py -x -q -k "prefetch or pinned or staged or staging or cache or announcement" 2>&1 | tail -4
