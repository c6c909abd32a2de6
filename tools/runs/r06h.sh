#!/bin/bash
# Round 6: what in a process's stream state slows the staging pipeline (tools/stream_state_probe.py), one condition per process;
# then the two worst again with more hardware queues (GPU_MAX_HW_QUEUES=8, read by the HIP runtime at start).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06h}
mkdir -p $O
cd $R
for c in none streams graph lookups latency none; do
  timeout -k 10 300 python3 tools/stream_state_probe.py $c 2>> $O/err.log | tail -1 | tee -a $O/stream_state.jsonl
done
for c in streams latency; do
  GPU_MAX_HW_QUEUES=8 timeout -k 10 300 python3 tools/stream_state_probe.py $c 2>> $O/err.log | tail -1 | tee -a $O/stream_state.jsonl
done
