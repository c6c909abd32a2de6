#!/bin/bash
# Round 6: the cached pinned-host step against the NUMBER of side streams the process used before the pipeline was built
# (tools/stream_state_probe.py streams --n-streams k): the period of the pattern is the runtime's stream -> hardware-queue map.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06i}
mkdir -p $O
cd $R
for k in 0 1 2 3 4 5 6 7 8 9 10; do
  timeout -k 10 300 python3 tools/stream_state_probe.py streams --n-streams $k 2>> $O/err.log | tail -1 | tee -a $O/stream_count.jsonl
done
